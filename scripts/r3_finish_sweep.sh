#!/bin/bash
# k_trace_finish threshold against one rank's share of the strong-scaling frame
for b in 262144 1048576 2097152 4194304 8388608; do
  echo "== finish_below=$b"
  CFTRT_FINISH=1,$b timeout -k 10 200 python3 scripts/strong_share.py 1 4 8 || exit 1
done
