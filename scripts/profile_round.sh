#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ for the default bench (run on the GPU box from the repo root):
#   bash scripts/profile_round.sh r02     -> gpurun_out/prof_r02/*  (copy what is to be judged into profiles/ as r02_*)
# Kernel trace and counters are separate runs; each --pmc pass is its own run within the per-block counter limits.
set -e
tag=$1; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-scan > $out/bench_under_rocprof.json
cp $out/stats/p_kernel_stats.csv $out/kernel_stats.csv
python3 scripts/level_times.py $out/stats > $out/level_times.txt
CFTRT_GRID=0 CFTRT_WAVE_CULL=0 CFTRT_TILE_LISTS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_plain -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plain-scan > $out/bench_plain_scan_under_rocprof.json
cp $out/stats_plain/p_kernel_stats.csv $out/kernel_stats_plain_scan.csv
timeout -k 10 400 python3 bench.py > $out/bench.json
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_sq 0 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_lds 0 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_tcp 0 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_fetch 0 FETCH_SIZE
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_write 0 WRITE_SIZE
# the L2's fabric-side requests behind FETCH_SIZE / WRITE_SIZE, per dispatch (4 TCC counters a pass)
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_tcc_rd 0 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_tcc_wr 0 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum
cp gpurun_out/pmc_${tag}_tcc_rd.json $out/pmc_tcc_read.json
cp gpurun_out/pmc_${tag}_tcc_wr.json $out/pmc_tcc_write.json
# the plain linear scan (what roofline.frac is measured on): SQ counters per level
( export CFTRT_GRID=0 CFTRT_WAVE_CULL=0 CFTRT_TILE_LISTS=0
  PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_sq_plain 0 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE )
cp gpurun_out/pmc_${tag}_sq_plain.json $out/pmc_sq_plain_scan.json
python3 scripts/pmc_levels.py gpurun_out/pmc_${tag}_sq_plain > $out/pmc_sq_plain_scan_levels.txt
python3 scripts/pmc_levels.py gpurun_out/pmc_${tag}_sq > $out/pmc_sq_levels.txt
# which sources the counters describe (bench.py compares it with the tree it runs from)
python3 -c "
import sys; sys.path.insert(0, '.')
import bench, json
json.dump({'kernel_source_sha256_16': bench.kernel_source_hash()}, open('$out/pmc_sq.meta.json', 'w'))"
rays=$(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['config']['rays_per_step_per_gpu'])")
paths=$(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['config']['paths_per_step_per_gpu'])")
l1=$(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['config']['rays_level1_per_gpu'])")
python3 scripts/hbm_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write $rays $out/hbm_traffic.json $paths $l1 > /dev/null
cp gpurun_out/pmc_${tag}_sq.json $out/pmc_sq.json
cp gpurun_out/pmc_${tag}_lds.json $out/pmc_lds_occupancy.json
cp gpurun_out/pmc_${tag}_tcp.json $out/pmc_ta_tcp.json
# C2 cornell (the reference's own second scene): kernel times and the SQ counters that name its bound
bash scripts/trace_config.sh ${tag}_c2 C2 > $out/c2_level_times.txt
bash scripts/pmc_config.sh ${tag}_c2_sq C2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
cp gpurun_out/pmc_${tag}_c2_sq.json $out/pmc_sq_c2_cornell.json
cp gpurun_out/trace_${tag}_c2/p_kernel_stats.csv $out/kernel_stats_c2_cornell.csv
# C5 S20k (grid walk at every level): per-level times and SQ / TA counters
bash scripts/trace_config.sh ${tag}_c5 C5 > $out/c5_level_times.txt
bash scripts/pmc_config.sh ${tag}_c5_sq C5 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
cp gpurun_out/pmc_${tag}_c5_sq.json $out/pmc_sq_c5_s20k.json
# where a wave's cycles go: parked at s_waitcnt / issue stalls / issuing (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES)
bash scripts/r3_wait_pass.sh ${tag} > $out/wave_cycles.txt
cp gpurun_out/pmc_${tag}_wait.json $out/pmc_wave_cycles.json
cp gpurun_out/pmc_${tag}_c5_wait.json $out/pmc_wave_cycles_c5_s20k.json
# shares of a wave's time by phase and the walk's counters (diagnostic build with shader-clock stamps)
CFTRT_STAMPS=1 CFTRT_LIB=$PWD/ray-tracing_amd/csrc/libcftrt_stamps.so python3 scripts/bench_configs.py C3 C2 C5 2> $out/stamps.txt > /dev/null
python3 scripts/bench_configs.py C1 C2 C2p C3 C5 C4 > $out/configs.jsonl
# one rank's share of the strong-scaling frame (64 spp split over N), and the timeline of the N = 8 share
python3 scripts/strong_share.py > $out/strong_share.txt
mkdir -p $out/tl8
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl8 -o p -- python3 scripts/strong_share.py 8 > /dev/null 2>&1
python3 scripts/timeline.py $out/tl8 > $out/strong_share_n8_timeline.txt
ls -la $out
