"""The N>1 path on CPU: two gloo ranks split one frame by sample range (ray-tracing_amd/multi.py) and
reduce their f32 accumulation images; the sum must equal the single-rank render.  The per-rank renderer
here is the CPU oracle (the GPU renderer takes its place on the GPU box; the split/reduce code is the same)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, SEED = 24, 16, 6, 6, 77


def _worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_binding as ob
    rt = importlib.import_module("ray-tracing_amd")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    world, cam = rt.scenes.cornell_rs()
    ow = ob.OracleWorld(world)
    accum = torch.zeros(H * W * 3, dtype=torch.float32)

    def render_range(b, e, acc):
        a, _ = ow.render(cam, W, H, DEPTH, SEED, b, e, nthreads=1)
        acc += torch.from_numpy(a.astype(np.float32).reshape(-1))

    b, e = rt.multi.render_sharded(render_range, SPP, accum, dist=dist, dst=0)
    np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([b, e]))
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sum_to_the_single_rank_image(tmp_path, rt, orc):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "range0.npy"), np.load(tmp_path / "range1.npy")
    assert list(r0) == [0, 3] and list(r1) == [3, 6]
    world, cam = rt.scenes.cornell_rs()
    full, _ = orc.OracleWorld(world).render(cam, W, H, DEPTH, SEED, 0, SPP, nthreads=2)
    got = np.load(tmp_path / "sum.npy")
    assert np.allclose(got, full.reshape(-1), rtol=1e-6, atol=1e-6)
    assert got.sum() > 0
