"""The N > 1 path with the PRODUCT renderer: sample-range split, one sum of the per-rank f32 images.
 * two processes on the one GPU of the test box, gloo over CPU tensors (ray-tracing_amd/multi.py is the code bench.py
   runs over RCCL on the 8-GPU node);
 * the C ABI's own reduce (cftrt_comm_* / cftrt_reduce over RCCL) at world size 1."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import hip_mem as hm

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_processes_split_one_frame(rt, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_worker.py"), str(r), "2", str(port), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1 = np.load(tmp_path / "range0.npy"), np.load(tmp_path / "range1.npy")
    assert list(r0) == [0, 4] and list(r1) == [4, 7]           # balanced, contiguous, covering [0, 7)
    W, H, SPP, DEPTH, SEED = 320, 184, 7, 50, 0x5EED0002
    world, cam = rt.scenes.s500(width=W, height=H, spp=SPP)
    whole = world.render(cam, W, H, DEPTH, seed=SEED, spp_begin=0, spp_end=SPP)
    for (b, e), part in ((r0, np.load(tmp_path / "part0.npy")), (r1, np.load(tmp_path / "part1.npy"))):
        # a rank's samples are the very paths the single process traces for that range: bit for bit
        assert np.array_equal(part, world.render(cam, W, H, DEPTH, seed=SEED, spp_begin=int(b), spp_end=int(e)))
    got = np.load(tmp_path / "sum.npy").reshape(H, W, 3)
    assert np.allclose(got, whole, rtol=1e-5, atol=1e-5)        # only the f32 order of the per-range sums differs
    assert got.sum() > 0


def test_c_abi_reduce_world_size_one(rt):
    """cftrt_comm_unique_id / cftrt_comm_init / cftrt_reduce / cftrt_comm_destroy through the C ABI: RCCL is loaded,
    a communicator of one rank is made on the ctx's device and the in-place f32 sum to root 0 leaves the rendered
    image as it was; the error paths return codes."""
    dev = rt.Device(0)
    world, cam = rt.scenes.cornell_rs()
    world.commit(dev)
    W, H = 64, 48
    acc = hm.DeviceArray(W * H * 3)
    world.render_device(cam, W, H, 10, acc.data_ptr(), seed=5, spp_begin=0, spp_end=4, device=dev)
    before = acc.numpy()
    assert before.any()
    with pytest.raises(rt.CftrtError) as e:
        dev.reduce(acc.data_ptr(), W * H * 3, 0)                 # no communicator yet
    assert e.value.code == 3
    uid = dev.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(rt.CftrtError):
        dev.comm_init(1, 1, uid)                                 # rank out of range
    dev.comm_init(1, 0, uid)
    with pytest.raises(rt.CftrtError):
        dev.comm_init(1, 0, uid)                                 # twice
    with pytest.raises(rt.CftrtError):
        dev.reduce(acc.data_ptr(), W * H * 3, 1)                 # root out of range
    dev.reduce(acc.data_ptr(), W * H * 3, 0)
    assert np.array_equal(acc.numpy(), before)
    dev.comm_destroy()
    dev.close()


def test_comm_calls_from_two_threads(rt):
    """cftrt.h invites one host thread per GPU: two threads, each with its OWN ctx (both on the one card here), go
    through cftrt_comm_unique_id / cftrt_comm_init / cftrt_reduce / cftrt_comm_destroy at the same time - the first use
    of RCCL (a dlopen behind a function-local static) races between them - each as a communicator of one rank, plus the
    argument and state errors.  Every call returns its code; nothing hangs or crashes."""
    import threading
    results, errors = {}, []
    start = threading.Barrier(2)

    def work(k):
        try:
            dev = rt.Device(0)
            acc = hm.DeviceArray(1024)
            start.wait(timeout=60)
            codes = []
            try:
                dev.reduce(acc.data_ptr(), 1024, 0)
            except rt.CftrtError as e:
                codes.append(e.code)                             # 3: no communicator yet
            uid = dev.comm_unique_id()
            for bad in ((0, 0), (1, 1), (1, -1)):                # size 0, rank = size, negative rank
                try:
                    dev.comm_init(bad[0], bad[1], uid)
                except rt.CftrtError as e:
                    codes.append(e.code)                         # 1 each
            dev.comm_init(1, 0, uid)
            try:
                dev.comm_init(1, 0, uid)
            except rt.CftrtError as e:
                codes.append(e.code)                             # 3: already has one
            dev.reduce(acc.data_ptr(), 1024, 0)
            dev.comm_destroy()
            dev.comm_destroy()                                   # idempotent
            dev.close()
            results[k] = codes
        except Exception as e:                                   # noqa: BLE001 - reported below
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errors, errors
    assert results == {0: [3, 1, 1, 1, 3], 1: [3, 1, 1, 1, 3]}
