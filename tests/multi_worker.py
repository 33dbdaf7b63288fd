"""One rank of the two-process test in test_y_multi_gpu.py (run as a script: rank world port outdir).
Renders its sample range of one frame with the product renderer (World.render_device on cuda:0; both ranks share
the one GPU of the test box, each with its own ctx) and reduces the f32 images with torch.distributed - gloo on CPU
tensors here, RCCL on the real multi-GPU node (bench.py).  torch is imported BEFORE the product library so the
process holds one HIP runtime."""
import importlib
import os
import sys

import numpy as np


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    rt = importlib.import_module("ray-tracing_amd")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H, SPP, DEPTH, SEED = 320, 184, 7, 50, 0x5EED0002
    scene, cam = rt.scenes.s500(width=W, height=H, spp=SPP)
    dev = rt.Device(0)
    scene.commit(dev)
    accum = torch.zeros(H * W * 3, dtype=torch.float32)

    def render_range(b, e, acc):
        a = scene.render(cam, W, H, DEPTH, seed=SEED, spp_begin=b, spp_end=e, device=dev)   # host buffer: gloo reduces on CPU
        acc += torch.from_numpy(a.reshape(-1))

    b, e = rt.multi.render_sharded(render_range, SPP, accum, dist=dist, dst=0)
    np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([b, e]))
    np.save(os.path.join(out_dir, f"part{rank}.npy"), scene.render(cam, W, H, DEPTH, seed=SEED, spp_begin=b, spp_end=e, device=dev))
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
