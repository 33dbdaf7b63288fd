#!/usr/bin/env python3
"""bench.py — Mrays/s (all bounces) of the trace loop on the synthetic S500 scene, 1920x1080, 64 spp,
max depth 50 (BASELINE.json configs[2]), on N GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full render of the frame: generate -> [intersect -> shade]*depth -> resolve into a device
accumulation buffer (+ for N>1 one RCCL reduce of the W*H*3 f32 image to rank 0).  Scaling is WEAK: every
rank renders its own 64 samples per pixel (sample range [64*rank, 64*(rank+1)); the Philox stream is keyed
by the global sample index), so the N-GPU image holds 64*N spp.  The scene is resident in HBM before the
timed region; nothing crosses PCIe inside it.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 64, 50
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"


def cpu_baseline():
    """The CPU restatement (oracle/, kind "port") timed on this box's host cores on a bounded sample of
    the same workload: same scene, camera, depth; 960x540 at 32 spp (1/4 of the pixels, 1/2 of the spp)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    rt = importlib.import_module("ray-tracing_amd")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    w, h, spp = 960, 540, 32
    world, cam = rt.scenes.s500(width=w, height=h, spp=spp)
    ow = ob.OracleWorld(world)
    t0 = time.perf_counter()
    _, rays = ow.render(cam, w, h, DEPTH, rt.scenes.SAMPLE_SEED, 0, spp, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"S500 {w}x{h} {spp}spp depth{DEPTH}: {rays} rays in {dt:.1f} s (f64 C++ restatement of "
                      "the Rust reference, std::thread over rows)"}


def hbm_traffic_per_launch(rays_per_step, launches_per_step, kernel="k_trace_level"):
    """HBM bytes per k_intersect launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled for gfx950,
    WRITE_SIZE as is; profiles/r01_hbm_traffic.json) scaled to this run's ray count; None if absent.
    PMC counters cannot be read inside the timed run."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        k = json.load(open(path))["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None
    return k["bytes_per_ray"] * rays_per_step / max(launches_per_step, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain-scan", action="store_true", help="skip the untimed plain-linear-scan leg (N=1 only)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    distributed = world_size > 1
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rt = importlib.import_module("ray-tracing_amd")
    W, H, spp = args.width, args.height, args.spp
    scene, cam = rt.scenes.s500(width=W, height=H, spp=spp)
    dev = rt.Device(local_rank)
    if os.environ.get("CFTRT_VARIANT"):
        dev.set_variant(int(os.environ["CFTRT_VARIANT"]))
    split = os.environ.get("CFTRT_PIPELINE", "fused") == "split"   # profiling aid: the two stages as separate kernels
    dev.set_pipeline(split)
    if os.environ.get("CFTRT_WAVE_CULL"):
        dev.set_wave_cull(int(os.environ["CFTRT_WAVE_CULL"]))   # 0: plain linear scan, for profiling
    if os.environ.get("CFTRT_GRID"):
        g = os.environ["CFTRT_GRID"].split(",")                # "mode[,cells_per_prim]"; 0: the plain lists, for profiling
        dev.set_grid(int(g[0]), float(g[1]) if len(g) > 1 else 0.0)
    if os.environ.get("CFTRT_FINISH"):
        f = os.environ["CFTRT_FINISH"].split(",")              # "on[,below]": deep levels path by path (profiling)
        dev.set_finish(int(f[0]), int(f[1]) if len(f) > 1 else 0)
    scene.commit(dev)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=f"cuda:{local_rank}")
    spp_begin, spp_end = rank * spp, (rank + 1) * spp

    def step():
        accum.zero_()
        torch.cuda.synchronize()          # the library runs on its own stream
        st = scene.render_device(cam, W, H, DEPTH, accum.data_ptr(), seed=rt.scenes.SAMPLE_SEED,
                                 spp_begin=spp_begin, spp_end=spp_end, device=dev)
        if distributed:
            dist.reduce(accum, dst=0)      # RCCL sum of the per-GPU f32 images over xGMI
        return st

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0

    # Untimed extra leg at N=1: the same frame with the wave cull and the grid off, i.e. the reference's plain linear
    # scan (every ray tests every sphere).  There the algorithmic flops are the executed ones, so this is the
    # roofline fraction that speaks about the kernel's arithmetic; the timed region above runs the default path.
    plain = None
    if world_size == 1 and not args.no_plain_scan and not split:
        dev.set_grid(0)
        dev.set_wave_cull(0)
        try:
            step()
            fence()
            t1 = time.perf_counter()
            ps = [step() for _ in range(2)]
            fence()
            dt1 = time.perf_counter() - t1
        finally:
            dev.set_grid(1)
            dev.set_wave_cull(1)
        pms = sum(x.ms_intersect for x in ps)
        pach = sum(x.flops for x in ps) / (pms * 1e-3) / 1e12
        plain = {"achieved": pach, "frac": pach / PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                 "avg_launch_ms": pms / max(sum(int(x.launches_intersect) for x in ps), 1),
                 "ms_per_step": dt1 / 2 * 1e3, "mrays_per_s": sum(int(x.rays) for x in ps) / dt1 / 1e6,
                 "basis": "wave cull and grid off: every ray tests every sphere, algorithmic flops = executed flops"}

    rays = sum(int(s.rays) for s in stats)
    t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
    r = torch.tensor([float(rays)], dtype=torch.float64, device=f"cuda:{local_rank}")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed_max, rays_all = float(t.item()), float(r.item())

    if rank == 0:
        ms_isect = sum(s.ms_intersect for s in stats)
        # split: the scan's flops over k_intersect's time; fused: the level kernel does scan + shading, so its
        # algorithmic work is the whole per-ray figure of SURVEY 8d (16 Nsph + 44 Ntri + 200)
        fl_isect = sum((s.flops_intersect if split else s.flops) for s in stats)
        n_launch = sum(int(s.launches_intersect) for s in stats)
        achieved = fl_isect / (ms_isect * 1e-3) / 1e12 if ms_isect > 0 else 0.0
        s0 = stats[-1]
        out = {
            "metric": "Mrays/s (all bounces)",
            "value": rays_all / elapsed_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"S500 (500 spheres, diffuse/metal/glass, SkyLight) {W}x{H} {spp}spp/GPU "
                                   f"max depth {DEPTH} (BASELINE.json configs[2]); sample-range split over GPUs, "
                                   "RCCL reduce of the f32 image",
                       "rays_per_step_per_gpu": int(s0.rays), "paths_per_step_per_gpu": int(s0.paths),
                       "trace_levels": int(s0.levels), "exact_second_evals": int(s0.exact_second),
                       "exact_rescans": int(s0.exact_rescan)},
            "roofline": {"bound": "valu_fp32", "kernel": "k_intersect" if split else "k_trace_level", "achieved": achieved,
                         "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": hbm_traffic_per_launch(int(s0.rays), n_launch // max(args.steps, 1),
                                                            "k_intersect" if split else "k_trace_level"),
                         "launches": n_launch, "avg_launch_ms": ms_isect / max(n_launch, 1),
                         "algorithmic_flops_per_launch": fl_isect / max(n_launch, 1),
                         "flops_per_ray": 16 * 500 + (0 if split else 200), "whole_loop_tflops": sum(s.flops for s in stats) / elapsed_max / 1e12,
                         "basis": "reference-equivalent: every ray is charged the reference's whole linear scan (SURVEY 8d: "
                                  "16 flop x 500 spheres + 200); the wave cull and the grid skip most of those tests, so "
                                  "this is work delivered, not VALU work executed - see plain_scan for the latter",
                         "plain_scan": plain},
            "kernel_ms_per_step": {"generate": sum(s.ms_generate for s in stats) / args.steps,
                                   ("intersect" if split else "trace_level"): ms_isect / args.steps,
                                   "shade": sum(s.ms_shade for s in stats) / args.steps,
                                   "resolve": sum(s.ms_resolve for s in stats) / args.steps,
                                   "device_total": sum(s.ms_total for s in stats) / args.steps},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
