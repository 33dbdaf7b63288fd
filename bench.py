#!/usr/bin/env python3
"""bench.py — Mrays/s (all bounces) of the trace loop on the synthetic S500 scene, 1920x1080, 64 spp,
max depth 50 (BASELINE.json configs[2]), on N GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full render of the frame: [generate +] (intersect + shade) x depth -> resolve into a device
accumulation buffer (+ for N>1 one RCCL reduce of the W*H*3 f32 image to rank 0).  The frame shards by SAMPLE
RANGE (ray-tracing_amd/multi.py, the code the tests run): `--scaling weak` (default) gives every rank its own 64
samples per pixel (the N-GPU image holds 64 N spp); `--scaling strong` splits the 64 samples of ONE frame over the
ranks.  The Philox stream is keyed by the global sample index.  The scene is resident in HBM before the timed
region; nothing crosses PCIe inside it.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 64, 50
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline():
    """The CPU restatement (oracle/, kind "port") timed on ALL of this process's host cores (sched_getaffinity) on a
    bounded sample of the same workload: same scene, camera and depth at a reduced image size and sample count,
    sized from a short probe so that the leg takes ~15 s whatever the core count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    rt = importlib.import_module("ray-tracing_amd")
    cores = host_cores()
    w, h = 960, 540
    world, cam = rt.scenes.s500(width=w, height=h, spp=1)
    ow = ob.OracleWorld(world)
    t0 = time.perf_counter()
    _, probe_rays = ow.render(cam, w, h, DEPTH, rt.scenes.SAMPLE_SEED, 0, 1, nthreads=cores)
    probe = time.perf_counter() - t0
    spp = int(max(2, min(64, round(15.0 / max(probe, 1e-3)))))
    t0 = time.perf_counter()
    _, rays = ow.render(cam, w, h, DEPTH, rt.scenes.SAMPLE_SEED, 1, 1 + spp, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"S500 {w}x{h} {spp}spp depth{DEPTH}: {rays} rays in {dt:.1f} s on {cores} threads "
                      f"(sched_getaffinity; f64 C++ restatement of the Rust reference, std::thread over rows; "
                      f"sample count sized by a 1-spp probe of {probe:.2f} s)"}


def committed_profile(pattern):
    """Newest committed profile summary matching profiles/<pattern> (round-tagged names sort by round)."""
    files = sorted(f for pat in pattern.split("|") for f in glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except (OSError, ValueError):
        return None, None


def hbm_traffic_per_launch(rays_per_step, launches_per_step):
    """HBM bytes per level-kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled for gfx950,
    WRITE_SIZE as is; scripts/hbm_traffic.py) scaled to this run's ray count; None if absent.  PMC counters cannot
    be read inside the timed run."""
    d, _ = committed_profile("r0*_hbm_traffic.json")
    if not d:
        return None
    try:
        per_ray = sum(v["bytes_per_ray"] for k, v in d["kernels"].items() if k.startswith("k_trace"))
    except (KeyError, TypeError):
        return None
    return per_ray * rays_per_step / max(launches_per_step, 1)


def kernel_source_hash():
    """sha256 over the kernel sources (the first 16 hex digits): what a committed counter profile was taken from."""
    h = hashlib.sha256()
    for f in ("kernels.h", "cftrt.hip", "device_math.h", "scene_dev.h"):
        h.update(open(os.path.join(ROOT, "ray-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def as_run_utilisation():
    """Instruction-based utilisation of the level kernels AS RUN (wave cull, grid) from the committed SQ counter pass
    (scripts/pmc_summary.py over `rocprofv3 --pmc SQ_*`): per kernel instance, VALU busy = SQ_ACTIVE_INST_VALU x 4 /
    (SIMDs x cycles), lanes active per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), and their
    product, the share of the chip's VALU lane-cycles that did work."""
    d, name = committed_profile("r0*_pmc_sq.json|r0*_pmc_sq_v*.json")   # (the S500 frame, not the per-config passes)
    if not d:
        return None
    out = {"source": "profiles/" + name, "kernels": []}
    # the profile is of the kernels as they were when it was taken: say which, and whether that is still this tree
    meta = None
    try:
        meta = json.load(open(os.path.join(ROOT, "profiles", name.replace(".json", ".meta.json"))))
    except (OSError, ValueError):
        pass
    out["profiled_source_sha256_16"] = meta.get("kernel_source_sha256_16") if meta else None
    out["current_source_sha256_16"] = kernel_source_hash()
    out["stale"] = out["profiled_source_sha256_16"] != out["current_source_sha256_16"]
    if out["stale"]:
        out["note"] = ("the kernel sources changed after this profile was taken (or it carries no hash): "
                       "figures describe the profiled build, not this one")
    for e in d:
        c = e.get("largest_dispatch", {})
        if "k_trace" not in e.get("name", ""):
            continue
        try:
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
            busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles)
            lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        except (KeyError, ZeroDivisionError):
            continue
        out["kernels"].append({"kernel": e["name"].replace("void ", ""), "duration_us": c.get("duration_us"),
                               "valu_busy": busy, "lanes_active_per_valu_inst": lanes,
                               "valu_lane_cycles_used": busy * lanes, "valu_insts": c.get("SQ_INSTS_VALU")})
    return out if out["kernels"] else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 64 spp per rank; strong: the 64 spp of one frame split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain-scan", action="store_true", help="skip the plain-linear-scan leg (N=1 only)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    args = ap.parse_args()

    import torch   # before the product library: one HIP runtime per process
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
    if os.environ.get("CFTRT_TILE_LISTS"):
        dev.set_tile_lists(int(os.environ["CFTRT_TILE_LISTS"]))
    if os.environ.get("CFTRT_SHARDS"):
        dev.set_shards(int(os.environ["CFTRT_SHARDS"]))
    scene.commit(dev)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=f"cuda:{local_rank}")
    # the whole job's sample count: weak = spp per rank, strong = spp in all (ray-tracing_amd/multi.py cuts the range)
    spp_total = spp * world_size if args.scaling == "weak" else spp
    last = {}

    def render_range(b, e, acc):
        last["st"] = scene.render_device(cam, W, H, DEPTH, acc.data_ptr(), seed=rt.scenes.SAMPLE_SEED,
                                         spp_begin=b, spp_end=e, device=dev)

    def step():
        accum.zero_()
        torch.cuda.synchronize()          # the library runs on its own stream (include/cftrt.h, "Device and stream")
        last["st"] = None
        rt.multi.render_sharded(render_range, spp_total, accum, dist=dist if distributed else None, dst=0)
        return last["st"]

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
    stats = [s for s in stats if s is not None]   # (a rank with an empty sample range rendered nothing)

    # N = 1 only: the same frame with the wave cull, the grid and the per-tile lists off, i.e. the reference's plain linear scan, where
    # every ray tests every sphere and the algorithmic flops of SURVEY 8d ARE the executed ones.  This leg, timed live
    # with the library's HIP events, is what `roofline.achieved` / `frac` report: it is the only reading in which the
    # flop count bounds the kernel.  The headline `value` above is the default path (cull + grid), which skips most of
    # those tests; its work-delivered figure sits beside it as `delivered_equiv`.
    plain = None
    if world_size == 1 and not args.no_plain_scan and not split:
        before = dev.settings()
        dev.set_grid(0)
        dev.set_wave_cull(0)
        dev.set_tile_lists(0)
        try:
            step()
            fence()
            t1 = time.perf_counter()
            ps = [step() for _ in range(2)]
            fence()
            dt1 = time.perf_counter() - t1
        finally:
            dev.restore(before)       # (whatever the CFTRT_* environment had chosen, not constants)
        pms = sum(x.ms_intersect for x in ps)
        pach = sum(x.flops for x in ps) / (pms * 1e-3) / 1e12
        pl = sum(int(x.launches_intersect) for x in ps)
        plain = {"achieved": pach, "frac": pach / PEAK_FP32_VALU_TFLOPS, "launches": pl, "avg_launch_ms": pms / max(pl, 1),
                 "algorithmic_flops_per_launch": sum(x.flops for x in ps) / max(pl, 1),
                 "ms_per_step": dt1 / 2 * 1e3, "mrays_per_s": sum(int(x.rays) for x in ps) / dt1 / 1e6}

    # N > 1 only: the other reading of the metric ("1080p 64 spp at 1/2/4/8 GPU") in the same run - the 64 samples of ONE
    # frame split over the ranks (strong scaling) when the headline ran weak, and the other way round; timed like the
    # headline (barrier + synchronize on both sides, MAX over ranks below).
    other = None
    if distributed:
        other_mode = "strong" if args.scaling == "weak" else "weak"
        spp_other = spp if other_mode == "strong" else spp * world_size

        def step_other():
            accum.zero_()
            torch.cuda.synchronize()
            last["st"] = None
            rt.multi.render_sharded(render_range, spp_other, accum, dist=dist, dst=0)
            return last["st"]

        step_other()
        fence()
        t2 = time.perf_counter()
        so = [step_other() for _ in range(args.steps)]
        fence()
        dt2 = time.perf_counter() - t2
        so = [x for x in so if x is not None]
        tt = torch.tensor([dt2, float(sum(int(x.rays) for x in so))], dtype=torch.float64, device=f"cuda:{local_rank}")
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        other = {"scaling": other_mode, "spp_in_all": spp_other, "value": float(tt[1].item()) / float(tmax[0].item()) / 1e6,
                 "unit": "Mrays/s", "ms_per_step": float(tmax[0].item()) / args.steps * 1e3, "steps": args.steps}

    rays = sum(int(s.rays) for s in stats)
    t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
    r = torch.tensor([float(rays)], dtype=torch.float64, device=f"cuda:{local_rank}")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed_max, rays_all = float(t.item()), float(r.item())

    if rank == 0:
        nst = max(len(stats), 1)
        ms_isect = sum(s.ms_intersect for s in stats)
        fl_isect = sum((s.flops_intersect if split else s.flops) for s in stats)
        n_launch = sum(int(s.launches_intersect) for s in stats)
        delivered = fl_isect / (ms_isect * 1e-3) / 1e12 if ms_isect > 0 else 0.0
        s0 = stats[-1]
        kernel = "k_intersect" if split else "k_trace_level"
        roof = {"bound": "valu_fp32", "kernel": kernel, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s"}
        if plain:
            roof["plain_scan"] = {"achieved": plain["achieved"], "frac": plain["frac"], "ms_per_step": plain["ms_per_step"],
                                  "mrays_per_s": plain["mrays_per_s"], "launches": plain["launches"],
                                  "avg_launch_ms": plain["avg_launch_ms"]}
            roof.update({"achieved": plain["achieved"], "frac": plain["frac"], "launches": plain["launches"],
                         "avg_launch_ms": plain["avg_launch_ms"],
                         "algorithmic_flops_per_launch": plain["algorithmic_flops_per_launch"],
                         "flops_per_ray": 16 * 500 + 200,
                         "basis": "executed work: the level kernel as the reference's plain linear scan (wave cull and grid "
                                  "off; every ray tests every sphere, so SURVEY 8d's 16 flop x 500 spheres + 200 per ray are "
                                  "the flops executed), timed live with HIP events in an extra leg of this run, all "
                                  "level launches and k_trace_finish in the denominator (= roofline.plain_scan); the "
                                  "timed `value` runs the default path (cone cull, tile lists, grid), whose kernels "
                                  "skip most of those tests: its figures are delivered_equiv and as_run_utilisation",
                         "plain_scan_ms_per_step": plain["ms_per_step"], "plain_scan_mrays_per_s": plain["mrays_per_s"]})
        else:
            roof.update({"achieved": None, "frac": None,
                         "basis": "the executed-work leg (plain linear scan) runs at N = 1 in the fused pipeline only"})
        roof["traffic"] = hbm_traffic_per_launch(int(s0.rays), n_launch // nst)
        roof["delivered_equiv"] = {
            "tflops": delivered, "of_peak": delivered / PEAK_FP32_VALU_TFLOPS, "launches": n_launch,
            "avg_launch_ms": ms_isect / max(n_launch, 1),
            "basis": "work DELIVERED by the default path per second of its level kernels: every ray is charged the "
                     "reference's whole scan although the wave cull and the grid skip most of it - not a utilisation, "
                     "and not bounded by the peak"}
        roof["as_run_utilisation"] = as_run_utilisation()
        out = {
            "metric": "Mrays/s (all bounces)",
            "value": rays_all / elapsed_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"S500 (500 spheres, diffuse/metal/glass, SkyLight) {W}x{H} max depth {DEPTH} "
                                   f"(BASELINE.json configs[2]); {spp_total} spp in all = "
                                   + (f"{spp} spp per GPU (weak)" if args.scaling == "weak" else f"{spp} spp split over the GPUs (strong)")
                                   + "; sample-range split, RCCL reduce of the f32 image",
                       "rays_per_step_per_gpu": int(s0.rays), "paths_per_step_per_gpu": int(s0.paths),
                       "rays_level1_per_gpu": int(s0.rays_per_level[1]),
                       "trace_levels": int(s0.levels), "exact_second_evals": int(s0.exact_second),
                       "exact_rescans": int(s0.exact_rescan)},
            "roofline": roof,
            "kernel_ms_per_step": {"generate": sum(s.ms_generate for s in stats) / nst,
                                   ("intersect" if split else "trace_level"): ms_isect / nst,
                                   "shade": sum(s.ms_shade for s in stats) / nst,
                                   "resolve": sum(s.ms_resolve for s in stats) / nst,
                                   "device_total": sum(s.ms_total for s in stats) / nst,
                                   "launches_per_step": n_launch / nst},
        }
        if other:
            out[other["scaling"]] = other
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
