"""Generates the committed fixtures under tests/golden/ from the CPU oracle (oracle/oracle.cpp).

The reference itself cannot run here (Rust, no toolchain: SURVEY.md §8c), so these are outputs of the
f64 restatement, pinned against the reference's own unit tests by tests/test_oracle_kat.py.
Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracing_amd")
import oracle_binding as ob  # noqa: E402

W, H = 64, 40
SEED = 0x5EED0002


def primary(name, world, cam):
    ow = ob.OracleWorld(world)
    rays64 = ob.emit_rays(cam, W, H, 0, W * H, SEED, jitter=False)
    rays = rays64.astype(np.float32)            # the f32 rays both sides are given
    ids, t, nrm, pt, fl = ow.hit(rays.astype(np.float64), seed=SEED)
    np.savez_compressed(os.path.join(HERE, f"primary_{name}.npz"), rays=rays, rays64=rays64, ids=ids, t=t,
                        normal=nrm, point=pt, outward=fl)
    print(name, "hits", int((ids >= 0).sum()), "of", len(ids))


def image(name, world, cam, w, h, spp, depth):
    ow = ob.OracleWorld(world)
    acc, rays = ow.render(cam, w, h, depth, SEED, 0, spp, nthreads=8)
    np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), accum=acc.astype(np.float32), rays=np.int64(rays),
                        spp=np.int32(spp), depth=np.int32(depth))
    print(name, "image mean", acc.mean(axis=(0, 1)) / spp, "rays", rays)


if __name__ == "__main__":
    w1, c1 = rt.scenes.scene_rs(motion=False)
    w2, c2 = rt.scenes.cornell_rs()
    w3, c3 = rt.scenes.s500()
    primary("scene_rs", w1, c1)
    primary("cornell", w2, c2)
    primary("s500", w3, c3)
    image("cornell", w2, c2, 40, 30, 64, 10)
    image("s500", w3, c3, 48, 27, 32, 50)
