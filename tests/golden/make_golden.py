#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/raytrace_oracle.cpp).

The reference holds no golden vectors for this path and cannot be run here
(SURVEY §4, §8c), so these fixtures are produced by the oracle itself, after it
passed the known-answer tests in tests/test_oracle_kat.py. They pin the oracle
against compiler / platform drift and let the GPU tests check the HIP path
without executing the oracle at all.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle  # noqa: E402
from ray_tracer_amd import engine  # noqa: E402
from util import cornell_scene, model_scene, seeded_rays  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # C1 at reduced size: Cornell + dielectric / mirror / diffuse spheres
    s = cornell_scene(True)
    W = H = 64
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=4)
    img, cnt = pyoracle.render(s, pc, W, H, threads=1)
    np.savez_compressed(os.path.join(OUT, "cornell_c1_64x64_4spp.npz"), rgba=img,
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.uint64), counter_names=np.array(sorted(cnt)))
    # bunny (908 tris) diffuse inside the Cornell box, 2 spp, frame 3
    s = model_scene("bunny.obj")
    W, H = 64, 48
    pc = engine.push_constants(W, H, raysPerPixel=2, frameCount=3)
    img, cnt = pyoracle.render(s, pc, W, H, threads=1)
    np.savez_compressed(os.path.join(OUT, "bunny908_64x48_2spp_frame3.npz"), rgba=img,
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.uint64), counter_names=np.array(sorted(cnt)))
    # per-ray hit records of calculateIntersections
    for name, sc in (("cornell", cornell_scene(True)), ("bunny908", model_scene("bunny.obj"))):
        o, d = seeded_rays(1024, seed=11)
        h = engine.hits_to_numpy(pyoracle.trace_rays(sc, o, d))
        np.savez_compressed(os.path.join(OUT, f"hits_{name}_1024.npz"), origins=o, dirs=d, **h)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
