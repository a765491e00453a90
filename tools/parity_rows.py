#!/usr/bin/env python3
"""Rows of a BASELINE configuration rendered by the GPU (both pipelines) and by the oracle, compared bit for bit; prints where they differ.
usage: parity_rows.py --scene sponza_dragons_flat --width 3840 --height 2160 --spp 8 --stride 64 [--tune k=v,...]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes
from oracle import pyoracle
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza_dragons_flat"); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--spp", type=int, default=8); ap.add_argument("--stride", type=int, default=64); ap.add_argument("--row0", type=int, default=0)
ap.add_argument("--tune", default="")
a = ap.parse_args()
scene, label = scenes.CONFIGS[a.scene]()
cam = scenes.sponza_camera if a.scene.startswith("sponza") else engine.push_constants
W, H = a.width, a.height
pc = cam(W, H, raysPerPixel=a.spp, progressive=1, singleRender=0)
tile = dict(row0=a.row0, rowStride=a.stride, nRows=(H - a.row0 + a.stride - 1) // a.stride)
pyoracle.lib().oracle_set_light_queries(0)
ref, rc = pyoracle.render(scene, pc, W, H, threads=pyoracle.effective_cpus(), **tile)
r = engine.Renderer(0)
for kv in filter(None, a.tune.split(",")):
    k, v = kv.split("="); r.set_tuning(k, int(v))
r.upload_scene(scene)
for pipe in (0, 1):
    r.set_tuning("pipeline", pipe); r.reset_counters()
    img = r.render(pc, W, H, **tile)
    bad = np.argwhere((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
    c = r.counters()
    print(f"{a.scene} {W}x{H} {a.spp} spp rows {a.row0}::{a.stride}  pipeline {pipe} ({r.last_kernel()}): {len(bad)} of {img.shape[0] * img.shape[1]} pixels differ; "
          f"boxTests gpu {c['boxTests']} oracle {rc['boxTests']}  rays gpu {c['raysTraced']} oracle {rc['raysTraced']}", flush=True)
    for y, x in bad[:6]:
        print("   row", a.row0 + y * a.stride, "x", x, "gpu", img[y, x], "oracle", ref[y, x])
