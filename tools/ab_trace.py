#!/usr/bin/env python3
"""A/B of traversal-kernel variants and knobs in ONE process, interleaved rounds
(cdna_hip_programming.md §5.4 rule 24). Prints median wall ms and k_trace ms per
dispatch, and checks that every variant returns the same pixels and counters."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=2)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--phase-stats", type=int, nargs="?", const=1, default=0, help="1: wave times of the last launch, 2 + k: of launch k")
ap.add_argument("--bounce", type=int, default=8)
ap.add_argument("--ntris", type=int, default=0, help="sponza stand-in triangle count (0 = default)")
ap.add_argument("--row-stride", type=int, default=1, help="render only rows 0, s, 2s, ... (what rank 0 of s GPUs renders)")
ap.add_argument("--row0", type=int, default=0)
ap.add_argument("--rows", type=int, default=0, help="rows of the tile (0 = all that fit)")
ap.add_argument("--frames", type=int, default=1, help="progressive frames per measurement, submitted at once (rt_render_frames)")
ap.add_argument("--variants", default="v0;v1,refill=8;v1,refill=16;v1,refill=24;v1,refill=32;v1,refill=48")
args = ap.parse_args()

scene, label = scenes.sponza(0, ntris=args.ntris) if (args.ntris and args.scene == 'sponza') else scenes.CONFIGS[args.scene]()
cam = scenes.sponza_camera if args.scene.startswith("sponza") else engine.push_constants
W, H = args.width, args.height
pc = cam(W, H, singleRender=1, sampleLimit=args.spp, bounceLimit=args.bounce) if args.frames == 1 else \
    cam(W, H, raysPerPixel=args.spp, progressive=1, bounceLimit=args.bounce)
r = engine.Renderer(0)
r.upload_scene(scene)
if args.phase_stats:
    r.set_tuning("phase_stats", args.phase_stats)
variants = []
for v in args.variants.split(";"):
    parts = v.split(",")
    kv = {"trace_variant": int(parts[0][1:])} if parts[0][0] == "v" else {"pipeline": 1}
    for p in parts[1:]:
        k, x = p.split("=")
        kv[k] = int(x)
    variants.append((v, kv))
res = {v: [] for v, _ in variants}
ref_img = ref_cnt = None
base = {'pipeline': 0, 'trace_variant': 1}  # variants: v0 / v1 (multi-kernel) / fused
for rnd in range(args.rounds + 1):
    for name, kv in variants:
        for k, x in {**base, **kv}.items():
            r.set_tuning(k, x)
        r.reset_counters()
        r.set_profiling(True)
        t = time.perf_counter()
        if args.frames == 1:
            img = r.render(pc, W, H, row0=args.row0, rowStride=args.row_stride, nRows=args.rows or None)
        else:
            img = r.render_frames(pc, W, H, args.frames, row0=args.row0, rowStride=args.row_stride, nRows=args.rows or None)
        dt = (time.perf_counter() - t) * 1e3
        tms, nl = r.trace_time_ms()
        r.set_profiling(False)
        cnt = r.counters()
        key = {k: cnt[k] for k in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments")}
        if ref_img is None:
            ref_img, ref_cnt = img, key
        else:
            assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)), f"{name}: pixels differ"
            assert key == ref_cnt, f"{name}: counters differ {key} {ref_cnt}"
        if rnd > 0:
            res[name].append((dt, tms, nl))
print(f"{args.scene} ({label}) {W}x{H} {args.spp} spp; rays {ref_cnt['raysTraced']/1e6:.1f} M unique, {ref_cnt['raysReference']/1e6:.1f} M reference")
alg = 32.0 * ref_cnt["boxTests"] + 36.0 * ref_cnt["triTests"] + 100.0 * ref_cnt["raysHit"]
for name, _ in variants:
    a = np.array(res[name])
    wall, tr = np.median(a[:, 0]), np.median(a[:, 1])
    print(f"{name:28s} wall {wall:8.1f} ms (min {a[:,0].min():8.1f})  k_trace {tr:8.1f} ms  launches {int(a[0,2])}  "
          f"unique Mrays/s {ref_cnt['raysTraced']/wall/1e3:8.1f}  trace GB/s {alg/tr/1e6:8.1f} ({alg/tr/1e6/8000*100:.1f}% of 8 TB/s)")
