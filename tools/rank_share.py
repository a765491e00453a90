#!/usr/bin/env python3
"""What ONE of N GPUs does in bench.py's timed region: rank 0's rows of the bench frame, `--steps` progressive frames in groups of
at most 10 N (bench.py's policy), automatic pipeline choice. Prints ms per step and the projected whole-job speed-up."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--tune", default="", help="comma-separated rt_set_tuning knobs")
ap.add_argument("--worlds", default="1,2,4,8")
args = ap.parse_args()
W, H = 1920, 1080
scene, label = scenes.CONFIGS["sponza"]()
r = engine.Renderer(0)
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("="); r.set_tuning(k, int(v))
r.upload_scene(scene)
pc = scenes.sponza_camera(W, H, raysPerPixel=args.spp, progressive=1, singleRender=0)
def groups(count, fif):
    k = (count + fif - 1) // fif
    return [count // k + (1 if j < count % k else 0) for j in range(k)]
base = None
for world in [int(w) for w in args.worlds.split(',')]:
    tile = dict(row0=0, rowStride=world)
    for rep in range(2):   # the first pass warms up (ray cost, allocations)
        r.sync(); t = time.perf_counter(); i = 0
        for n in groups(args.steps, 10 * world):
            pc.frameCount = i
            r.render_frames(pc, W, H, n, sync=False, **tile) if n > 1 else r.render(pc, W, H, sync=False, **tile)
            i += n
        r.sync(); dt = time.perf_counter() - t
    ms = dt / args.steps * 1e3
    base = base or ms
    print(f"N = {world}: rank 0's rows, {args.steps} steps in groups of {groups(args.steps, 10 * world)}: {ms:.2f} ms per step "
          f"-> x{base / ms:.2f} ({['multi-kernel', 'fused'][r.last_pipeline()]})", flush=True)
