"""Does ray re-ordering pay for the traversal kernel?  (experiment, GPU box)

Builds bounce-k ray batches of the Sponza BASELINE frame with the public
rt_trace_rays API (primary rays in the pipeline's 8x8-tile slot order, then
cosine-distributed bounces off the reported hits), and times the traversal
kernel on the same batch in its natural order and after sorting by a few keys.
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402


def tiled_pixels(W, H):
    tx = (W + 7) // 8
    ty = (H + 7) // 8
    s = np.arange(tx * ty * 64, dtype=np.int64)
    t, i = s // 64, s % 64
    x = (t % tx) * 8 + (i % 8)
    y = (t // tx) * 8 + (i // 8)
    ok = (x < W) & (y < H)
    return x[ok], y[ok]


def primary(pc, W, H):
    x, y = tiled_pixels(W, H)
    cam = pc.camInfo
    ph = cam.nearPlane * np.tan(np.radians(cam.fov / 2)) * 2
    pw = ph * cam.aspectRatio
    p = np.stack([-pw / 2 + pw * (x / W), -ph / 2 + ph * (y / H), np.full(x.shape, 0.1)], 1)
    d = p / np.linalg.norm(p, axis=1, keepdims=True)
    R = np.array(list(cam.cameraRotation), np.float64).reshape(4, 4).T  # column-major -> matrix
    d = d @ R[:3, :3].T
    o = np.broadcast_to(np.array(list(cam.pos), np.float64), d.shape)
    return o.astype(np.float32), d.astype(np.float32)


def bounce(h, rng):
    ok = h["didHit"] != 0
    n = h["normal"][ok].astype(np.float64)
    p = h["hitPoint"][ok].astype(np.float64)
    r1, r2 = rng.random(len(n)), rng.random(len(n))
    a = np.where(np.abs(n[:, :1]) < 1, np.array([[1.0, 0, 0]]), np.array([[0, 0, 1.0]]))
    t = np.cross(n, a)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    phi = 2 * np.pi * r1
    d = t * (np.cos(phi) * np.sqrt(r2))[:, None] + b * (np.sin(phi) * np.sqrt(r2))[:, None] + n * np.sqrt(1 - r2)[:, None]
    return (p + n * 1e-5).astype(np.float32), d.astype(np.float32)


def morton3(q, bits):
    k = np.zeros(len(q), np.int64)
    for b in range(bits):
        for a in range(3):
            k |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return k


def keys(o, d, kind, grid):
    octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
    lo, hi = o.min(0), o.max(0)
    q = np.minimum(((o - lo) / (hi - lo + 1e-9) * grid).astype(np.int64), grid - 1)
    bits = int(np.log2(grid))
    m = morton3(q, bits)
    if kind == "octant":
        return octant
    if kind == "origin":
        return m
    if kind == "octant_origin":
        return (octant << (3 * bits)) | m
    if kind == "origin_octant":
        return (m << 3) | octant
    if kind == "random":
        return np.random.default_rng(5).permutation(len(o))
    raise KeyError(kind)


def timed(r, o, d, reps=3):
    best = 1e9
    for _ in range(reps):
        r.reset_counters()
        h = r.trace_rays(o, d)
        ms, n = r.trace_time_ms()
        best = min(best, ms)
    return best, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--grid", type=int, default=16)
    ap.add_argument("--phase-stats", action="store_true")
    a = ap.parse_args()
    W, H = a.width, a.height
    scene, label = scenes.CONFIGS[a.scene]()
    r = engine.Renderer(0)
    r.upload_scene(scene)
    r.set_profiling(True)
    r.set_tuning("pipeline", 0)
    pc = (scenes.sponza_camera if a.scene.startswith("sponza") else engine.push_constants)(W, H)
    rng = np.random.default_rng(11)
    o, d = primary(pc, W, H)
    print(f"{label}: {len(o)} primary rays", flush=True)
    for k in range(a.bounces + 1):
        ms, hits = timed(r, o, d)
        c = r.counters()
        if a.phase_stats:
            r.set_tuning("phase_stats", 1)
            timed(r, o, d, reps=1)
            r.counters()
            r.set_tuning("phase_stats", 0)
        print(f"bounce {k}: {len(o)} rays  natural {ms:7.3f} ms  box/ray {c['boxTests'] / len(o):.1f} tri/ray {c['triTests'] / len(o):.1f}", flush=True)
        if k > 0:
            if not a.phase_stats:
                # upper bound of longest-ray-first scheduling: order by the cost this very batch was measured to have
                hn = engine.hits_to_numpy(hits)
                cost = hn["boxTests"].astype(np.int64) + hn["triTests"]
                for nm, idx in (("cost descending", np.argsort(-cost, kind="stable")), ("cost ascending", np.argsort(cost, kind="stable")),
                                ("longest 10 % first", np.argsort(-(cost >= np.percentile(cost, 90)).astype(np.int64), kind="stable"))):
                    ms2, _ = timed(r, o[idx], d[idx])
                    print(f"          {nm:24s} {ms2:7.3f} ms  ({ms / ms2:.2f}x)", flush=True)
            for kind in (() if a.phase_stats else ("octant", "origin", "octant_origin", "origin_octant", "random")):
                idx = np.argsort(keys(o, d, kind, a.grid), kind="stable")
                if a.phase_stats:
                    r.set_tuning("phase_stats", 1)
                ms2, _ = timed(r, o[idx], d[idx])
                if a.phase_stats:
                    r.counters()
                    r.set_tuning("phase_stats", 0)
                print(f"          sorted by {kind:14s} {ms2:7.3f} ms  ({ms / ms2:.2f}x)", flush=True)
        o, d = bounce(engine.hits_to_numpy(hits), rng)


if __name__ == "__main__":
    main()
