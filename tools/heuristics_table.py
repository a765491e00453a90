#!/usr/bin/env python3
"""Are the launch heuristics shaped by the BASELINE stand-ins? Scenes built from the assets that ship with the reference, structurally
unlike them, at 1920x1080 and 8 spp: the automatic choice (pipeline by paths per dispatch and measured ray cost, pixel hand-out,
interior-step threshold) against every forced choice, one frame per dispatch and ten frames in flight. Markdown on stdout.

  klein8      default Cornell box + klein_bottle.obj (35 840 triangles) instanced eight times under different placements
  bobadog     default Cornell box + bobadog.obj (eight usemtl groups with their own materials)
  objects45   tests/test_gpu_parity.py's 45-object scene (general / identity / leaf-root objects beyond the 32-bit object mask)
  bunnies256  N2's measurement: a floor and 16 x 16 separated, rotated instances of bunny.obj (908 triangles each) — what the
              reference's linear object loop (raytrace.comp:289-350) is worst at. --flat adds the same geometry baked into ONE mesh.
usage: heuristics_table.py [scene ...] [--flat] [--phase-stats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402

A = engine.ASSET_DIR


def klein8():
    s = engine.Scene(); s.prepare_storage_buffers()
    for k in range(8):
        where = (-0.6 + 0.4 * (k % 4), 0.1 - 0.5 * (k // 4), -0.3 + 0.25 * (k % 3))
        s.read_obj(os.path.join(A, "klein_bottle.obj"), engine.placement(position=where, scale=0.18 + 0.02 * k, rotation=(15 * k, 40 * k, 5 * k)), [0, 4, 5, 1][k % 4])
    return s, engine.push_constants


def bobadog():
    s = engine.Scene(); s.prepare_storage_buffers()
    s.read_obj(os.path.join(A, "bobadog", "bobadog.obj"), engine.placement(position=(0, 0.3, 0), scale=0.35, rotation=(0, 160, 0)), 0)
    return s, engine.push_constants


def objects45():
    s = engine.Scene(); s.prepare_storage_buffers()
    glow = s.add_material(engine.default_material(albedo=(0.9, 0.9, 0.3), emissionColor=(1.0, 0.9, 0.4), emissionStrength=2.0))
    mats = [0, 1, 2, 4, 5, glow]
    card = np.array([[[-0.06, 0, -0.06], [0.06, 0, -0.06], [0.06, 0, 0.06]], [[-0.06, 0, -0.06], [0.06, 0, 0.06], [-0.06, 0, 0.06]]], np.float32)
    ncard = np.zeros_like(card); ncard[..., 1] = -1
    for k in range(36):
        where = (-0.8 + 0.32 * (k % 6), -0.85 + 0.3 * (k // 6), -0.7 + 0.25 * (k % 5))
        if k % 3 == 0:
            pos, nrm = scenes.blob(96 + 8 * k, seed=40 + k, radius=1.0)
            s.add_mesh(f"g{k}", pos, nrm, engine.placement(position=where, scale=(0.09, 0.12, 0.07), rotation=(10 * k, 25 * k, 5 * k)), mats[k % 6])
        elif k % 3 == 1:
            pos, nrm = scenes.blob(64 + 6 * k, seed=40 + k, radius=0.09, center=where)
            s.add_mesh(f"i{k}", pos, nrm, engine.placement(), mats[k % 6])
        else:
            s.add_mesh(f"c{k}", card, ncard, engine.placement(position=where, rotation=(35 * k, 0, 20 * k)), mats[k % 6])
    return s, engine.push_constants


BUNNY_SCALE = float(os.environ.get("BUNNY_SCALE", "0.12"))   # 0.12: boxes 0.2 wide on a 1.0 grid (separated); 0.45: boxes that nearly tile the floor


def _bunny_grid():
    for k in range(256):
        gx, gz = k % 16, k // 16
        yield k, (-7.5 + gx, 0.5 - 0.33 * BUNNY_SCALE, -7.5 + gz), BUNNY_SCALE * (1.0 + 0.02 * (k % 7)), (0.0, 22.5 * k, 0.0)


def _lit_floor_scene():
    s = engine.Scene()
    for i in range(10):
        s.set_sphere(i, (0, 0, 0), 0.0, 0)
    for m in (engine.default_material(), engine.default_material(albedo=(1, 0, 0)), engine.default_material(albedo=(0, 1, 0)),
              engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4),
              engine.default_material(reflectance=1.0), engine.default_material(ior=2.0)):
        s.add_material(m)
    fpos, fnrm = scenes.grid_patch((-9.0, 0.5, -9.0), (18.0, 0, 0), (0, 0, 18.0), 8, 8)
    s.add_mesh("floor", fpos, fnrm, engine.placement(), 0)
    s.read_obj(os.path.join(A, "light2.obj"), engine.placement(position=(0, -1.5, 0), frontOnly=True), 3)
    cam = lambda W, H, **kw: engine.push_constants(W, H, pos=(0.0, -3.0, -11.0), cameraAngles=(18.0, 0.0, 0.0), fov=60.0, environmentOn=True, **kw)  # noqa: E731
    return s, cam


def bunnies256():
    s, cam = _lit_floor_scene()
    for k, where, scale, rot in _bunny_grid():
        s.read_obj(os.path.join(A, "bunny.obj"), engine.placement(position=where, scale=scale, rotation=rot), [0, 1, 2, 4, 5][k % 5])
    return s, cam


def bunnies256_flat():
    """The same 256 bunnies with their placements baked into the vertices, one mesh, one object (all diffuse: one material per object)."""
    s, cam = _lit_floor_scene()
    one = engine.Scene()
    one.read_obj(os.path.join(A, "bunny.obj"), engine.placement(), 0)
    a = one.numpy()
    pts = a["triPoints"].view(np.float32).reshape(-1, 8)          # TrianglePoint: pos.xyz, u | nrm.xyz, v
    idx = a["triangles"].view(np.uint32).reshape(-1, 12)[:, :3]    # Triangle: v0, v1, v2, frontOnly, ...
    pos, nrm = pts[idx][:, :, 0:3].copy(), pts[idx][:, :, 4:7].copy()
    P, N = [], []
    for k, where, scale, rot in _bunny_grid():
        fp, fn = scenes._bake_y(pos.reshape(-1, 9), nrm.reshape(-1, 9), where, scale, rot[1])
        P.append(fp); N.append(fn)
    s.add_mesh("flat:bunnies", np.concatenate(P), np.concatenate(N), engine.placement(), 0)
    return s, cam


def _config(key):
    def make():
        s, _ = scenes.CONFIGS[key]()
        return s, (scenes.sponza_camera if key.startswith("sponza") else engine.push_constants)
    return make


SCENES = {**{k: _config(k) for k in scenes.CONFIGS}, "klein8": klein8, "bobadog": bobadog, "objects45": objects45, "bunnies256": bunnies256, "bunnies256_flat": bunnies256_flat}


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["klein8", "bobadog", "objects45", "bunnies256"]
    if "--flat" in sys.argv:
        names.append("bunnies256_flat")
    W, H, SPP = 1920, 1080, 8
    r = engine.Renderer(0)
    forced = [("auto", {}), ("multi-kernel", {"pipeline": 0}), ("fused, pixels replaced at 8 free lanes", {"pipeline": 1, "pixel_refill": 8}),
              ("fused, a block at a time", {"pipeline": 1, "pixel_refill": 64}), ("fused, fast_lanes 24", {"pipeline": 1, "fast_lanes": 24}),
              ("fused, fast_lanes 40", {"pipeline": 1, "fast_lanes": 40})]
    reset = {"pipeline": -1, "pixel_refill": 0, "fast_lanes": 0}
    print("| scene | objects | triangles | box tests / ray | segments / path | frames per dispatch | choice | ms per 8-spp frame | auto picked | auto vs best |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name in names:
        scene, cam = SCENES[name]()
        r.upload_scene(scene)
        cnt = scene.counts()
        pc = cam(W, H, raysPerPixel=SPP, progressive=1, singleRender=0)
        r.reset_counters()
        r.render(pc, W, H); r.render(pc, W, H)   # the context measures the scene's ray cost on its first dispatches
        c0 = r.counters(); seg_per_path = c0["segments"] / max(c0["paths"], 1)
        for frames in (1, 10):
            res = {}
            for label, kv in forced:
                for k, v in {**reset, **kv}.items():
                    r.set_tuning(k, v)
                best = 1e9
                for rep in range(3):
                    pc.frameCount = 0
                    r.sync(); t = time.perf_counter()
                    if frames == 1:
                        r.render(pc, W, H, sync=False)
                    else:
                        r.render_frames(pc, W, H, frames, sync=False)
                    r.sync(); best = min(best, (time.perf_counter() - t) / frames * 1e3)
                res[label] = (best, ["multi-kernel", "fused"][r.last_pipeline()])
            for k, v in reset.items():
                r.set_tuning(k, v)
            fastest = min(v[0] for v in res.values())
            for label, (ms, pipe) in res.items():
                print(f"| {name} | {cnt['objects']} | {cnt['triangles']} | {r.ray_cost():.0f} | {seg_per_path:.2f} | {frames} | {label} | {ms:.2f} | {pipe if label == 'auto' else ''} | "
                      f"{'%+.1f %%' % ((ms / fastest - 1) * 100) if label == 'auto' else ''} |", flush=True)
        if "--phase-stats" in sys.argv:
            r.set_tuning("pipeline", 0); r.set_tuning("phase_stats", 1)
            r.reset_counters(); pc.frameCount = 0
            r.render_frames(pc, W, H, 4)
            r.counters()   # prints the phase statistics of k_trace_pw on stderr
            r.set_tuning("phase_stats", 0); r.set_tuning("pipeline", -1)


if __name__ == "__main__":
    main()
