"""One single-render dispatch (the reference's singleRender mode: all samples in one run_compute) of a scene that the
context has never rendered before, with and without the ray-cost probe.  usage: single_render_timing.py [scene] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, H = 1920, 1080
scene, label = scenes.CONFIGS[name]()
cam = scenes.sponza_camera if name.startswith("sponza") else engine.push_constants
pc = cam(W, H, singleRender=1, sampleLimit=spp)
r = engine.Renderer(0)
for probe in (1, 0, 1, 0):
    r.set_tuning("probe", probe)
    r.upload_scene(scene)
    r.reset_counters()
    t = time.perf_counter()
    r.render(pc, W, H)
    dt = time.perf_counter() - t
    c = r.counters()
    print(f"{label} {W}x{H} {spp} spp single render, probe={probe}: {dt * 1e3:8.1f} ms  {c['raysReference'] / dt / 1e6:7.1f} Mrays/s  "
          f"pipeline {r.last_pipeline()}  box tests per ray {r.ray_cost():.1f}", flush=True)
