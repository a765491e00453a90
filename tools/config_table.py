"""BASELINE.json's configurations on one MI355X at their quoted samples per pixel: Mrays/s in the reference's ray
accounting, executed rays, samples/pixel/s and the algorithmic-bytes fraction of the 8 TB/s roofline, for the whole frame
and for rank 0's interleaved rows of 2 / 4 / 8 GPUs (what one of N GPUs renders).  Markdown on stdout.
usage: config_table.py [--quick]   (--quick: an eighth of the samples)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes  # noqa: E402

quick = "--quick" in sys.argv
CONFIGS = [  # name, scene, width, height, spp, spp per dispatch
    ("C1 Cornell + 3 spheres", "cornell", 512, 512, 4, 4),
    ("C2 Cornell + bunny", "bunny", 1920, 1080, 64, 8),          # 8 spp per dispatch: bench.py's step (the reference's default is 1 per frame,
    ("C3 Cornell + dragon (mirror)", "dragon", 1920, 1080, 256, 8),   # src/vk_engine.h:164, accumulated progressively up to sampleLimit)
    ("C4 Sponza", "sponza", 1920, 1080, 1024, 8),
    ("C5 Sponza + 16 dragons, 4K (128 of 4096 spp)", "sponza_dragons", 3840, 2160, 128, 8),
]
r = engine.Renderer(0)
print("| config | GPUs (rows of rank 0) | spp | time | Mrays/s (reference accounting) | executed Mrays/s | spp/s | algorithmic GB/s | of 8 TB/s | pipeline |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, key, W, H, spp, per in CONFIGS:
    scene, label = scenes.CONFIGS[key]()
    cam = scenes.sponza_camera if key.startswith("sponza") else engine.push_constants
    if quick:
        spp = max(per if per < 8 else 8, spp // 8)
        per = min(per, spp)
    r.upload_scene(scene)
    pcw = cam(W, H, raysPerPixel=1, progressive=1, singleRender=0)
    r.render(pcw, W, H)   # the first dispatch of a scene measures its rays (launch parameters follow the ray length)
    r.render(pcw, W, H)
    for world in (1, 2, 4, 8):
        pc = cam(W, H, raysPerPixel=per, progressive=1, singleRender=0)
        n = spp // per
        fif = 10 * world                     # as bench.py: at most 10 N dispatches (progressive frames) in flight on one of N GPUs
        ng = (n + fif - 1) // fif            # (rt_render_frames), in even groups
        sizes = [n // ng + (1 if j < n % ng else 0) for j in range(ng)]
        for timed in (False, True):          # the first group of a new shape allocates its path state: one untimed group first
            r.reset_counters()
            r.sync()
            t = time.perf_counter()
            i = 0
            for k in (sizes if timed else sizes[:1]):
                pc.frameCount = i
                if k == 1:
                    r.render(pc, W, H, row0=0, rowStride=world, sync=False)
                else:
                    r.render_frames(pc, W, H, k, row0=0, rowStride=world, sync=False)
                i += k
            r.sync()
            dt = time.perf_counter() - t
        c = r.counters()
        alg = 32.0 * c["boxTests"] + 36.0 * c["triTests"] + 100.0 * c["raysHit"]
        print(f"| {name} ({label}) {W}x{H} | {world} | {spp} | {dt * 1e3:.1f} ms | {c['raysReference'] / dt / 1e6:.0f} | {c['raysTraced'] / dt / 1e6:.0f} | "
              f"{spp / dt:.1f} | {alg / dt / 1e9:.0f} | {alg / dt / 8e12 * 100:.0f} % | {['multi-kernel', 'fused'][r.last_pipeline()]} |", flush=True)
