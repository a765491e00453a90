"""Command-line driver: the place of the reference's `main.cpp` + ImGui panels.

    python -m ray_tracer_amd.render --scene cornell --width 1728 --height 1117 \
        --single-render --sample-limit 100 --out cornell.png

Every tunable of the reference's "Ray Tracer Info", "Camera Info" and
"Environment" panels (src/vk_engine.cpp:1503-1534) is a flag with the
reference's default (src/vk_engine.h:145-171,325). The frame loop follows
draw() (src/vk_engine.cpp:1774-1815): dispatches run while
totalSamples < sampleLimit; a single render is one dispatch of sampleLimit
samples per pixel, otherwise each dispatch adds raysPerPixel samples and, when
progressive, is blended into the fp32 frame with weight 1/(frame+1).
"""
import argparse
import sys
import time

import numpy as np

from . import engine, scenes


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m ray_tracer_amd.render", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="cornell", choices=sorted(scenes.CONFIGS) + ["obj"])
    ap.add_argument("--obj", help="with --scene obj: an OBJ file placed in the default Cornell box")
    ap.add_argument("--obj-material", type=int, default=0, help="0 white 1 red 2 green 3 light 4 mirror 5 dielectric")
    ap.add_argument("--obj-scale", type=float, default=0.7)
    ap.add_argument("--obj-position", type=float, nargs=3, default=(0.0, 0.53, 0.0))
    ap.add_argument("--width", type=int, default=1728)
    ap.add_argument("--height", type=int, default=1117)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", help="PNG (8-bit sRGB) or .npy (fp32 RGBA) output file")
    # Ray Tracer Info panel
    ap.add_argument("--progressive", action="store_true")
    ap.add_argument("--single-render", action="store_true")
    ap.add_argument("--debug", type=int, default=-1, choices=[-1, 0, 1, 2])
    ap.add_argument("--rays-per-pixel", type=int, default=1)
    ap.add_argument("--bounce-limit", type=int, default=8)
    ap.add_argument("--triangle-cap", type=int, default=50)
    ap.add_argument("--box-cap", type=int, default=200)
    ap.add_argument("--sample-limit", type=int, default=10)
    # Camera Info panel
    ap.add_argument("--fov", type=float, default=None)
    ap.add_argument("--camera-angles", type=float, nargs=3, default=None, metavar=("X", "Y", "Z"))
    ap.add_argument("--camera-position", type=float, nargs=3, default=None, metavar=("X", "Y", "Z"))
    # Environment panel
    ap.add_argument("--environment", action="store_true", help="environment lighting on (lightDir.w = 1)")
    ap.add_argument("--sun-direction", type=float, nargs=3, default=None)
    ap.add_argument("--sun-focus", type=float, default=None)
    ap.add_argument("--sun-intensity", type=float, default=None)
    return ap


def make_scene(args):
    if args.scene == "obj":
        if not args.obj:
            raise SystemExit("--scene obj needs --obj FILE")
        s = engine.Scene()
        s.prepare_storage_buffers()
        s.read_obj(args.obj, engine.placement(position=args.obj_position, scale=args.obj_scale, samplerIndex=1), args.obj_material)
        return s, args.obj
    return scenes.CONFIGS[args.scene]()


def make_constants(args):
    W, H = args.width, args.height
    kw = dict(progressive=int(args.progressive), singleRender=int(args.single_render), debug=args.debug,
              raysPerPixel=args.rays_per_pixel, bounceLimit=args.bounce_limit, triangleCap=args.triangle_cap,
              boxCap=args.box_cap, sampleLimit=args.sample_limit)
    cam = scenes.sponza_camera if args.scene.startswith("sponza") else engine.push_constants
    if args.fov is not None:
        kw["fov"] = args.fov
    if args.camera_angles is not None:
        kw["cameraAngles"] = args.camera_angles
    if args.camera_position is not None:
        kw["pos"] = args.camera_position
    if args.environment:
        kw["environmentOn"] = True
    pc = cam(W, H, **kw)
    if args.sun_direction is not None:
        d = np.asarray(args.sun_direction, np.float32)
        pc.environment.lightDir[0:3] = [float(x) for x in d]
    if args.sun_focus is not None:
        pc.environment.horizonColor[3] = args.sun_focus
    if args.sun_intensity is not None:
        pc.environment.zenithColor[3] = args.sun_intensity
    return pc


def main(argv=None):
    args = build_parser().parse_args(argv)
    scene, label = make_scene(args)
    pc = make_constants(args)
    r = engine.Renderer(args.device)
    r.upload_scene(scene)
    W, H = args.width, args.height
    t0 = time.perf_counter()
    frames = 0
    img = None
    while True:
        out = r.run_compute(pc, W, H)
        if out is None:
            break
        img = out
        frames += 1
    dt = time.perf_counter() - t0
    c = r.counters()
    print(f"{label}: {W}x{H}, {r.totalSamples} spp in {frames} dispatch(es), {dt:.3f} s, "
          f"{c['raysReference'] / dt / 1e6:.0f} Mrays/s (reference semantics), {c['raysTraced'] / dt / 1e6:.0f} M executed rays/s")
    if args.out and img is not None:
        if args.out.endswith(".npy"):
            np.save(args.out, img)
        else:
            from PIL import Image
            Image.fromarray(r.read_rgba8_srgb()[..., :3]).save(args.out)
        print("wrote", args.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
