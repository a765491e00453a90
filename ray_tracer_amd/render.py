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

Two deliberate deviations from draw(), both because this is a command that has
to end and the reference is a window that does not: (1) with singleRender off
the reference resets totalSamples to 0 after every frame
(src/vk_engine.cpp:1813-1814), so sampleLimit never stops it; here
totalSamples accumulates and the loop ends at sampleLimit. (2) With neither
--progressive nor --single-render every dispatch would be the identical
frameCount-0 frame (the reference re-renders it until the window closes); here
exactly one such frame is rendered.

Several GPUs of one node: launch one process per GPU,

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m ray_tracer_amd.render --scene sponza --single-render --sample-limit 1024 --out sponza.png

rank r renders rows r, r+N, ... of the frame (the same pixels, bit for bit, as a
single process), and one RCCL gather at the end brings the strips to rank 0,
which writes the file.
"""
import argparse
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the first HIP call: the renderer's three streams beside torch's and RCCL's (rt_amd.h, "lane_grid_pct")

from . import engine, scenes  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m ray_tracer_amd.render", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="cornell", choices=sorted(scenes.CONFIGS) + ["obj"])
    ap.add_argument("--obj", help="with --scene obj: an OBJ file placed in the default Cornell box")
    ap.add_argument("--obj-material", type=int, default=0, help="0 white 1 red 2 green 3 light 4 mirror 5 dielectric")
    ap.add_argument("--obj-scale", type=float, default=0.7)
    ap.add_argument("--obj-position", type=float, nargs=3, default=(0.0, 0.53, 0.0))
    ap.add_argument("--obj-rotation", type=float, nargs=3, default=(0.0, 0.0, 0.0))
    ap.add_argument("--obj-albedo-map", help="with --scene obj: an image bound as the albedo texture of the OBJ's materials "
                                             "(what an MTL's map_Kd line does; dread.mtl has none, the author bound dread_alb.png by hand)")
    ap.add_argument("--textures", action="store_true",
                    help="sample the maps the scene's MTL files name (and --obj-albedo-map). Off by default: the reference snapshot uploads "
                         "textures and never samples them (raytrace.comp declares TextureBuffer / TextureSampler and reads neither), so the "
                         "untextured image is the one that matches its shader; the sampling semantics are this build's declaration "
                         "(DESIGN.md 3a, parity unpinned)")
    ap.add_argument("--width", type=int, default=1728)
    ap.add_argument("--height", type=int, default=1117)
    ap.add_argument("--device", type=int, default=0, help="GPU of a single-process run (one process per GPU uses LOCAL_RANK)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="several processes: nccl (= RCCL, one GPU per rank); gloo lets all ranks share --device (rehearsal on one GPU)")
    ap.add_argument("--out", help="PNG (8-bit sRGB) or .npy (fp32 RGBA) output file")
    # Ray Tracer Info panel
    ap.add_argument("--progressive", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="progressive accumulation: frames handed to the GPU at once (rt_render_frames: the same image as one "
                         "dispatch per frame, sooner — more paths per dispatch; 1 = the reference's frame loop)")
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
        n0 = s.counts()["materials"]
        s.read_obj(args.obj, engine.placement(position=args.obj_position, scale=args.obj_scale, rotation=args.obj_rotation, samplerIndex=1), args.obj_material)
        if args.obj_albedo_map:
            slot = s.add_texture(args.obj_albedo_map)
            for mi in range(n0, s.counts()["materials"]) or [args.obj_material]:
                m = s.material(mi)
                m.albedoIndex = slot
                s.set_material(mi, m)
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


def srgb8(img):
    """Display encoding of the fp32 frame, the one rt_read_rgba8_srgb applies."""
    v = np.clip(np.nan_to_num(img, nan=0.0), 0.0, 1.0)
    rgb = np.where(v[..., :3] <= 0.0031308, 12.92 * v[..., :3], 1.055 * np.power(v[..., :3], 1.0 / 2.4) - 0.055)
    return (np.concatenate([rgb, v[..., 3:]], axis=-1) * 255.0 + 0.5).astype(np.uint8)


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = args.device
    if world > 1:
        import torch
        import torch.distributed as dist
        from . import tiling
        if args.backend == "nccl":
            tiling.prepare_rccl_env()
            device = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    scene, label = make_scene(args)
    pc = make_constants(args)
    r = engine.Renderer(device)
    r.upload_scene(scene)
    paths = scene.texture_paths()
    textured = False
    if args.textures and paths and all(os.path.exists(p) for p in paths):   # the MTL files' maps (src/vk_engine.cpp:1155); absent files: untextured
        r.upload_textures(engine.load_textures(scene))
        textured = True
    elif paths and rank == 0:
        print(f"{len(paths)} texture map(s) named by the scene are not sampled (the reference's shader samples none; --textures applies this build's declared semantics)")
    W, H = args.width, args.height
    tile = dict(row0=rank, rowStride=world) if world > 1 else {}
    t0 = time.perf_counter()
    frames = 0
    img = None
    while True:
        out = r.run_compute(pc, W, H, frames=args.frames_in_flight, **tile)
        if out is None:
            break
        img = out
        frames += 1
        if not args.progressive and not args.single_render:
            break  # every further dispatch would be this very frame again (frameCount does not advance)
    if world > 1 and img is not None:   # strips -> frame on rank 0
        on = f"cuda:{device}" if args.backend == "nccl" else "cpu"
        strip = torch.from_numpy(img).to(on)
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device=on) if rank == 0 else None
        tiling.gather_frame(strip, frame, H, world, rank)
        img = frame.cpu().numpy() if rank == 0 else None
    dt = time.perf_counter() - t0
    c = r.counters()
    if world > 1:
        tot = torch.tensor([float(c["raysReference"]), float(c["raysTraced"])], dtype=torch.float64,
                           device=f"cuda:{device}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tot)
        c = {"raysReference": float(tot[0]), "raysTraced": float(tot[1])}
    if rank == 0:
        print(f"{label}{' [textures sampled: declared semantics, parity unpinned]' if textured else ''}: {W}x{H}, {r.totalSamples} spp in {frames} dispatch(es)" + (f" on {world} GPUs" if world > 1 else "") +
              f", {dt:.3f} s, {c['raysReference'] / dt / 1e6:.0f} Mrays/s (reference semantics), {c['raysTraced'] / dt / 1e6:.0f} M executed rays/s")
        if args.out and img is not None:
            if args.out.endswith(".npy"):
                np.save(args.out, img)
            else:
                from PIL import Image
                Image.fromarray((srgb8(img) if world > 1 else r.read_rgba8_srgb())[..., :3]).save(args.out)
            print("wrote", args.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
