"""The command-line driver (ray_tracer_amd/render.py): panel fields -> PushConstants
(src/vk_engine.cpp:1503-1534), and the draw() dispatch loop on the GPU."""
import numpy as np
import pytest

from ray_tracer_amd import render


def test_panel_flags_reach_push_constants():
    args = render.build_parser().parse_args(
        "--width 320 --height 200 --progressive --rays-per-pixel 3 --bounce-limit 5 --triangle-cap 70 "
        "--box-cap 300 --sample-limit 12 --debug 1 --fov 50 --camera-position 1 2 3 --environment "
        "--sun-focus 20 --sun-intensity 4 --sun-direction 0 1 0".split())
    pc = render.make_constants(args)
    t = pc.rayTraceParams
    assert (t.progressive, t.singleRender, t.debug, t.raysPerPixel, t.bounceLimit, t.triangleCap, t.boxCap,
            t.sampleLimit) == (1, 0, 1, 3, 5, 70, 300, 12)
    assert pc.camInfo.fov == 50.0 and list(pc.camInfo.pos) == [1.0, 2.0, 3.0]
    assert np.float32(pc.camInfo.aspectRatio) == np.float32(320) / np.float32(200)
    assert list(pc.environment.lightDir) == [0.0, 1.0, 0.0, 1.0]
    assert pc.environment.horizonColor[3] == 20.0 and pc.environment.zenithColor[3] == 4.0


def test_defaults_are_the_reference_defaults():
    pc = render.make_constants(render.build_parser().parse_args([]))
    t = pc.rayTraceParams
    # src/vk_engine.h:160-171
    assert (t.progressive, t.singleRender, t.debug, t.raysPerPixel, t.bounceLimit, t.triangleCap, t.boxCap,
            t.sampleLimit) == (0, 0, -1, 1, 8, 50, 200, 10)
    assert pc.environment.lightDir[3] == 0.0 and pc.environment.horizonColor[3] == 1000.0


@pytest.mark.gpu
def test_cli_renders_cornell(tmp_path):
    out = tmp_path / "c.npy"
    assert render.main(f"--scene cornell --width 96 --height 64 --sample-limit 3 --progressive --out {out}".split()) == 0
    img = np.load(out)
    assert img.shape == (64, 96, 4) and np.isfinite(img).all() and img[..., :3].max() > 0
    png = tmp_path / "c.png"
    assert render.main(f"--scene cornell --width 96 --height 64 --single-render --sample-limit 2 --out {png}".split()) == 0
    assert png.stat().st_size > 100


@pytest.mark.gpu
def test_cli_frames_in_flight_write_the_same_frame(tmp_path):
    """--frames-in-flight: the progressive loop hands several frames to rt_render_frames at once (7 frames as 3 + 3 + 1):
    the same fp32 frame, bit for bit, as the reference's one dispatch per frame."""
    job = "--scene bunny --width 96 --height 64 --progressive --rays-per-pixel 2 --sample-limit 14"
    one, many = tmp_path / "one.npy", tmp_path / "many.npy"
    assert render.main(f"{job} --out {one}".split()) == 0
    assert render.main(f"{job} --frames-in-flight 3 --out {many}".split()) == 0
    assert np.array_equal(np.load(one).view(np.uint32), np.load(many).view(np.uint32))


@pytest.mark.gpu
def test_cli_on_two_ranks_writes_the_single_process_frame(tmp_path):
    """python -m torch.distributed.run ... -m ray_tracer_amd.render: two ranks (gloo, both on this box's one GPU) render
    the interleaved rows of a progressive three-dispatch job, rank 0 stitches and writes: the same fp32 frame, bit for
    bit, as one process, and the same PNG."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    job = "--scene cornell --width 96 --height 63 --progressive --rays-per-pixel 2 --sample-limit 6"
    one, two = tmp_path / "one.npy", tmp_path / "two.npy"
    assert render.main(f"{job} --out {one}".split()) == 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", "-m", "ray_tracer_amd.render", *job.split(), "--backend", "gloo", "--out", str(two)],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "on 2 GPUs" in p.stdout
    assert np.array_equal(np.load(one).view(np.uint32), np.load(two).view(np.uint32))
    png1, png2 = tmp_path / "one.png", tmp_path / "two.png"
    assert render.main(f"{job} --out {png1}".split()) == 0
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29548", "-m", "ray_tracer_amd.render", *job.split(), "--backend", "gloo", "--out", str(png2)],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    from PIL import Image
    a, b = np.asarray(Image.open(png1)).astype(int), np.asarray(Image.open(png2)).astype(int)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1   # numpy pow against the library's rt_pow in the display encoding
