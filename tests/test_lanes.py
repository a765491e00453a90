"""The multi-kernel pipeline in several overlapping parts ("lanes"): contiguous ranges of path slots, each with its own queues,
counters and HIP stream, forked from and joined to the ctx stream. A pixel's path depends on nothing but its slot, so any split
must give the bits of the whole — pixels, counters, heat maps, several frames per dispatch, tiles whose size no part boundary
divides. The parity tests' images are too small to be split by default (parts start at 1 M paths), so the split is forced here
("lanes_min_kslots" 1: from 1024 paths)."""
import numpy as np
import pytest

from oracle import pyoracle
from ray_tracer_amd import engine, scenes

from util import cornell_scene, model_scene

pytestmark = pytest.mark.gpu
KEYS = ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests")


@pytest.fixture
def forced_parts(renderer):
    renderer.set_tuning("pipeline", 0)
    renderer.set_tuning("lanes_min_kslots", 1)
    yield renderer
    for k, v in (("pipeline", -1), ("lanes_min_kslots", 1024), ("lanes", 0), ("lane_grid_pct", 0), ("lds_stack", 24)):
        renderer.set_tuning(k, v)


@pytest.mark.parametrize("lanes,pct", [(1, 100), (2, 100), (2, 60), (3, 50), (3, 20), (4, 30)])
def test_parts_give_the_whole(forced_parts, lanes, pct):
    r = forced_parts
    r.set_tuning("lanes", lanes)
    r.set_tuning("lane_grid_pct", pct)
    cases = [(cornell_scene(True), 97, 61, dict(singleRender=1, sampleLimit=3)),                       # 5917 slots: no boundary divides it
             (model_scene("bunny.obj", material=5, spheres=True), 128, 96, dict(singleRender=1, sampleLimit=2)),
             (model_scene("klein_bottle.obj", material=4, scale=0.5, position=(0.0, -0.2, 0.0)), 96, 64, dict(singleRender=1, sampleLimit=2, debug=2, boxCap=300, triangleCap=60))]
    for s, W, H, kw in cases:
        pc = engine.push_constants(W, H, **kw)
        ref, rc = pyoracle.render(s, pc, W, H)
        r.upload_scene(s)
        r.reset_counters()
        img = r.render(pc, W, H)
        assert r.last_pipeline() == 0 and r.last_parts() == lanes
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"lanes {lanes}, {pct} %: pixels differ"
        c = r.counters()
        for k in KEYS:
            assert c[k] == rc[k], f"lanes {lanes}: counter {k}: gpu {c[k]} oracle {rc[k]}"


def test_parts_with_several_frames_per_dispatch_and_an_overflow_stack(forced_parts):
    """rt_render_frames through parts: the slots of a tile block's frames stay in one part; progressive blend in frame order; the
    overflow stack (LDS part cut to 8 entries) has a buffer per part, since the parts' launches run at the same time."""
    r = forced_parts
    s = model_scene("klein_bottle.obj", material=0, scale=0.5, position=(0.0, -0.2, 0.0), spheres=True)
    W, H, frames = 100, 75, 5
    pc = engine.push_constants(W, H, raysPerPixel=2, progressive=1)
    prev, ref = None, None
    tot = {k: 0 for k in KEYS}
    for f in range(frames):
        pc.frameCount = f
        ref, rc = pyoracle.render(s, pc, W, H, prev=prev)
        prev = ref
        for k in KEYS:
            tot[k] += rc[k]
    r.upload_scene(s)
    for lanes, cap in ((3, 24), (2, 8), (4, 8)):
        r.set_tuning("lanes", lanes); r.set_tuning("lds_stack", cap)
        r.clear_framebuffer(); r.reset_counters()
        pc.frameCount = 0
        img = r.render_frames(pc, W, H, frames)
        assert r.last_parts() == lanes
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"lanes {lanes}, stack cap {cap}"
        c = r.counters()
        for k in KEYS:
            assert c[k] == tot[k], (lanes, cap, k)


def test_parts_on_rows_of_a_tile(forced_parts):
    """rank 1's rows of three GPUs, through parts: the slot -> pixel map of a strided tile is the same in every part."""
    r = forced_parts
    s, _ = scenes.sponza(0, ntris=20000)
    W, H = 256, 135
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=2)
    tile = dict(row0=1, rowStride=3, nRows=(H - 1 + 2) // 3)
    ref, rc = pyoracle.render(s, pc, W, H, **tile)
    r.upload_scene(s)
    for lanes in (2, 3):
        r.set_tuning("lanes", lanes)
        r.reset_counters()
        img = r.render(pc, W, H, **tile)
        assert r.last_parts() == lanes
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        assert r.counters()["boxTests"] == rc["boxTests"]
