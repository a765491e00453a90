"""BASELINE.json's five configurations at their real triangle counts and resolutions, HIP path against the CPU oracle.

The oracle cannot render a 1080p / 4K frame of an 871 k-triangle scene in test time, but it does not have to: a pixel
depends only on its global index and the scene (raytrace.comp:563-564), so image rows are independent. Each config is
therefore compared on rows spread over the frame, at the frame's own size and camera:

  * the whole frame is dispatched once on the GPU (automatic pipeline choice, i.e. what bench.py runs) at >= 4 samples per
    pixel, and its sampled rows must equal the oracle's bit for bit;
  * the same rows are rendered as a tile in each of the three pipeline modes (multi-kernel, fused block-at-a-time, fused
    with pixel refill) — pixels and all seven counters against the oracle;
  * two rows run a 64-sample chain (the RNG state is carried serially through a pixel's samples).

C1 (Cornell + 3 spheres, 512x512, 4 spp) is small enough for the oracle to render whole.
The real bunny / dragon / Sponza files are not part of the reference snapshot: the seeded stand-ins with the
same triangle counts are used, as everywhere else, unless the files are present under assets/.
"""
import functools

import numpy as np
import pytest

from oracle import pyoracle
from ray_tracer_amd import engine, scenes

pytestmark = pytest.mark.gpu

MODES = [("multikernel", 0, 0), ("fused", 1, 64), ("fused-refill", 1, 8)]
KEYS = ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests")

# name -> (scene factory key, width, height, triangles the stand-in must have, rows of the frame that are compared)
CONFIGS = {
    "C2_bunny_1080p": ("bunny", 1920, 1080, 51 + scenes.BUNNY_TRIS),      # the default Cornell scene has 51 triangles
    "C3_dragon_1080p": ("dragon", 1920, 1080, 51 + scenes.DRAGON_TRIS),
    "C4_sponza_1080p": ("sponza", 1920, 1080, scenes.SPONZA_TRIS + 10),   # + the emitter light2.obj (10 triangles)
    "C5_sponza_dragons_4k": ("sponza_dragons", 3840, 2160, scenes.SPONZA_TRIS + 10 + scenes.DRAGON_TRIS),
    "C5_sponza_dragons_flat_4k": ("sponza_dragons_flat", 3840, 2160, scenes.SPONZA_TRIS + 10 + 16 * scenes.DRAGON_TRIS),
}


@functools.lru_cache(maxsize=1)   # one big scene in host memory at a time
def _scene(key):
    return scenes.CONFIGS[key]()


def _cam(key):
    return scenes.sponza_camera if key.startswith("sponza") else engine.push_constants


def _set_mode(r, pipeline, refill):
    r.set_tuning("pipeline", pipeline)
    r.set_tuning("pixel_refill", refill)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_c1_cornell_512_whole_frame(renderer):
    """C1: Cornell + dielectric / mirror / diffuse spheres, 512x512, 4 spp — every pixel and every counter."""
    s, _ = scenes.cornell(True)
    W = H = 512
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=4)
    ref, rc = pyoracle.render(s, pc, W, H)
    renderer.upload_scene(s)
    try:
        for name, pipe, refill in MODES + [("auto", -1, 0)]:
            _set_mode(renderer, pipe, refill)
            renderer.reset_counters()
            img = renderer.render(pc, W, H)
            cnt = renderer.counters()
            assert np.array_equal(_bits(img), _bits(ref)), f"{name}: pixels differ from the oracle"
            assert {k: cnt[k] for k in KEYS} == {k: rc[k] for k in KEYS}, name
    finally:
        _set_mode(renderer, -1, 0)


@pytest.mark.parametrize("config", list(CONFIGS))
def test_config_at_full_size_on_sampled_rows(renderer, config):
    key, W, H, tris = CONFIGS[config]
    s, label = _scene(key)
    if label.startswith("synthetic"):
        assert s.counts()["triangles"] == tris, "the stand-in must have the configuration's triangle count"
    cam = _cam(key)
    r = renderer
    r.upload_scene(s)
    try:
        # ---- 4 spp: eight rows spread over the frame
        spp = 4
        pc = cam(W, H, singleRender=1, sampleLimit=spp)
        tile = dict(row0=H // 16 + 3, rowStride=H // 8, nRows=8)
        rows = slice(tile["row0"], None, tile["rowStride"])
        ref, rc = pyoracle.render(s, pc, W, H, **tile)
        assert rc["stackOverflow"] == 0 and rc["lightQueryMismatch"] == 0
        _set_mode(r, -1, 0)
        full = r.render(pc, W, H)             # the whole frame, as bench.py dispatches it
        assert full.shape == (H, W, 4)
        assert np.array_equal(_bits(full[rows][:8]), _bits(ref)), "full-frame dispatch differs from the oracle on the sampled rows"
        for name, pipe, refill in MODES:
            _set_mode(r, pipe, refill)
            r.reset_counters()
            img = r.render(pc, W, H, **tile)
            cnt = r.counters()
            assert np.array_equal(_bits(img), _bits(ref)), f"{name}: tile pixels differ from the oracle"
            assert {k: cnt[k] for k in KEYS} == {k: rc[k] for k in KEYS}, name
        # ---- 64 serial samples on two rows
        pc = cam(W, H, singleRender=1, sampleLimit=64)
        tile = dict(row0=H // 3, rowStride=H // 3, nRows=2)
        ref, rc = pyoracle.render(s, pc, W, H, **tile)
        for name, pipe, refill in MODES:
            _set_mode(r, pipe, refill)
            r.reset_counters()
            img = r.render(pc, W, H, **tile)
            cnt = r.counters()
            assert np.array_equal(_bits(img), _bits(ref)), f"{name}: 64-spp rows differ from the oracle"
            assert {k: cnt[k] for k in KEYS} == {k: rc[k] for k in KEYS}, name
    finally:
        _set_mode(r, -1, 0)
