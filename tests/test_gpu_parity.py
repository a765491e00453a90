"""GPU parity tests: the HIP wavefront pipeline (through the C ABI) against the
scalar CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): pixels within 1e-4 relative fp32. Because both
sides evaluate the same IEEE operations in the same order (rt_det_math.h), the
tests ask for more: bit-identical pixels, hit records and box/triangle
counters. The 1e-4 tolerance is asserted first so a failure says which bar
broke."""
import numpy as np
import pytest

from oracle import pyoracle
from ray_tracer_amd import engine, scenes

from util import assert_hits_equal, cornell_scene, model_scene, seeded_rays

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # the north_star tolerance


@pytest.fixture(params=[(0, 0), (1, 64), (1, 8)], ids=["multikernel", "fused", "fused-refill"], autouse=True)
def pipeline(request, renderer):
    """Every test runs on both pipelines: 0 = k_raygen / k_trace_pw / k_shade / k_resolve launched per round,
    1 = k_render_fused (each wave runs the same stages on its own pixels) — a block of pixels at a time (what it does
    for scenes with short rays) and with finished pixels replaced as soon as eight lanes are free (long rays)."""
    renderer.set_tuning("pipeline", request.param[0])
    renderer.set_tuning("pixel_refill", request.param[1])
    yield request.param[0]
    renderer.set_tuning("pipeline", -1)
    renderer.set_tuning("pixel_refill", 0)


def _render_both(r, scene, pc, W, H, **tile):
    r.upload_scene(scene)
    r.reset_counters()
    img = r.render(pc, W, H, **tile)
    cnt = r.counters()
    ref, rc = pyoracle.render(scene, pc, W, H, **tile)
    return img, cnt, ref, rc


def _check(img, cnt, ref, rc):
    assert img.shape == ref.shape
    assert np.allclose(img, ref, rtol=RTOL, atol=1e-7, equal_nan=True), \
        f"{int((~np.isclose(img, ref, rtol=RTOL, atol=1e-7)).sum())} values beyond 1e-4"
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "pixels not bit-identical"
    for k in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests"):
        assert cnt[k] == rc[k], f"counter {k}: gpu {cnt[k]} oracle {rc[k]}"
    assert rc["stackOverflow"] == 0
    assert rc["lightQueryMismatch"] == 0, "a light query answered from the emitter list disagrees with the shader's closest hit"


def test_device_selftest(renderer):
    assert renderer.selftest() == 0x0F


def test_cornell_c1_config(renderer):
    """C1: Cornell + dielectric/mirror/diffuse spheres, 4 spp (reduced to 160x160 for the oracle)."""
    s = cornell_scene(True)
    W = H = 160
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=4)
    _check(*_render_both(renderer, s, pc, W, H))


def test_cornell_frame_seeds_and_rays_per_pixel(renderer):
    s = cornell_scene(False)
    W, H = 96, 64
    for frame in (0, 1, 7):
        pc = engine.push_constants(W, H, raysPerPixel=3, frameCount=frame)
        _check(*_render_both(renderer, s, pc, W, H))


@pytest.mark.parametrize("bounce", [0, 1, 5, 8, 13])
def test_bounce_limits(renderer, bounce):
    s = cornell_scene(True)
    W, H = 80, 60
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=bounce)
    _check(*_render_both(renderer, s, pc, W, H))


def test_bunny_diffuse_and_mirror(renderer):
    for mat in (0, 4, 5):
        s = model_scene("bunny.obj", material=mat)
        W, H = 128, 96
        pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
        _check(*_render_both(renderer, s, pc, W, H))


def test_klein_bottle_35k_tris(renderer):
    s = model_scene("klein_bottle.obj", material=4, scale=0.5, position=(0.0, -0.2, 0.0), spheres=True)
    W, H = 128, 96
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    _check(*_render_both(renderer, s, pc, W, H))


def test_multi_material_obj_with_mtl(renderer):
    import os
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.read_obj(os.path.join(engine.ASSET_DIR, "bobadog", "bobadog.obj"),
               engine.placement(position=(0, 0.3, 0), scale=0.35, rotation=(0, 160, 0)), 0)
    W, H = 96, 96
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    _check(*_render_both(renderer, s, pc, W, H))


def test_environment_light_and_camera(renderer):
    s = cornell_scene(True)
    W, H = 96, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, environmentOn=True, cameraAngles=(-12.0, 25.0, 5.0),
                               pos=(0.8, -0.9, -3.0), fov=80.0)
    _check(*_render_both(renderer, s, pc, W, H))


def test_sponza_standin_small(renderer):
    s, label = scenes.sponza(0, ntris=20000)
    assert label == "synthetic-20000-tris"
    W, H = 96, 54
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=2)
    _check(*_render_both(renderer, s, pc, W, H))


def test_tiling_is_invisible(renderer):
    """Interleaved row strips (the multi-GPU partition) reproduce the full frame bit for bit."""
    s = cornell_scene(True)
    W, H = 64, 50
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    renderer.upload_scene(s)
    full = renderer.render(pc, W, H)
    for world in (2, 3, 8):
        out = np.zeros_like(full)
        for rank in range(world):
            n = len(range(rank, H, world))
            out[rank::world] = renderer.render(pc, W, H, row0=rank, rowStride=world, nRows=n)
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
    # and the oracle agrees on a strip
    ref, _ = pyoracle.render(s, pc, W, H, row0=1, rowStride=3, nRows=len(range(1, H, 3)))
    assert np.array_equal(ref.view(np.uint32), full[1::3].view(np.uint32))


def test_debug_heatmaps_and_progressive(renderer):
    s = model_scene("bunny.obj")
    W, H = 64, 48
    for dbg in (0, 1, 2):
        pc = engine.push_constants(W, H, raysPerPixel=1, debug=dbg, boxCap=60, triangleCap=20)
        _check(*_render_both(renderer, s, pc, W, H))
    # progressive: frame k blends 1/(k+1) into the fp32 buffer
    renderer.upload_scene(s)
    prev = None
    for k in range(3):
        pc = engine.push_constants(W, H, raysPerPixel=2, progressive=1, frameCount=k)
        img = renderer.render(pc, W, H)
        ref, _ = pyoracle.render(s, pc, W, H, prev=prev)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        prev = ref


def test_zero_samples_is_magenta(renderer):
    s = cornell_scene(False)
    W, H = 16, 8
    pc = engine.push_constants(W, H, raysPerPixel=0)
    _check(*_render_both(renderer, s, pc, W, H))


@pytest.mark.parametrize("name", ["cornell", "bunny", "klein"])
def test_trace_rays_hit_records(renderer, name):
    """calculateIntersections per ray: t, object, triangle, frontFace, point, normal and counters."""
    s = {"cornell": lambda: cornell_scene(True), "bunny": lambda: model_scene("bunny.obj"),
         "klein": lambda: model_scene("klein_bottle.obj", scale=0.5, position=(0, -0.2, 0))}[name]()
    o, d = seeded_rays(4096, seed={"cornell": 101, "bunny": 202, "klein": 303}[name])  # fixed: hash(str) varies with PYTHONHASHSEED
    renderer.upload_scene(s)
    g = engine.hits_to_numpy(renderer.trace_rays(o, d))
    c = engine.hits_to_numpy(pyoracle.trace_rays(s, o, d))
    assert c["didHit"].sum() > 1000
    assert_hits_equal(g, c)


def test_update_buffers(renderer):
    """update_buffer semantics: edit materials / spheres / object transforms in place."""
    s = cornell_scene(True)
    W, H = 64, 48
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    renderer.upload_scene(s)
    a = renderer.render(pc, W, H)
    s.set_sphere(1, (0.5, 0.1, 0.0), 0.4, 1)   # mirror sphere becomes red diffuse
    renderer.update_spheres(s)
    b = renderer.render(pc, W, H)
    ref, _ = pyoracle.render(s, pc, W, H)
    assert not np.array_equal(a, b)
    assert np.array_equal(b.view(np.uint32), ref.view(np.uint32))


def test_error_paths(built):
    r = engine.Renderer(0)
    pc = engine.push_constants(8, 8)
    with pytest.raises(engine.RtError):
        r._counts = {"spheres": 0, "objects": 0}
        r.render(pc, 8, 8)  # before upload
    with pytest.raises(engine.RtError):
        engine.Renderer(4096)  # no such device
    r.close()


def test_full_size_properties_1080p(renderer):
    """BASELINE-size frame (1920x1080): size-independent properties instead of a CPU oracle pass:
    (1) two interleaved half-frames stitch to the full frame bit for bit;
    (2) counters obey the shader's accounting: raysReference = segments + 3 * diffuse bounces and
        raysTraced <= raysReference; paths = pixels * spp;
    (3) every k-th row agrees with the oracle bit for bit."""
    s, _ = scenes.sponza(0, ntris=60000)
    W, H = 1920, 1080
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=1)
    renderer.upload_scene(s)
    renderer.reset_counters()
    full = renderer.render(pc, W, H)
    cnt = renderer.counters()
    assert cnt["paths"] == W * H
    assert cnt["raysTraced"] <= cnt["raysReference"]
    assert (cnt["raysReference"] - cnt["segments"]) % 3 == 0
    assert np.isfinite(full).all() and (full[..., 3] == 1).all()
    halves = np.zeros_like(full)
    for rank in range(2):
        halves[rank::2] = renderer.render(pc, W, H, row0=rank, rowStride=2, nRows=H // 2)
    assert np.array_equal(halves.view(np.uint32), full.view(np.uint32))
    ref, _ = pyoracle.render(s, pc, W, H, row0=7, rowStride=135, nRows=8)
    assert np.array_equal(ref.view(np.uint32), full[7::135].view(np.uint32))


def test_gpu_against_committed_golden_fixtures(renderer):
    """The HIP path against tests/golden without running the oracle."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(g, "cornell_c1_64x64_4spp.npz"), allow_pickle=False)
    renderer.upload_scene(cornell_scene(True))
    renderer.reset_counters()
    img = renderer.render(engine.push_constants(64, 64, singleRender=1, sampleLimit=4), 64, 64)
    assert np.allclose(img, z["rgba"], rtol=RTOL, atol=1e-7)
    assert np.array_equal(img.view(np.uint32), z["rgba"].view(np.uint32))
    cnt = renderer.counters()
    gold = dict(zip([str(k) for k in z["counter_names"]], [int(v) for v in z["counters"]]))
    for k in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments"):
        assert cnt[k] == gold[k], k
    z = np.load(os.path.join(g, "bunny908_64x48_2spp_frame3.npz"), allow_pickle=False)
    renderer.upload_scene(model_scene("bunny.obj"))
    img = renderer.render(engine.push_constants(64, 48, raysPerPixel=2, frameCount=3), 64, 48)
    assert np.array_equal(img.view(np.uint32), z["rgba"].view(np.uint32))
    for name, sc in (("cornell", cornell_scene(True)), ("bunny908", model_scene("bunny.obj"))):
        z = np.load(os.path.join(g, f"hits_{name}_1024.npz"), allow_pickle=False)
        renderer.upload_scene(sc)
        h = engine.hits_to_numpy(renderer.trace_rays(z["origins"], z["dirs"]))
        assert_hits_equal(h, {k: z[k] for k in h})


def test_both_traversal_kernels_and_knobs_agree(renderer):
    """k_trace (one ray per lane) and k_trace_pw (persistent waves) under several knob settings:
    same pixels, same counters (the knobs are performance-only)."""
    s = model_scene("bunny.obj", material=5, spheres=True)
    W, H = 100, 50   # width not a multiple of 8: row-major slot order; 50 rows: 48 tiled + 2
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    ref, rc = pyoracle.render(s, pc, W, H)
    renderer.upload_scene(s)
    try:
        for knobs in ({"trace_variant": 0}, {"trace_variant": 1}, {"trace_variant": 1, "refill": 1, "fast_lanes": 65},
                      {"trace_variant": 1, "refill": 64, "chunk": 16, "w_setup": 1, "w_leaf": 64},
                      {"trace_variant": 1, "lds_stack": 8, "tile_slots": 0, "blocks_per_cu": 1}):
            for k, v in knobs.items():
                renderer.set_tuning(k, v)
            renderer.reset_counters()
            img = renderer.render(pc, W, H)
            _check(img, renderer.counters(), ref, rc)
    finally:
        for k, v in {"trace_variant": 1, "refill": 8, "mk_refill": 16, "fast_lanes": 24, "chunk": 256, "w_setup": 32, "w_leaf": 8,
                     "lds_stack": 24, "tile_slots": 1, "blocks_per_cu": 0}.items():
            renderer.set_tuning(k, v)
    with pytest.raises(engine.RtError):
        renderer.set_tuning("no_such_knob", 1)


def test_top_level_pairs_from_lds(renderer):
    """k_trace_pw<HOT>: the child pairs of the meshes' top levels come first in the device numbering and are served from a copy
    in LDS — as many as fit beside the stacks of six work-groups per CU ("hot_pairs" 1), of five (2, the default), or not at
    all (0). Scenes whose BVH depths select each stack size (8, 16, 20, 24 entries) and the overflow stack, which has no table."""
    cases = [(cornell_scene(True), 64, 48),
             (model_scene("bunny.obj", material=5, spheres=True), 96, 64),
             (model_scene("klein_bottle.obj", material=4, scale=0.5, position=(0.0, -0.2, 0.0)), 96, 64)]
    sp, _ = scenes.sponza(0, ntris=60000)
    try:
        renderer.set_tuning("pipeline", 0)
        for s, W, H in cases:
            pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
            ref, rc = pyoracle.render(s, pc, W, H)
            renderer.upload_scene(s)
            for hot in (0, 1, 2):
                renderer.set_tuning("hot_pairs", hot)
                renderer.reset_counters()
                _check(renderer.render(pc, W, H), renderer.counters(), ref, rc)
        W, H = 1920, 1080
        pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=2)
        tile = dict(row0=101, rowStride=135, nRows=8)
        ref, rc = pyoracle.render(sp, pc, W, H, **tile)
        renderer.upload_scene(sp)
        for hot, cap in ((0, 24), (1, 24), (2, 24), (2, 8)):   # cap 8: the overflow stack, no table
            renderer.set_tuning("hot_pairs", hot)
            renderer.set_tuning("lds_stack", cap)
            renderer.reset_counters()
            _check(renderer.render(pc, W, H, **tile), renderer.counters(), ref, rc)
    finally:
        renderer.set_tuning("hot_pairs", 2)
        renderer.set_tuning("lds_stack", 24)
        renderer.set_tuning("pipeline", -1)


def test_overflow_stack_beyond_the_lds_part(renderer):
    """With the LDS part of the traversal stack cut to 8 entries, the klein bottle's BVH (depth ~20)
    and the bunny's keep spilling into the global overflow buffer: pixels and counters must not move."""
    for name, kw in (("klein_bottle.obj", dict(material=4, scale=0.5, position=(0.0, -0.2, 0.0))), ("bunny.obj", dict(material=0))):
        s = model_scene(name, **kw)
        assert s.last_bvh_stats()["maxDepth"] > 8
        W, H = 96, 64
        pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
        ref, rc = pyoracle.render(s, pc, W, H)
        renderer.upload_scene(s)
        try:
            for cap in (8, 16, 24):
                renderer.set_tuning("lds_stack", cap)
                renderer.reset_counters()
                _check(renderer.render(pc, W, H), renderer.counters(), ref, rc)
        finally:
            renderer.set_tuning("lds_stack", 24)


def test_pipelines_odd_shapes(renderer, pipeline):
    """Odd widths, strided tiles, heat maps and zero samples on both pipelines."""
    cases = [(cornell_scene(True), dict(singleRender=1, sampleLimit=3), 100, 50),
             (model_scene("bunny.obj", material=5, spheres=True), dict(raysPerPixel=2, frameCount=2), 96, 64),
             (model_scene("klein_bottle.obj", material=4, scale=0.5, position=(0.0, -0.2, 0.0)), dict(singleRender=1, sampleLimit=2, debug=2, boxCap=300, triangleCap=60), 64, 48)]
    try:
        for s, kw, W, H in cases:
            pc = engine.push_constants(W, H, **kw)
            _check(*_render_both(renderer, s, pc, W, H))
        # tiles and zero samples
        s = cornell_scene(True)
        pc = engine.push_constants(40, 30, singleRender=1, sampleLimit=2)
        _check(*_render_both(renderer, s, pc, 40, 30, row0=1, rowStride=3, nRows=len(range(1, 30, 3))))
        pc = engine.push_constants(16, 8, raysPerPixel=0)
        _check(*_render_both(renderer, s, pc, 16, 8))
        assert renderer.last_pipeline() == pipeline
    finally:
        pass


def test_long_sample_chains_on_sampled_rows(renderer):
    """64 serial samples per pixel (the RNG state runs through all of them) on the Sponza stand-in at the
    BASELINE frame size: eight rows of the 1920x1080 frame, GPU tile vs oracle, bit for bit, with counters."""
    s, _ = scenes.sponza(0, ntris=40000)
    W, H = 1920, 1080
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=64)
    tile = dict(row0=67, rowStride=135, nRows=8)
    _check(*_render_both(renderer, s, pc, W, H, **tile))


def test_automatic_pipeline_choice(renderer):
    """pipeline -1: small tiles go to the fused kernel, big tiles of long-ray scenes to the multi-kernel pipeline, and a
    scene whose measured rays are short goes to the fused kernel at any size (the ray cost comes back asynchronously:
    it is known at the latest two synchronised dispatches after the first)."""
    r = renderer
    r.set_tuning("pipeline", -1)
    scene = cornell_scene()
    r.upload_scene(scene)
    pc = engine.push_constants(256, 256, singleRender=1, sampleLimit=1)
    r.render(pc, 256, 256)
    assert r.last_pipeline() == 1
    W, H = 2560, 1440  # 3.7 M pixels
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=1)
    for _ in range(3):
        r.render(pc, W, H)
    assert r.last_pipeline() == 1, "Cornell has ~10 box tests per ray"
    sp, _ = scenes.sponza(0, ntris=20000)
    r.upload_scene(sp)
    W, H = 3840, 2160  # 8.3 M pixels
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=1)
    r.render(pc, W, H)
    assert r.last_pipeline() == 0, "by size"
    for _ in range(2):
        r.render(pc, W, H)
    assert r.last_pipeline() == 0, "long rays (26 objects): the multi-kernel pipeline keeps the biggest tiles"
    for w, h in ((1280, 720), (960, 540)):   # below 1.5 M paths the fused kernel keeps scenes of any ray length
        r.render(scenes.sponza_camera(w, h, singleRender=1, sampleLimit=1), w, h)
        assert r.last_pipeline() == 1


def test_ray_cost_probe_before_a_big_first_dispatch(renderer):
    """A scene's first dispatch of >= 8 M pixel-samples is preceded by a probe (eight rows, one sample) that measures its
    box tests per ray, so that a single-render job runs with the launch parameters of its ray length. The probe leaves no
    trace: counters and pixels equal those of a context that never probed, and those of the oracle on sampled rows."""
    r = renderer
    r.set_tuning("pipeline", -1)
    sp, _ = scenes.sponza(0, ntris=20000)
    W, H = 1920, 1080
    pc = scenes.sponza_camera(W, H, singleRender=1, sampleLimit=4)
    out = []
    for probe in (1, 0):
        r.set_tuning("probe", probe)
        r.upload_scene(sp)                      # a new scene: ray cost unknown
        assert r.ray_cost() < 0
        r.reset_counters()
        img = r.render(pc, W, H)
        out.append((img, r.counters(), r.ray_cost()))
    r.set_tuning("probe", 1)
    assert out[0][2] > 90, "measured before the dispatch (Sponza stand-in: long rays)"
    assert out[1][2] < 0 or out[1][2] > 90      # without the probe: unknown until the counters come back
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    keys = ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments")
    assert {k: out[0][1][k] for k in keys} == {k: out[1][1][k] for k in keys}
    tile = dict(row0=50, rowStride=270, nRows=4)
    ref, _ = pyoracle.render(sp, pc, W, H, **tile)
    assert np.array_equal(out[0][0][50::270][:4].view(np.uint32), ref.view(np.uint32))


def test_instances_nonuniform_transforms_emissive_mesh_and_ten_spheres(renderer):
    """One OBJ loaded four times (cached BVH, src/vk_engine.cpp:802-815) under rotated, non-uniformly scaled and mirrored
    placements, one instance emissive (a second light the hard-wired NEE knows nothing about), one dielectric; all ten
    sphere slots in use with every material kind, some overlapping; environment on."""
    import os
    s = engine.Scene()
    s.prepare_storage_buffers()
    bunny = os.path.join(engine.ASSET_DIR, "bunny.obj")
    glow = s.add_material(engine.default_material(albedo=(0.9, 0.6, 0.2), emissionColor=(1.0, 0.5, 0.1), emissionStrength=3.0))
    s.read_obj(bunny, engine.placement(position=(-0.45, 0.55, 0.2), scale=(0.5, 0.9, 0.5), rotation=(0, 40, 0)), 0)
    s.read_obj(bunny, engine.placement(position=(0.45, 0.55, 0.1), scale=(0.6, 0.4, 0.8), rotation=(15, -70, 10)), 5)
    s.read_obj(bunny, engine.placement(position=(0.0, -0.6, 0.5), scale=(-0.5, 0.5, 0.5), rotation=(180, 0, 0)), glow)
    s.read_obj(bunny, engine.placement(position=(0.0, 0.6, -0.5), scale=(0.3, 0.3, 0.3), rotation=(0, 0, 90)), 4)
    kinds = [0, 1, 2, 4, 5, glow, 5, 4, 0, 3]
    for i in range(10):
        a = i * 0.7
        s.set_sphere(i, (0.7 * np.cos(a), -0.1 + 0.08 * i - 0.4, 0.6 * np.sin(a)), 0.12 + 0.02 * (i % 4), kinds[i])
    W, H = 112, 80
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6, environmentOn=True)
    _check(*_render_both(renderer, s, pc, W, H))


def test_repeated_spheres_of_different_materials(renderer):
    """Spheres that repeat an earlier sphere's centre and radius bit for bit are left out by the rays' creators
    (DevScene::sphereTestMask): the shader's loop would compute the earlier sphere's result again and never prefer it
    (strict `<`, raytrace.comp:282-287). Here the repeats carry other materials (emissive, mirror, dielectric) than their
    originals, sit before and after distinct spheres, one pair is emissive twice, and a dispatch with fewer spheres than
    were uploaded cuts a pair in two; then a sphere is edited in place so that a repeat becomes an original."""
    s = cornell_scene(False)
    glow = s.add_material(engine.default_material(albedo=(0.2, 0.2, 0.9), emissionColor=(0.2, 0.4, 1.0), emissionStrength=2.0))
    a, b, c = ((-0.45, 0.35, 0.1), 0.3), ((0.4, 0.2, -0.2), 0.35), ((0.0, -0.55, 0.3), 0.2)
    z = ((0.0, 0.0, 0.0), 0.0)  # the reference's unused sphere slots
    layout = [(a, 0), (b, 4), (a, glow), (c, glow), (b, 5), (c, glow), (a, 4), (z, 0), (z, 3), (b, 0)]
    for i, ((pos, rad), m) in enumerate(layout):
        s.set_sphere(i, pos, rad, m)
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6)
    _check(*_render_both(renderer, s, pc, W, H))
    pc5 = engine.push_constants(W, H, singleRender=1, sampleLimit=2, bounceLimit=6, sphereCount=5)
    _check(*_render_both(renderer, s, pc5, W, H))
    s.set_sphere(0, (-0.5, 0.3, 0.0), 0.25, 1)   # spheres 2 and 6 repeated sphere 0: now sphere 2 is the original of the two
    renderer.update_spheres(s)
    renderer.reset_counters()
    img = renderer.render(pc, W, H)
    ref, rc = pyoracle.render(s, pc, W, H)
    _check(img, renderer.counters(), ref, rc)


def test_camera_inside_a_mesh_and_grazing_rays(renderer):
    """The camera sits inside the klein bottle (back faces first, frontFace = false paths of the dielectric) and looks along
    a wall (grazing primary rays); a wide field of view sends rays past every edge of the box."""
    s = model_scene("klein_bottle.obj", material=5, scale=0.9, position=(0.0, 0.2, 0.0))
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, bounceLimit=10, pos=(0.02, 0.15, 0.03),
                               cameraAngles=(3.0, 88.0, 0.0), fov=120.0)
    _check(*_render_both(renderer, s, pc, W, H))
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, pos=(-0.995, -0.3, -0.9), cameraAngles=(0.0, 0.5, 0.0), fov=60.0)
    _check(*_render_both(renderer, s, pc, W, H))


def test_degenerate_and_coincident_triangles(renderer):
    """Zero-area triangles, zero-length normals (normalize(0) = NaN in the reference, H8), two coincident quads of
    different materials (equal hit distances: the strict '<' keeps the first one met), a triangle through the camera."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    quad = np.array([[[-0.5, 0.2, 0.3], [0.5, 0.2, 0.3], [0.5, -0.6, 0.3]], [[-0.5, 0.2, 0.3], [0.5, -0.6, 0.3], [-0.5, -0.6, 0.3]]], np.float32)
    nq = np.zeros_like(quad); nq[..., 2] = -1
    s.add_mesh("quad_a", quad, nq, engine.placement(), 1)
    s.add_mesh("quad_b", quad.copy(), nq, engine.placement(), 2)            # same place, other material
    deg = np.array([[[0.1, 0.1, 0.0], [0.1, 0.1, 0.0], [0.1, 0.1, 0.0]],     # a point
                    [[-0.3, 0.0, -0.2], [0.3, 0.0, -0.2], [0.0, 0.0, -0.2]],  # a segment
                    [[-0.2, -0.3, -0.4], [0.2, -0.3, -0.4], [0.0, -0.1, -0.4]]], np.float32)
    nd = np.zeros_like(deg)                                                  # zero normals on a real triangle too
    s.add_mesh("degenerate", deg, nd, engine.placement(), 0)
    thru = np.array([[[0.0, -0.5, -3.6], [0.3, -0.2, -2.0], [-0.3, -0.2, -2.0]]], np.float32)  # passes next to the eye
    nt = np.zeros_like(thru); nt[..., 1] = -1
    s.add_mesh("through", thru, nt, engine.placement(), 4)
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3)
    _check(*_render_both(renderer, s, pc, W, H))


def test_many_separate_identity_objects_of_different_materials(renderer):
    """Eight small meshes with identity transforms, far apart and each of another material, inside the Cornell box: most
    rays miss most of their root boxes, so the traversal jumps over runs of objects (the rays' object masks) and must still
    credit every hit to the object it is in and count the skipped objects' box tests."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    glow = s.add_material(engine.default_material(albedo=(0.2, 0.3, 0.9), emissionColor=(0.2, 0.4, 1.0), emissionStrength=1.5))
    teal = s.add_material(engine.default_material(albedo=(0.1, 0.8, 0.7)))
    mats = [0, 1, 2, 4, 5, glow, teal, 1]
    for k in range(8):
        pos, nrm = scenes.blob(180 + 40 * k, seed=10 + k, radius=0.13, center=(-0.75 + 0.5 * (k % 4), -0.75 + 0.9 * (k // 4), -0.5 + 0.3 * (k % 3)))
        s.add_mesh(f"blob{k}", pos, nrm, engine.placement(), mats[k])
    W, H = 128, 96
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6)
    _check(*_render_both(renderer, s, pc, W, H))


# RT_RANDOM_SEEDS=n widens the sweep (a soak run; the default keeps the suite short)
@pytest.mark.parametrize("seed", list(range(1, 1 + int(__import__("os").environ.get("RT_RANDOM_SEEDS", "8")))))
def test_random_scenes(renderer, seed):
    """Seeded random scenes: the Cornell box around (seeds 3 and 6: nothing around), up to four blob meshes under random (also mirrored and
    non-uniformly scaled) placements, random spheres, random materials (diffuse, mirror, dielectric, emissive, mixtures
    the panels allow), random camera, environment on or off, random bounce limit — pixels and counters against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    s = engine.Scene()
    if seed % 3:
        s.prepare_storage_buffers()
    else:
        for m in (engine.default_material(), engine.default_material(albedo=(1, 0, 0)), engine.default_material(albedo=(0, 1, 0)),
                  engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4),
                  engine.default_material(reflectance=1.0), engine.default_material(ior=2.0)):
            s.add_material(m)
        for i in range(10):
            s.set_sphere(i, (0, 0, 0), 0.0, 0)
    mats = list(range(6))
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 4)
        kw = dict(albedo=tuple(rng.random(3)))
        if kind == 1: kw.update(reflectance=float(rng.random() * 0.9 + 0.1))
        if kind == 2: kw.update(ior=float(1.1 + rng.random() * 1.5))
        if kind == 3: kw.update(emissionColor=tuple(rng.random(3)), emissionStrength=float(rng.random() * 4))
        mats.append(s.add_material(engine.default_material(**kw)))
    for k in range(int(rng.integers(1, 5))):
        pos, nrm = scenes.blob(int(rng.integers(40, 900)), seed=int(rng.integers(1, 1000)), radius=1.0)
        sc = rng.uniform(0.08, 0.35, 3) * rng.choice([-1.0, 1.0], 3, p=[0.15, 0.85])
        pl = engine.placement(position=tuple(rng.uniform(-0.7, 0.7, 3)), scale=tuple(sc) if rng.random() < 0.6 else float(abs(sc[0])),
                              rotation=tuple(rng.uniform(-180, 180, 3)) if rng.random() < 0.7 else (0, 0, 0))
        s.add_mesh(f"r{seed}_{k}", pos, nrm, pl, int(rng.choice(mats)))
    for i in range(int(rng.integers(0, 6))):
        s.set_sphere(i, tuple(rng.uniform(-0.8, 0.8, 3)), float(rng.uniform(0.05, 0.35)), int(rng.choice(mats)))
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=int(rng.integers(1, 5)), bounceLimit=int(rng.integers(1, 11)),
                               pos=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.8, 0.2)), float(rng.uniform(-3.8, -2.0))),
                               cameraAngles=(float(rng.uniform(-10, 10)), float(rng.uniform(-15, 15)), float(rng.uniform(-5, 5))),
                               fov=float(rng.uniform(35, 90)), environmentOn=bool(rng.random() < 0.5), frameCount=int(rng.integers(0, 4)))
    _check(*_render_both(renderer, s, pc, W, H))


def test_more_objects_than_the_object_mask_has_bits(renderer):
    """Forty-five objects: the Cornell box's nine, then thirty-six small meshes alternating between rotated / scaled
    placements (general transforms: entered only when the ray can reach their padded box), identity placements and
    two-triangle cards whose BVH root is a leaf. The rays' object masks cover 32 objects only (from the first placed one); the
    objects beyond them, and the boundary itself, must be walked, counted and credited exactly as the reference's linear loop does."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    glow = s.add_material(engine.default_material(albedo=(0.9, 0.9, 0.3), emissionColor=(1.0, 0.9, 0.4), emissionStrength=2.0))
    mats = [0, 1, 2, 4, 5, glow]
    card = np.array([[[-0.06, 0, -0.06], [0.06, 0, -0.06], [0.06, 0, 0.06]], [[-0.06, 0, -0.06], [0.06, 0, 0.06], [-0.06, 0, 0.06]]], np.float32)
    ncard = np.zeros_like(card); ncard[..., 1] = -1
    for k in range(36):
        where = (-0.8 + 0.32 * (k % 6), -0.85 + 0.3 * (k // 6), -0.7 + 0.25 * (k % 5))
        if k % 3 == 0:      # general transform, interior root
            pos, nrm = scenes.blob(96 + 8 * k, seed=40 + k, radius=1.0)
            s.add_mesh(f"g{k}", pos, nrm, engine.placement(position=where, scale=(0.09, 0.12, 0.07), rotation=(10 * k, 25 * k, 5 * k)), mats[k % 6])
        elif k % 3 == 1:    # identity, interior root
            pos, nrm = scenes.blob(64 + 6 * k, seed=40 + k, radius=0.09, center=where)
            s.add_mesh(f"i{k}", pos, nrm, engine.placement(), mats[k % 6])
        else:               # general transform, root is a leaf
            s.add_mesh(f"c{k}", card, ncard, engine.placement(position=where, rotation=(35 * k, 0, 20 * k)), mats[k % 6])
    assert s.counts()["objects"] == 45
    W, H = 128, 96
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6)
    _check(*_render_both(renderer, s, pc, W, H))


@pytest.mark.parametrize("tree_min", [0, 2, 48])
def test_object_hierarchy_over_many_placed_objects(renderer, tree_min):
    """N2: the reference walks its objects linearly (raytrace.comp:289-350). From 48 placed objects on, the set-up step's skipping
    loop jumps aligned blocks of 2..256 consecutive placed objects whose union box the ray cannot reach, at the reference's cost
    for them, in the reference's order ("object_tree_min": 0 = never, 2 = already here). 150 objects: the Cornell box's nine, then
    runs of placed meshes and leaf-root cards broken by identity-transform meshes (blocks must not straddle those), an emissive
    one among them; pixels and every counter as the oracle's linear loop gives them."""
    renderer.set_tuning("object_tree_min", tree_min)
    try:
        s = engine.Scene()
        s.prepare_storage_buffers()
        glow = s.add_material(engine.default_material(albedo=(0.9, 0.9, 0.3), emissionColor=(1.0, 0.9, 0.4), emissionStrength=2.0))
        mats = [0, 1, 2, 4, 5, glow]
        card = np.array([[[-0.05, 0, -0.05], [0.05, 0, -0.05], [0.05, 0, 0.05]], [[-0.05, 0, -0.05], [0.05, 0, 0.05], [-0.05, 0, 0.05]]], np.float32)
        ncard = np.zeros_like(card); ncard[..., 1] = -1
        blobs = [scenes.blob(60 + 10 * k, seed=200 + k, radius=1.0) for k in range(4)]
        for k in range(141):
            where = (-0.85 + 0.17 * (k % 11), -0.9 + 0.11 * (k // 11), -0.8 + 0.16 * (k % 9))
            if k % 37 == 36:      # an identity-transform mesh breaks the run of placed objects
                pos, nrm = scenes.blob(40, seed=300 + k, radius=0.05, center=where)
                s.add_mesh(f"i{k}", pos, nrm, engine.placement(), mats[k % 6])
            elif k % 5 == 4:
                s.add_mesh(f"c{k}", card, ncard, engine.placement(position=where, rotation=(35 * k, 0, 20 * k)), mats[k % 6])
            else:
                pos, nrm = blobs[k % 4]
                s.add_mesh(f"b{k % 4}", pos, nrm, engine.placement(position=where, scale=(0.04, 0.05, 0.035), rotation=(9 * k, 23 * k, 4 * k)), mats[k % 6])
        assert s.counts()["objects"] == 150
        W, H = 112, 84
        pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6)
        _check(*_render_both(renderer, s, pc, W, H))
        pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, debug=2, boxCap=900, triangleCap=60)   # per-pixel counts through the heat map
        _check(*_render_both(renderer, s, pc, W, H))
    finally:
        renderer.set_tuning("object_tree_min", 48)


@pytest.mark.parametrize("n_identity,n_placed", [(34, 20), (3, 40), (31, 2)])
def test_object_mask_window_starts_at_the_first_placed_object(renderer, n_identity, n_placed):
    """C5's shape: identity-transform groups first, placed (general-transform) objects after them. The 32 bits of a ray's
    object mask start at the first placed object (DevScene::maskBase), so 34 groups + 20 placed objects are all under the
    mask although they are objects 34..53; with 40 placed objects the window ends inside them; with 31 + 2 it straddles
    object 32. Leaf-root cards among the placed objects, an emissive one, a light query ending inside a skipped run."""
    import os
    s = engine.Scene()            # no Cornell box (its placed walls would come first): materials and spheres as scenes.sponza() sets them
    for i in range(10):
        s.set_sphere(i, (0, 0, 0), 0.0, 0)
    for m in (engine.default_material(), engine.default_material(albedo=(1, 0, 0)), engine.default_material(albedo=(0, 1, 0)),
              engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4),
              engine.default_material(reflectance=1.0), engine.default_material(ior=2.0)):
        s.add_material(m)
    glow = s.add_material(engine.default_material(albedo=(0.9, 0.9, 0.3), emissionColor=(1.0, 0.9, 0.4), emissionStrength=2.0))
    mats = [0, 1, 2, 4, 5, glow]
    card = np.array([[[-0.06, 0, -0.06], [0.06, 0, -0.06], [0.06, 0, 0.06]], [[-0.06, 0, -0.06], [0.06, 0, 0.06], [-0.06, 0, 0.06]]], np.float32)
    ncard = np.zeros_like(card); ncard[..., 1] = -1
    fpos, fnrm = scenes.grid_patch((-1.0, 1.0, -1.0), (2.0, 0, 0), (0, 0, 2.0), 6, 6)
    s.add_mesh("floor", fpos, fnrm, engine.placement(), 0)
    first = s.counts()["objects"]
    for k in range(n_identity):
        where = (-0.8 + 0.27 * (k % 7), 0.9 - 0.12 * (k // 7), -0.8 + 0.2 * (k % 5))
        pos, nrm = scenes.blob(48 + 4 * k, seed=70 + k, radius=0.06, center=where)
        s.add_mesh(f"i{k}", pos, nrm, engine.placement(), mats[k % 5])
    for k in range(n_placed):
        where = (-0.75 + 0.3 * (k % 6), -0.8 + 0.22 * (k // 6), -0.6 + 0.3 * (k % 4))
        if k % 4 == 3:
            s.add_mesh(f"c{k}", card, ncard, engine.placement(position=where, rotation=(35 * k, 0, 20 * k)), mats[k % 6])
        else:
            pos, nrm = scenes.blob(80 + 6 * k, seed=90 + k, radius=1.0)
            s.add_mesh(f"g{k}", pos, nrm, engine.placement(position=where, scale=(0.08, 0.1, 0.07), rotation=(12 * k, 31 * k, 7 * k)), mats[k % 6])
    s.read_obj(os.path.join(engine.ASSET_DIR, "light2.obj"), engine.placement(position=(0, -1.5, 0), frontOnly=True), 3)
    assert s.counts()["objects"] == first + n_identity + n_placed + 1
    W, H = 112, 84
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6, environmentOn=True)
    _check(*_render_both(renderer, s, pc, W, H))


@pytest.mark.parametrize("config", ["cornell", "bunny", "dragon", "sponza", "sponza_dragons", "sponza_dragons_flat"])
def test_every_bench_scene_is_identical_across_the_three_kernels(renderer, config):
    """The BASELINE configs (full triangle counts) through k_trace (one ray per lane, no object skipping, no shared traversal
    code), k_trace_pw and k_render_fused: same pixels, same counters. The oracle is too slow for these sizes; k_trace, which
    the small-scene tests pin to the oracle, stands in for it."""
    scene, _ = scenes.CONFIGS[config]()
    r = renderer
    r.upload_scene(scene)
    W, H = 384, 216
    cam = scenes.sponza_camera if config.startswith("sponza") else engine.push_constants
    pc = cam(W, H, singleRender=1, sampleLimit=2)
    out = []
    for knobs in ({"pipeline": 0, "trace_variant": 0}, {"pipeline": 0, "trace_variant": 1}, {"pipeline": 1}):
        for k, v in knobs.items():
            r.set_tuning(k, v)
        r.reset_counters()
        img = r.render(pc, W, H)
        c = r.counters()
        out.append((img, {k: c[k] for k in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments")}))
    r.set_tuning("trace_variant", 1)
    for img, cnt in out[1:]:
        assert np.array_equal(img.view(np.uint32), out[0][0].view(np.uint32))
        assert cnt == out[0][1]


# ---------------------------------------------------------------- update_buffer (src/vk_engine.cpp:1446-1475,1545,1572,1603)
def _after_edit(renderer, ed, pc, W, H):
    renderer.reset_counters()
    img = renderer.render(pc, W, H)
    cnt = renderer.counters()
    ref, rc = pyoracle.render(ed, pc, W, H)
    _check(img, cnt, ref, rc)
    return img


def test_update_materials_in_place(renderer):
    """The "Update Buffer" button of the material editor: the white diffuse material becomes emissive, then a mirror, then
    a dielectric, each time without a new upload; the frame must follow the oracle on the edited arrays."""
    from util import EditedScene
    s = cornell_scene(True)
    W, H = 80, 60
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3)
    renderer.upload_scene(s)
    ed = EditedScene(s)
    frames = [_after_edit(renderer, ed, pc, W, H)]
    m = ed.materials[0]
    m.emissionColor[:] = [0.9, 0.7, 0.4]; m.emissionStrength = 0.8        # diffuse -> emissive (a second light NEE does not know)
    ed.push(renderer, "materials")
    frames.append(_after_edit(renderer, ed, pc, W, H))
    m.emissionStrength = 0.0; m.reflectance = 1.0                          # -> mirror
    ed.push(renderer, "materials")
    frames.append(_after_edit(renderer, ed, pc, W, H))
    m.reflectance = 0.0; m.ior = 1.5                                       # -> dielectric
    ed.push(renderer, "materials")
    frames.append(_after_edit(renderer, ed, pc, W, H))
    ed.materials[3].emissionStrength = 0.0                                 # the light switched off: nothing emits
    ed.push(renderer, "materials")
    frames.append(_after_edit(renderer, ed, pc, W, H))
    for a, b in zip(frames, frames[1:]):
        assert not np.array_equal(a, b)


def _identity_blobs_scene():
    s = engine.Scene()
    for m in (engine.default_material(albedo=(0.8, 0.8, 0.8)), engine.default_material(albedo=(0.9, 0.2, 0.2)),
              engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4),
              engine.default_material(reflectance=1.0), engine.default_material(ior=1.7)):
        s.add_material(m)
    for i in range(10):
        s.set_sphere(i, (0, 0, 0), 0.0, 0)
    for k in range(4):
        pos, nrm = scenes.blob(300 + 60 * k, seed=70 + k, radius=0.3, center=(-0.9 + 0.6 * k, -0.5 + 0.1 * k, 0.2 * k))
        s.add_mesh(f"ub{k}", pos, nrm, engine.placement(), [0, 1, 3, 4][k])
    s.read_obj(__import__("os").path.join(engine.ASSET_DIR, "light2.obj"), engine.placement(position=(0, -1.5, 0), frontOnly=True), 2)
    return s


def test_update_objects_identity_to_general_and_back(renderer):
    """The object editor: identity placements become rotated and non-uniformly scaled (which switches the kernels to the
    template with the object-skipping code and rebuilds the padded world boxes, masks and skip costs), then identity again."""
    from util import EditedScene
    s = _identity_blobs_scene()
    W, H = 96, 64
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, bounceLimit=6, environmentOn=True, pos=(0, -0.5, -3.0))
    renderer.upload_scene(s)
    ed = EditedScene(s)
    a = _after_edit(renderer, ed, pc, W, H)
    ed.set_transform(0, engine.placement(position=(0.2, 0.1, 0.3), rotation=(20, 45, 10), scale=(1.3, 0.6, 0.9)))
    ed.set_transform(1, engine.placement(position=(-0.3, 0.0, -0.2), rotation=(0, -70, 35), scale=(0.5, 1.4, -1.0)))   # mirrored
    ed.set_transform(2, engine.placement(position=(0.0, 0.3, 0.0), scale=1.0))                                          # a pure translation
    ed.push(renderer, "objects")
    b = _after_edit(renderer, ed, pc, W, H)
    assert not np.array_equal(a, b)
    for k in range(3):
        ed.set_transform(k, engine.placement())
    ed.push(renderer, "objects")
    c = _after_edit(renderer, ed, pc, W, H)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), "back to the identity placements: the first frame again"


def test_update_objects_reorder_across_the_mask_boundary_and_repoint_an_instance(renderer):
    """Forty objects: entries swapped across the 32-object boundary of the rays' object masks, an object re-pointed to
    another mesh's BVH (bvhIndex) and given another material — the derived tables follow, hits are credited as the
    reference's linear object loop credits them."""
    from util import EditedScene
    s = engine.Scene()
    s.prepare_storage_buffers()
    for k in range(31):
        where = (-0.8 + 0.27 * (k % 7), -0.9 + 0.35 * (k // 7), -0.6 + 0.3 * (k % 4))
        pos, nrm = scenes.blob(60 + 10 * k, seed=300 + k, radius=1.0)
        pl = engine.placement(position=where, scale=(0.08, 0.1, 0.07), rotation=(13 * k, 29 * k, 7 * k)) if k % 2 else engine.placement(position=where, scale=0.09)
        s.add_mesh(f"m{k}", pos, nrm, pl, [0, 1, 2, 4, 5][k % 5])
    assert s.counts()["objects"] == 40
    W, H = 112, 84
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, bounceLimit=5)
    renderer.upload_scene(s)
    ed = EditedScene(s)
    a = _after_edit(renderer, ed, pc, W, H)
    import ctypes as C
    from ray_tracer_amd._capi import RenderObject
    for i, j in ((10, 36), (31, 32), (0, 39)):                     # swap entries across the boundary
        tmp = RenderObject()
        C.memmove(C.byref(tmp), C.byref(ed.objects[i]), C.sizeof(RenderObject))
        C.memmove(C.byref(ed.objects[i]), C.byref(ed.objects[j]), C.sizeof(RenderObject))
        C.memmove(C.byref(ed.objects[j]), C.byref(tmp), C.sizeof(RenderObject))
    ed.push(renderer, "objects")
    b = _after_edit(renderer, ed, pc, W, H)
    ed.objects[12].bvhIndex = ed.objects[20].bvhIndex               # an instance of another mesh now
    ed.objects[12].materialIndex = 4
    ed.objects[35].bvhIndex = ed.objects[2].bvhIndex
    ed.push(renderer, "objects")
    c = _after_edit(renderer, ed, pc, W, H)
    assert not np.array_equal(b, c)
    assert a.shape == b.shape


def test_far_camera_and_far_instances(renderer):
    """the padded world-space boxes that let a ray skip general-transform objects are only trusted while the
    ray starts within 1e3 object scales; a camera 1e5 units away and an instance translated to large coordinates must
    still give the oracle's pixels and counters."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    pos, nrm = scenes.blob(400, seed=5, radius=1.0)
    s.add_mesh("far_a", pos, nrm, engine.placement(position=(0.3, 0.2, 0.1), scale=(0.2, 0.3, 0.2), rotation=(10, 20, 30)), 1)
    s.add_mesh("far_b", pos, nrm, engine.placement(position=(5000.0, -3000.0, 8000.0), scale=(30.0, 20.0, 25.0), rotation=(40, 10, 5)), 4)
    W, H = 64, 48
    for kw in (dict(pos=(0.0, -0.5, -100000.0), fov=0.002, cameraAngles=(0, 0, 0)), dict(pos=(0.0, -0.5, -2000.0), fov=0.1, cameraAngles=(0, 0, 0)),
               dict(pos=(5000.0, -3000.0, 7800.0), fov=40.0),
               dict(pos=(0.0, -0.5, -3.5))):
        pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, environmentOn=True, **kw)
        _check(*_render_both(renderer, s, pc, W, H))


def test_srgb8_readback_against_numpy(renderer):
    """rt_read_rgba8_srgb (the reference's display format, R8G8B8A8_SRGB, src/vk_engine.cpp:1380) against the sRGB transfer
    function evaluated in float64 on the oracle's frame: at most 1 LSB apart (the device-side pow is the polynomial one), and
    exact for more than 99.5 % of the values."""
    s = cornell_scene(True)
    W, H = 96, 64
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=4, environmentOn=True)
    renderer.upload_scene(s)
    img = renderer.render(pc, W, H)
    got = renderer.read_rgba8_srgb()
    ref, _ = pyoracle.render(s, pc, W, H)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    v = np.clip(np.nan_to_num(ref.astype(np.float64), nan=0.0), 0.0, 1.0)
    enc = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055)
    enc[..., 3] = v[..., 3]
    want = np.floor(enc * 255.0 + 0.5).astype(np.int32)
    diff = np.abs(got.astype(np.int32) - want)
    assert diff.max() <= 1
    assert (diff == 0).mean() > 0.995
    assert got[..., :3].max() > 200 and got[..., :3].min() < 30 and (got[..., 3] == 255).all()


def test_light_queries_on_and_off(renderer):
    """The NEE ray and the cosine probe answered from the emitter list (default) or traversed in full (light_queries 0):
    the same pixels, and in either mode the executed-work counters the oracle predicts for that mode. Scenes: the Cornell
    light only; an emissive sphere and a second emissive quad; an emissive mesh beyond RT_EMIT_MAX_TRIS (shortcut off)."""
    import os
    scn = []
    scn.append(("cornell", cornell_scene(True)))
    s = cornell_scene(True)
    glow = s.add_material(engine.default_material(albedo=(0.1, 0.1, 0.1), emissionColor=(0.3, 0.6, 1.0), emissionStrength=1.2))
    s.set_sphere(3, (-0.6, -0.9, 0.4), 0.2, glow)
    quad = np.array([[[-0.2, -1.499, 0.5], [0.2, -1.499, 0.5], [0.2, -1.499, 0.8]], [[-0.2, -1.499, 0.5], [0.2, -1.499, 0.8], [-0.2, -1.499, 0.8]]], np.float32)
    nq = np.zeros_like(quad); nq[..., 1] = 1
    s.add_mesh("second_light", quad, nq, engine.placement(rotation=(0, 15, 0)), glow)
    scn.append(("sphere+quad", s))
    s = cornell_scene(False)
    glow = s.add_material(engine.default_material(albedo=(0.9, 0.6, 0.2), emissionColor=(1.0, 0.5, 0.1), emissionStrength=3.0))
    s.read_obj(os.path.join(engine.ASSET_DIR, "bunny.obj"), engine.placement(position=(0.0, 0.4, 0.0), scale=0.5), glow)
    scn.append(("emissive-bunny", s))
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3)
    try:
        for name, s in scn:
            frames = []
            for on in (1, 0):
                renderer.set_tuning("light_queries", on)
                pyoracle.lib().oracle_set_light_queries(on)
                img, cnt, ref, rc = _render_both(renderer, s, pc, W, H)
                _check(img, cnt, ref, rc)
                frames.append((img, cnt))
            assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32)), name
            if name != "emissive-bunny":
                assert frames[0][1]["raysTraced"] < 0.8 * frames[1][1]["raysTraced"], "the shortcut must save light queries"
                assert frames[0][1]["emitterTests"] > 0
            else:
                assert frames[0][1]["emitterTests"] == 0 and frames[0][1]["raysTraced"] == frames[1][1]["raysTraced"]
    finally:
        renderer.set_tuning("light_queries", 1)
        pyoracle.lib().oracle_set_light_queries(1)


def test_render_frames_equals_frame_by_frame(renderer):
    """rt_render_frames: several progressive frames of a tile in one launch (their pixels are independent until they are
    blended) against the same frames dispatched one by one, and against the oracle's chain of dispatches — pixels and
    counters, for a whole frame, an interleaved tile, a frame count that does not divide by the frames per launch, and the
    non-progressive case (every frame overwrites the last)."""
    s = model_scene("bunny.obj", material=0, spheres=True)
    W, H = 96, 64
    renderer.upload_scene(s)
    try:
        for tile, nF, f0, prog, fpl in ((dict(), 5, 0, 1, 0), (dict(row0=1, rowStride=3, nRows=len(range(1, H, 3))), 7, 2, 1, 3),
                                        (dict(), 3, 4, 0, 0), (dict(), 1, 0, 1, 0)):
            renderer.set_tuning("frames_per_launch", fpl)
            rows = tile.get("nRows", H)
            # frame by frame on the GPU (the context's own image keeps the progressive history), and the oracle's chain
            ref = None
            renderer.clear_framebuffer()
            renderer.reset_counters()
            for f in range(nF):
                pc = engine.push_constants(W, H, raysPerPixel=2, progressive=prog, frameCount=f0 + f)
                a = renderer.render(pc, W, H, **tile)
                ref, _ = pyoracle.render(s, pc, W, H, prev=ref, **tile)
            c_seq = renderer.counters()
            # all at once
            renderer.clear_framebuffer()
            renderer.reset_counters()
            pc = engine.push_constants(W, H, raysPerPixel=2, progressive=prog, frameCount=f0)
            b = renderer.render_frames(pc, W, H, nF, **tile)
            c_all = renderer.counters()
            assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), "frame by frame differs from the oracle"
            assert np.array_equal(b.view(np.uint32), a.view(np.uint32)), f"rt_render_frames differs (nFrames {nF}, tile {tile})"
            for k in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests"):
                assert c_all[k] == c_seq[k], k
    finally:
        renderer.set_tuning("frames_per_launch", 0)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("RT_RANDOM_SEEDS", "6")))))
def test_random_emitter_scenes(renderer, seed):
    """Random emissive triangles and spheres among random occluders (tests/util.py): the light queries' skip and early stop
    against the oracle — pixels, executed-work counters, and the oracle's own check of every answer."""
    from util import random_emitter_scene
    s, pc, W, H = random_emitter_scene(seed)
    _check(*_render_both(renderer, s, pc, W, H))
