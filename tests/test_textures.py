"""Texture sampling.

PARITY UNPINNED: the reference snapshot uploads textures (src/vk_engine.cpp:1109-1166, src/vk_textures.cpp:103-200) and
interpolates hit.uv (shaders/raytrace.comp:249-256) but its shader never samples one, so there is nothing in it to be
faithful to beyond the set-up: R8G8B8A8_SRGB images, a nearest-filter repeat sampler and a nearest-filter clamp-to-edge
sampler (:525-531) chosen by RenderObject.samplerIndex, the slot order of read_mtl. The semantics implemented and tested
here are this build's declared choice (include/rt_amd.h, rt_upload_textures): albedo *= texel(albedoIndex, hit.uv).
The reference's renders/dread_texture.png (parameters unrecorded) is an eyeball check only.

The other three slots read_mtl claims (map_Ks -> metalnessIndex, map_d -> alphaIndex, map_bump -> bumpIndex,
src/vk_engine.cpp:1109-1141) are likewise declared (include/rt_det_math.h): alpha cuts triangle hits out inside the traversal,
metalness replaces the material's reflectance, bump tilts the interpolated normal. No shipped MTL reaches them with an image
the snapshot holds, so the tests bind generated maps to test_plane.obj's material by hand.

CPU: properties of the oracle's restatement. GPU: the HIP path against the oracle, bit for bit, on dread.obj with its
albedo map (assets of the reference, data) and on test_plane.obj with its MTL's two maps."""
import os
import shutil

import numpy as np
import pytest

from oracle import pyoracle
from ray_tracer_amd import engine

from util import EditedScene, assert_hits_equal, cornell_scene, seeded_rays


def _dread_scene():
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.read_obj(os.path.join(engine.ASSET_DIR, "dread.obj"), engine.placement(position=(0.0, 0.45, 0.0), scale=0.45, rotation=(0, 200, 0)), 0)
    # dread.mtl names no map; the author bound dread_alb.png by hand (renders/dread_texture.png): the same here
    slot = s.add_texture(os.path.join(engine.ASSET_DIR, "dread_alb.png"))
    mi = s.find_material(os.path.join(engine.ASSET_DIR, "dread.mtl") + "/M_Body")
    assert mi >= 0 and slot == 0
    m = s.material(mi)
    m.albedoIndex = slot
    s.set_material(mi, m)
    return s


def _checker(w, h, a=(230, 40, 40, 255), b=(30, 60, 220, 255), cell=4):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.where((((xx // cell) + (yy // cell)) % 2 == 0)[..., None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)
    img[0, :, :] = (255, 255, 255, 255)     # first row and column marked: clamping and wrapping differ there
    img[:, 0, :] = (10, 200, 10, 255)
    return img


def _plane_scene(tmp_path, sampler, uv_scale=1.0):
    """test_plane.obj + test_plane.mtl (map_Bump is skipped: case-sensitive; map_Kd claims slot 0) in a temporary directory
    next to a generated vase_dif.png; the plane replaces the Cornell box's floor region."""
    d = tmp_path / f"plane_{sampler}_{uv_scale}"
    d.mkdir(parents=True)
    for f in ("test_plane.obj", "test_plane.mtl"):
        shutil.copy(os.path.join(engine.ASSET_DIR, f), d / f)
    if uv_scale != 1.0:   # uvs beyond [0, 1]: the two samplers must disagree
        lines = []
        for ln in open(d / "test_plane.obj"):
            if ln.startswith("vt "):
                u, v = (float(x) for x in ln.split()[1:3])
                ln = f"vt {u * uv_scale - 0.75:.6f} {v * uv_scale - 0.75:.6f}\n"
            lines.append(ln)
        open(d / "test_plane.obj", "w").writelines(lines)
    from PIL import Image
    Image.fromarray(_checker(24, 16)).save(d / "vase_dif.png")
    s = cornell_scene(False)
    s.read_obj(str(d / "test_plane.obj"), engine.placement(position=(0.0, 0.3, 0.2), scale=0.6, samplerIndex=sampler), 0)
    assert s.texture_paths() == [str(d / "vase_dif.png")]
    # read_obj leaves the samplerIndex of a file's last (here: only) group at 0 whatever the placement says
    # (src/vk_engine.cpp:1009-1019 never copies it); the object editor's field is what selects the clamp sampler
    ed = EditedScene(s)
    assert ed.objects[ed.nObjects - 1].samplerIndex == 0
    ed.objects[ed.nObjects - 1].samplerIndex = sampler
    ed.texture_paths = s.texture_paths
    ed.find_material, ed.material, ed.set_material = s.find_material, s.material, s.set_material
    return ed


def _plane_material(s):
    return s.find_material(s.texture_paths()[0].replace("vase_dif.png", "test_plane.mtl") + "/Material.001")


def _grey(w, h, value):
    """A map whose red channel is `value` (a number, or an [h, w] array) — the channel the three maps read."""
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = value
    img[..., 1] = 7; img[..., 2] = 201; img[..., 3] = 255   # the other channels must not matter
    return img


def _bound(tmp_path, sampler=0, uv_scale=1.0, **slots):
    """The plane scene with map slots bound on the plane's material: _bound(tmp, alphaIndex=1, reflectance=1.0, ...)."""
    s = _plane_scene(tmp_path, sampler, uv_scale)
    mi = _plane_material(s)
    for k, v in slots.items():
        setattr(s.materials[mi], k, v)
    return s


def test_alpha_map_cuts_hits_out(tmp_path):
    """Declared semantics: a triangle hit whose alpha texel decodes below 0.5 (red byte < 188) is no hit, for every kind of ray.
    Opaque everywhere == no map; transparent everywhere == the plane is not there; the threshold sits between bytes 187 and 188."""
    W, H = 72, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, environmentOn=True)

    def run(tex, scene):
        pyoracle.set_textures(tex)
        return pyoracle.render(scene, pc, W, H)[0]
    try:
        chk = _checker(24, 16)
        plain = run([chk], _bound(tmp_path))
        assert np.array_equal(run([chk, _grey(8, 8, 255)], _bound(tmp_path / "a", alphaIndex=1)), plain)
        assert np.array_equal(run([chk, _grey(8, 8, 188)], _bound(tmp_path / "b", alphaIndex=1)), plain)
        gone = run([chk, _grey(8, 8, 0)], _bound(tmp_path / "c", alphaIndex=1))
        assert np.array_equal(run([chk, _grey(8, 8, 187)], _bound(tmp_path / "d", alphaIndex=1)), gone)
        away = _plane_scene(tmp_path / "e", 0)
        away.set_transform(away.nObjects - 1, engine.placement(position=(0.0, 50.0, 0.2), scale=0.6))
        assert np.array_equal(run([chk], away), gone)
        assert not np.array_equal(gone, plain)
        holes = np.where((np.mgrid[0:8, 0:8].sum(axis=0) % 2) == 0, 255, 0)
        half = run([chk, _grey(8, 8, holes)], _bound(tmp_path / "f", alphaIndex=1))
        assert not np.array_equal(half, plain) and not np.array_equal(half, gone)
        # a slot beyond the uploaded table binds nothing
        assert np.array_equal(run([chk], _bound(tmp_path / "g", alphaIndex=1, metalnessIndex=5, bumpIndex=63)), plain)
        # the byte threshold is the decoded value 0.5
        lin = lambda b: float(pyoracle.glsl_probe(np.array([[0] * 15 + [b] + [0] * 16], np.float32))[0, 54])  # noqa: E731
        assert lin(187) < 0.5 <= lin(188)
    finally:
        pyoracle.set_textures([])


def test_metalness_map_replaces_the_reflectance(tmp_path):
    """Declared semantics: reflectance = decoded red of the metalness texel; the shader only asks reflectance != 0 (mirror)."""
    W, H = 72, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, environmentOn=True)

    def run(tex, scene):
        pyoracle.set_textures(tex)
        return pyoracle.render(scene, pc, W, H)[0]
    try:
        chk = _checker(24, 16)
        diffuse = run([chk], _bound(tmp_path))
        mirror = run([chk], _bound(tmp_path / "a", reflectance=1.0))
        assert not np.array_equal(diffuse, mirror)
        assert np.array_equal(run([chk, _grey(4, 4, 0)], _bound(tmp_path / "b", metalnessIndex=1, reflectance=1.0)), diffuse)
        assert np.array_equal(run([chk, _grey(4, 4, 255)], _bound(tmp_path / "c", metalnessIndex=1)), mirror)
        assert np.array_equal(run([chk, _grey(4, 4, 3)], _bound(tmp_path / "d", metalnessIndex=1)), mirror)   # any value but 0
        stripes = np.where(np.arange(8)[None, :] % 2 == 0, 255, 0) * np.ones((8, 1), int)
        mixed = run([chk, _grey(8, 8, stripes)], _bound(tmp_path / "e", metalnessIndex=1))
        assert not np.array_equal(mixed, diffuse) and not np.array_equal(mixed, mirror)
    finally:
        pyoracle.set_textures([])


def test_bump_map_tilts_the_normal(tmp_path):
    """Declared semantics (rt_bump_normal): n' = normalize(n) - (hx * dP/du^ - hy * dP/dv^), hx / hy the height steps to the next
    texel of the row / column; a level height field changes nothing, bit for bit."""
    W, H = 72, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, environmentOn=True)
    try:
        chk = _checker(24, 16)
        pyoracle.set_textures([chk])
        plain = pyoracle.render(_bound(tmp_path), pc, W, H)[0]
        pyoracle.set_textures([chk, _grey(6, 5, 131)])
        assert np.array_equal(pyoracle.render(_bound(tmp_path / "a", bumpIndex=1), pc, W, H)[0], plain)
        ramp = (np.arange(16) * 16)[None, :] * np.ones((16, 1), int)      # height grows along the row, i.e. with u
        pyoracle.set_textures([chk, _grey(16, 16, ramp)])
        s = _bound(tmp_path / "b", bumpIndex=1)
        assert not np.array_equal(pyoracle.render(s, pc, W, H)[0], plain)
        # one ray straight down onto the plane's middle: the plane is test_plane.obj scaled by 0.6 (no rotation), uv (u, v) at
        # object-space x = 2 u - 1, so dP/du^ = +x, dP/dv^ = (0, 1.644569, 2)^; hy = 0 on this map
        o, d = np.array([[0.03, -2.0, 0.21]], np.float32), np.array([[0.0, 1.0, 0.0]], np.float32)
        hit = pyoracle.trace_rays(s, o, d)[0]
        pyoracle.set_textures([chk])
        flat = pyoracle.trace_rays(s, o, d)[0]
        assert hit.didHit and flat.didHit and hit.triHitIndex == flat.triHitIndex and hit.dst == flat.dst
        n0 = np.array(flat.normal[:], np.float64)
        x = int(np.floor(((0.03 / 0.6) + 1) / 2 * 16))
        lin = lambda b: float(pyoracle.glsl_probe(np.array([[0] * 15 + [b] + [0] * 16], np.float32))[0, 54])  # noqa: E731
        hx = lin(ramp[0, x + 1]) - lin(ramp[0, x])
        sgn = 1.0 if flat.frontFace else -1.0
        want = sgn * (sgn * n0 - hx * np.array([1.0, 0.0, 0.0]))
        want /= np.linalg.norm(want)
        assert np.allclose(np.array(hit.normal[:]), want, atol=2e-6), (hit.normal[:], want)
        assert abs(np.array(hit.normal[:])[0] - n0[0]) > 0.01
    finally:
        pyoracle.set_textures([])


def test_mtl_slots_and_paths(tmp_path):
    s = _plane_scene(tmp_path, 0)
    mi = s.find_material(s.texture_paths()[0].replace("vase_dif.png", "test_plane.mtl") + "/Material.001")
    m = s.material(mi)
    assert (m.albedoIndex, m.bumpIndex, m.metalnessIndex, m.alphaIndex) == (0, -1, -1, -1)   # map_Bump: skipped (case-sensitive)
    s2 = engine.Scene()
    s2.read_mtl(os.path.join(engine.ASSET_DIR, "sponza.mtl"))
    paths = s2.texture_paths()
    assert len(paths) > 10 and all(p.startswith(engine.ASSET_DIR + "/") for p in paths)
    assert os.path.basename(paths[0]) == "lion.png"     # sponza.mtl's first map_Kd; its map_Bump lines claim nothing


def test_constant_texture_equals_scaled_albedo(tmp_path):
    """A one-colour texture is the material's albedo times that colour: the same pixels, bit for bit, as the untextured
    scene with the product as its albedo."""
    s = _plane_scene(tmp_path, 0)
    W, H = 72, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3)
    tex = np.full((5, 7, 4), (188, 64, 230, 255), np.uint8)
    try:
        pyoracle.set_textures([tex])
        a, _ = pyoracle.render(s, pc, W, H)
        pyoracle.set_textures([])
        plain, _ = pyoracle.render(s, pc, W, H)
        lin = [float(pyoracle.glsl_probe(np.array([[0] * 15 + [b] + [0] * 16], np.float32))[0, 54]) for b in (188, 64, 230)]
        mi = [i for i in range(s.counts()["materials"]) if s.material(i).albedoIndex == 0][0]
        m = s.material(mi)
        m.albedo[:] = [float(np.float32(m.albedo[k]) * np.float32(lin[k])) for k in range(3)]
        m.albedoIndex = -1
        s.set_material(mi, m)
        b, _ = pyoracle.render(EditedScene(s.scene), pc, W, H)
    finally:
        pyoracle.set_textures([])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(a, plain)


def test_repeat_and_clamp_samplers_differ_beyond_the_unit_square(tmp_path):
    W, H = 72, 54
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    out = {}
    try:
        pyoracle.set_textures([_checker(24, 16)])
        for sampler in (0, 1):
            out[sampler, 1.0], _ = pyoracle.render(_plane_scene(tmp_path, sampler), pc, W, H)
            out[sampler, 2.5], _ = pyoracle.render(_plane_scene(tmp_path, sampler, 2.5), pc, W, H)
    finally:
        pyoracle.set_textures([])
    assert not np.array_equal(out[0, 2.5], out[1, 2.5])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(0, 0), (1, 64), (1, 8)], ids=["multikernel", "fused", "fused-refill"])
def test_textured_scenes_against_the_oracle(renderer, tmp_path, mode):
    """HIP path == oracle, pixels and counters, with textures: dread.obj + dread_alb.png (2 376 triangles with uvs), and
    test_plane.obj under both samplers with uvs inside and beyond the unit square."""
    renderer.set_tuning("pipeline", mode[0])
    renderer.set_tuning("pixel_refill", mode[1])
    keys = ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests")
    try:
        cases = [(_dread_scene(), None, dict(sampleLimit=3))]
        for sampler in (0, 1):
            for sc in (1.0, 2.5):
                cases.append((_plane_scene(tmp_path, sampler, sc), [_checker(24, 16)], dict(sampleLimit=2, environmentOn=True)))
        W, H = 112, 84
        for s, tex, kw in cases:
            tex = tex if tex is not None else engine.load_textures(s)
            pc = engine.push_constants(W, H, singleRender=1, **kw)
            renderer.upload_scene(s.scene if isinstance(s, EditedScene) else s)
            if isinstance(s, EditedScene):
                s.push(renderer, "objects")      # the edited samplerIndex
            renderer.upload_textures(tex)
            renderer.reset_counters()
            img = renderer.render(pc, W, H)
            cnt = renderer.counters()
            pyoracle.set_textures(tex)
            ref, rc = pyoracle.render(s, pc, W, H)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "textured pixels differ from the oracle"
            assert {k: cnt[k] for k in keys} == {k: rc[k] for k in keys}
            # and the texture is really in the picture
            renderer.upload_textures([])
            assert not np.array_equal(renderer.render(pc, W, H), img)
    finally:
        pyoracle.set_textures([])
        renderer.upload_textures([])
        renderer.set_tuning("pipeline", -1)
        renderer.set_tuning("pixel_refill", 0)


def _map_set(seed=5):
    """Generated maps: slot 0 the albedo checker, 1 alpha (holes), 2 metalness (stripes and a few faint texels), 3 bump (noise)."""
    rng = np.random.default_rng(seed)
    holes = np.where(rng.uniform(size=(12, 20)) < 0.35, rng.integers(0, 188, (12, 20)), rng.integers(188, 256, (12, 20)))
    stripes = np.where(np.arange(10)[None, :] % 3 == 0, rng.integers(1, 256, (7, 10)), 0)
    return [_checker(24, 16), _grey(20, 12, holes), _grey(10, 7, stripes), _grey(16, 16, rng.integers(0, 256, (16, 16)))]


KEYS = ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "emitterTests")


def _same_as_oracle(renderer, s, tex, pc, W, H, what, kernel=None, maps=True):
    renderer.upload_scene(s.scene)
    s.push(renderer, "objects")
    s.push(renderer, "materials")
    renderer.upload_textures(tex)
    renderer.reset_counters()
    img = renderer.render(pc, W, H)
    cnt = renderer.counters()
    pyoracle.set_textures(tex)
    ref, rc = pyoracle.render(s, pc, W, H)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{what}: pixels differ from the oracle's ({renderer.last_kernel()})"
    assert {k: cnt[k] for k in KEYS} == {k: rc[k] for k in KEYS}, what
    assert rc["lightQueryMismatch"] == 0
    assert not maps or renderer.last_pipeline() == 0, "scenes that bind these maps belong to the multi-kernel pipeline"
    if kernel:
        assert renderer.last_kernel() == kernel, (what, renderer.last_kernel())
    return img


@pytest.mark.gpu
def test_metalness_alpha_bump_maps_against_the_oracle(renderer, tmp_path):
    """HIP path == oracle, pixels and counters, with each of the three maps alone and all together, under both samplers, with uvs
    inside and beyond the unit square, a forced fused pipeline (ignored: the maps' kernels are multi-kernel ones), heat maps, an
    emitter with holes (which leaves the emitter list), several frames in overlapping parts, and per-ray hit records."""
    tex = _map_set()
    W, H = 112, 84
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3, environmentOn=True)
    try:
        n = 0
        for sampler in (0, 1):
            for uvs in (1.0, 2.5):
                for slots in (dict(alphaIndex=1), dict(metalnessIndex=2), dict(bumpIndex=3), dict(alphaIndex=1, metalnessIndex=2, bumpIndex=3),
                              dict(alphaIndex=3, metalnessIndex=1, bumpIndex=2, albedoIndex=-1)):
                    n += 1
                    s = _bound(tmp_path / f"c{n}", sampler, uvs, **slots)
                    kern = "k_trace_pw_alpha<false>" if "alphaIndex" in slots else None
                    img = _same_as_oracle(renderer, s, tex, pc, W, H, f"sampler {sampler} uv x{uvs} {slots}", kern)
                    if n <= 5:   # and the maps are really in the picture
                        renderer.upload_textures(tex[:1])
                        assert not np.array_equal(renderer.render(pc, W, H), img), slots
        # slots beyond the uploaded table bind nothing (the ordinary kernels run); one-texel maps; a map narrower than two texels
        s = _bound(tmp_path / "x1", 1, 2.5, alphaIndex=7, metalnessIndex=5, bumpIndex=63)
        _same_as_oracle(renderer, s, tex, pc, W, H, "slots out of range", maps=False)
        assert "alpha" not in renderer.last_kernel()
        tiny = [tex[0], _grey(1, 1, 200), _grey(1, 1, 90), _grey(1, 3, np.array([[10], [120], [250]]))]
        _same_as_oracle(renderer, _bound(tmp_path / "x2", 0, 2.5, alphaIndex=1, metalnessIndex=2, bumpIndex=3), tiny, pc, W, H, "one-texel maps", "k_trace_pw_alpha<false>")
        _same_as_oracle(renderer, _bound(tmp_path / "x3", 1, 2.5, alphaIndex=1, metalnessIndex=2, bumpIndex=3), tiny, pc, W, H, "one-texel maps, clamped")
        full = dict(alphaIndex=1, metalnessIndex=2, bumpIndex=3)
        # "pipeline" 1 is overruled; heat maps count per pixel through k_trace_pw_alpha<true>
        renderer.set_tuning("pipeline", 1)
        _same_as_oracle(renderer, _bound(tmp_path / "f1", 0, 2.5, **full), tex, pc, W, H, "forced fused")
        renderer.set_tuning("pipeline", -1)
        for dbg in (0, 1, 2):
            pcd = engine.push_constants(W, H, singleRender=1, sampleLimit=2, environmentOn=True, debug=dbg, boxCap=300, triangleCap=40)
            _same_as_oracle(renderer, _bound(tmp_path / f"d{dbg}", 1, 2.5, **full), tex, pcd, W, H, f"heat map {dbg}", "k_trace_pw_alpha<true>")
        # an emissive material with holes: light queries are traversed in full (no emitter list); and one without the alpha map beside it
        for slots in (dict(full, emissionStrength=1.5), dict(metalnessIndex=2, bumpIndex=3, emissionStrength=1.5)):
            s = _bound(tmp_path / f"e{len(slots)}", 0, 1.0, **slots)
            mi = _plane_material(s)
            s.materials[mi].emissionColor[:] = [1.0, 0.8, 0.6]
            _same_as_oracle(renderer, s, tex, pc, W, H, f"emissive {slots}")
        # per-ray hit records: distance, triangle, point and the bumped normal
        s = _bound(tmp_path / "h", 1, 2.5, **full)
        renderer.upload_scene(s.scene); s.push(renderer, "objects"); s.push(renderer, "materials"); renderer.upload_textures(tex)
        pyoracle.set_textures(tex)
        o, d = seeded_rays(4096, seed=77)
        g, c = engine.hits_to_numpy(renderer.trace_rays(o, d)), engine.hits_to_numpy(pyoracle.trace_rays(s, o, d))
        assert_hits_equal(g, c)
        pyoracle.set_textures(tex[:1])
        assert not np.array_equal(engine.hits_to_numpy(pyoracle.trace_rays(s, o, d))["dst"], c["dst"])
        pyoracle.set_textures(tex)
        # progressive frames in one dispatch, in overlapping parts
        for k, v in (("lanes", 3), ("lanes_min_kslots", 1)):
            renderer.set_tuning(k, v)
        pcp = engine.push_constants(W, H, raysPerPixel=2, progressive=1, environmentOn=True)
        renderer.clear_framebuffer(); pcp.frameCount = 0
        img = renderer.render_frames(pcp, W, H, 3)
        assert renderer.last_parts() == 3
        prev = None
        for f in range(3):
            pcp.frameCount = f
            prev, _ = pyoracle.render(s, pcp, W, H, prev=prev)
        assert np.array_equal(img.view(np.uint32), prev.view(np.uint32)), "frames in flight with maps differ from the oracle's"
    finally:
        pyoracle.set_textures([])
        renderer.upload_textures([])
        for k, v in (("pipeline", -1), ("lanes", 0), ("lanes_min_kslots", 1024)):
            renderer.set_tuning(k, v)


@pytest.mark.gpu
def test_alpha_map_on_a_deep_bvh_with_placed_objects(renderer):
    """k_trace_pw_alpha's overflow stack (leaves deeper than its 24 LDS entries) and its object culling: a lopsided mesh of depth
    > 24 with uvs and all three maps, next to two placed copies of a small mesh with maps of their own sampler."""
    from test_instantiations import skewed, soup, _normals  # noqa: F401
    rng = np.random.default_rng(9)
    tex = _map_set(11)
    s = engine.Scene()
    glow = s.add_material(engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 0.9, 0.8), emissionStrength=3.0))
    mapped = engine.default_material(albedo=(0.8, 0.7, 0.6))
    mapped.albedoIndex, mapped.alphaIndex, mapped.metalnessIndex, mapped.bumpIndex = 0, 1, 2, 3
    mapped = s.add_material(mapped)
    holes = engine.default_material(albedo=(0.3, 0.7, 0.4))
    holes.alphaIndex = 1
    holes = s.add_material(holes)
    tri, nrm = skewed(100000, 4, 5)
    s.add_mesh("deep", tri, nrm, engine.placement(), mapped, uvs=rng.uniform(-1.5, 2.5, (tri.shape[0], 6)).astype(np.float32))
    assert s.last_bvh_stats()["maxDepth"] > 24
    quad = np.array([[[-0.3, -1.5, -0.3], [0.3, -1.5, -0.3], [0.3, -1.5, 0.3]], [[-0.3, -1.5, -0.3], [0.3, -1.5, 0.3], [-0.3, -1.5, 0.3]]], np.float32)
    s.add_mesh("light", quad, np.tile(np.array([0, 1, 0], np.float32), (2, 3, 1)), engine.placement(), glow)
    t2, n2 = soup(60, 77, 0.15)
    uv2 = rng.uniform(0, 1, (60, 6)).astype(np.float32)
    s.add_mesh("placed_a", t2, n2, engine.placement(position=(0.5, 0.1, 0.2), rotation=(20, 35, 10), scale=(0.4, 0.5, 0.4), samplerIndex=1), holes, uvs=uv2)
    s.add_mesh("placed_b", t2, n2, engine.placement(position=(-0.5, 0.0, -0.1), rotation=(-15, 70, 5), scale=(0.5, 0.4, 0.6)), mapped, uvs=uv2)
    W, H = 96, 72
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, bounceLimit=5, environmentOn=True)
    try:
        renderer.upload_scene(s)
        renderer.upload_textures(tex)
        pyoracle.set_textures(tex)
        renderer.reset_counters()
        img = renderer.render(pc, W, H)
        cnt = renderer.counters()
        ref, rc = pyoracle.render(s, pc, W, H)
        assert renderer.last_kernel() == "k_trace_pw_alpha<false>"
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        assert {k: cnt[k] for k in KEYS} == {k: rc[k] for k in KEYS}
        renderer.upload_textures([])
        assert not np.array_equal(renderer.render(pc, W, H), img)
    finally:
        pyoracle.set_textures([])
        renderer.upload_textures([])
