"""Host scene surface: read_obj / read_mtl / cornell_box /
prepare_storage_buffers / build_bvh semantics of src/vk_engine.cpp."""
import os

import numpy as np
import pytest

from oracle import scene_ref
from ray_tracer_amd import engine

A = engine.ASSET_DIR


def _nodes(scene):
    raw = scene.numpy()["bvhNodes"]
    return raw.view(np.float32)[:, :6].copy(), raw.view(np.uint32)[:, 6].copy(), raw.view(np.uint32)[:, 7].copy()


def test_default_cornell_invariants():
    """51 triangles, 153 unshared points, 9 objects, 4 unique meshes, 6 built-in materials."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    assert s.counts() == dict(spheres=10, materials=6, triPoints=153, triangles=51, objects=9, bvhNodes=44)
    n = s.numpy()
    obj = n["objects"].view(np.uint32)
    # cube, cube2 (instance), light2, plane x3 (instances), ceiling, plane x2
    assert list(obj[:, 17]) == [0, 0, 7, 12, 12, 12, 13, 12, 12]
    assert list(obj[:, 18]) == [0, 0, 3, 0, 2, 1, 0, 0, 0]   # left wall green (2), right wall red (1)
    mats = n["materials"].view(np.float32)
    assert list(mats[3, 4:8]) == [1, 1, 1, np.float32(2.4)] and list(mats[3, :3]) == [0, 0, 0]
    assert mats[4, 8] == 1.0 and mats[5, 9] == 2.0 and mats[0, 9] == -1.0
    tris = n["triangles"].view(np.uint32)
    assert set(tris[:, :3].ravel()) == set(range(153))            # every corner its own point
    assert tris[:12, 3].sum() == 0 and tris[12:, 3].all()         # cubes two-sided, the rest frontOnly
    sph = n["spheres"].view(np.float32)
    assert not sph.any()                                          # ten zero-radius spheres at the origin


def test_transform_is_T_Rx_Ry_Rz_S():
    s = engine.Scene()
    s.prepare_storage_buffers()
    M = s.numpy()["objects"].view(np.float32)[0, :16].reshape(4, 4).T   # cube: s 0.25, ry -30, t (-0.4,0.25,-0.45)
    c, sn = np.cos(np.radians(-30.0)), np.sin(np.radians(-30.0))
    R = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]])
    assert np.allclose(M[:3, :3], R * 0.25, atol=1e-6)
    assert np.allclose(M[:3, 3], [-0.4, 0.25, -0.45]) and np.allclose(M[3], [0, 0, 0, 1])


def test_obj_parser_quirks(tmp_path):
    # '#' lines are skipped (light2.obj uses that to delete faces); a trailing blank on an 'f' line is tolerated
    s = engine.Scene()
    s.add_material(engine.default_material())
    s.read_obj(os.path.join(A, "light2.obj"))
    assert s.counts()["triangles"] == 10
    s.read_obj(os.path.join(A, "plane.obj"))
    assert s.counts()["triangles"] == 12
    # a missing file is a silent no-op, like the reference (:834)
    assert s.read_obj(str(tmp_path / "nope.obj")) == 1
    assert s.counts()["objects"] == 2
    # without normals every corner re-reads the first token and the normal is zero (:906,921-922)
    p = tmp_path / "nonormal.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\n")
    s.read_obj(str(p))
    tp = s.numpy()["triPoints"].view(np.float32)[-3:]
    assert (tp[:, :3] == 0).all() and (tp[:, 4:7] == 0).all()
    # uv goes to position.w / normal.w
    q = tmp_path / "uv.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0.25 0.75\nvt 1 0\nvt 0 1\nf 1/1/1 2/2/1 3/3/1\n")
    s.read_obj(str(q))
    tp = s.numpy()["triPoints"].view(np.float32)[-3:]
    assert tp[0, 3] == 0.25 and tp[0, 7] == 0.75 and list(tp[1, :3]) == [1, 0, 0]
    # empty group: undefined in the reference, an error here
    e = tmp_path / "empty.obj"
    e.write_text("v 0 0 0\n")
    with pytest.raises(engine.RtError):
        s.read_obj(str(e))


def test_mtl_semantics_and_usemtl_groups():
    s = engine.Scene()
    s.prepare_storage_buffers()
    base = s.counts()
    path = os.path.join(A, "bobadog", "bobadog.obj")
    s.read_obj(path, engine.placement(scale=0.3), 0)
    c = s.counts()
    groups = sum(1 for l in open(path) if l.startswith("usemtl"))
    assert c["objects"] - base["objects"] == groups              # one RenderObject + BVH per usemtl group
    assert c["triangles"] - base["triangles"] == sum(1 for l in open(path) if l.startswith("f "))
    mtl = os.path.join(A, "bobadog", "bobadog.mtl")
    i = s.find_material(mtl + "/Bobadog")
    assert i >= 6
    m = s.numpy()["materials"]
    alb = m.view(np.float32)[i, :3]
    assert np.allclose(alb, [0.496933, 0.076185, 0.428691])      # Ka (1,1,1) * Kd
    assert m.view(np.int32)[i, 10] >= 0                          # map_Ka claimed a texture slot
    # instancing of a multi-material file re-uses only the last group's BVH (:985,1022)
    before = s.counts()
    s.read_obj(path, engine.placement(position=(1, 0, 0)), 2)
    after = s.counts()
    assert after["objects"] == before["objects"] + 1 and after["triangles"] == before["triangles"]
    # Blender's map_Bump is skipped (case-sensitive match), map_Kd is taken
    t = engine.Scene()
    t.read_mtl(os.path.join(A, "test_plane.mtl"))
    mm = t.numpy()["materials"]
    assert np.allclose(mm.view(np.float32)[0, :3], [0.8, 0.8, 0.8])
    assert mm.view(np.int32)[0, 10] == 0 and mm.view(np.int32)[0, 13] == -1


def _check_bvh_invariants(scene):
    bounds, index, count = _nodes(scene)
    n = scene.numpy()
    tris = n["triangles"].view(np.uint32)
    pts = n["triPoints"].view(np.float32)
    roots = sorted(set(n["objects"].view(np.uint32)[:, 17]))
    seen = np.zeros(len(tris), int)
    for root in roots:
        stack = [(root, 0)]
        while stack:
            i, d = stack.pop()
            assert d <= 64
            if count[i]:
                sl = slice(index[i], index[i] + count[i])
                seen[sl] += 1
                p = pts[tris[sl, :3].ravel(), :3]
                assert np.array_equal(bounds[i, 0::2], p.min(0)) and np.array_equal(bounds[i, 1::2], p.max(0))
            else:
                l, r = index[i], index[i] + 1
                for c in (l, r):  # children inside the parent
                    assert (bounds[c, 0::2] >= bounds[i, 0::2]).all() and (bounds[c, 1::2] <= bounds[i, 1::2]).all()
                stack += [(l, d + 1), (r, d + 1)]
    assert (seen == 1).all()


def test_bvh_invariants_cornell_bunny_klein():
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.read_obj(os.path.join(A, "bunny.obj"), engine.placement(scale=0.7, position=(0, 0.53, 0)), 0)
    st = s.last_bvh_stats()
    assert st["maxTri"] >= 1 and 8 <= st["maxDepth"] <= 64
    s.read_obj(os.path.join(A, "klein_bottle.obj"), engine.placement(scale=0.5), 4)
    _check_bvh_invariants(s)


@pytest.mark.parametrize("name", ["cube.obj", "ceiling.obj", "light2.obj", "bunny.obj"])
def test_builder_matches_python_restatement(name):
    """C++ read_obj + build_bvh against oracle/scene_ref.py: same nodes, same triangle order, bit for bit."""
    path = os.path.join(A, name)
    s = engine.Scene()
    s.add_material(engine.default_material())
    s.read_obj(path)
    tp, tn, tu = scene_ref.parse_obj(path)
    nodes, order = scene_ref.build_bvh(tp)
    n = s.numpy()
    tris = n["triangles"].view(np.uint32)
    pts = n["triPoints"].view(np.float32)
    got_pos = pts[tris[:, :3].ravel(), :3].reshape(-1, 3, 3)
    got_nrm = pts[tris[:, :3].ravel(), 4:7].reshape(-1, 3, 3)
    assert np.array_equal(got_pos.view(np.uint32), tp[order].view(np.uint32))
    assert np.array_equal(got_nrm.view(np.uint32), tn[order].view(np.uint32))
    bounds, index, count = _nodes(s)
    assert len(nodes) == len(bounds)
    ref_b = np.array([[float(x) for x in nd[:6]] for nd in nodes], np.float32)
    assert np.array_equal(ref_b.view(np.uint32), bounds.view(np.uint32))
    assert [nd[6] for nd in nodes] == list(index) and [nd[7] for nd in nodes] == list(count)


def test_camera_rotation_and_defaults():
    pc = engine.push_constants(1728, 1117)
    R = np.array(list(pc.camInfo.cameraRotation), np.float32).reshape(4, 4).T
    c, s_ = np.cos(np.radians(4.0)), np.sin(np.radians(4.0))
    # rotX with the column constructor of src/vk_engine.cpp:1636-1640: columns (1,0,0),(0,c,-s),(0,s,c)
    assert np.allclose(R[:3, :3], np.array([[1, 0, 0], [0, c, s_], [0, -s_, c]]), atol=1e-7)
    assert np.isclose(pc.camInfo.aspectRatio, 1728 / 1117) and pc.camInfo.fov == 50 and pc.camInfo.nearPlane == np.float32(0.1)
    assert list(pc.camInfo.pos) == [0, -0.5, -3.5]
    t = pc.rayTraceParams
    assert (t.raysPerPixel, t.bounceLimit, t.sampleLimit, t.boxCap, t.triangleCap, t.debug) == (1, 8, 10, 200, 50, -1)
    assert pc.environment.lightDir[3] == 0 and np.isclose(np.linalg.norm(list(pc.environment.lightDir)[:3]), 1, atol=1e-6)
