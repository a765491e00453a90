"""Shared helpers for the parity tests."""
import numpy as np

from ray_tracer_amd import engine, scenes


def cornell_scene(spheres=True):
    s, _ = scenes.cornell(spheres)
    return s


def model_scene(obj, material=0, scale=0.7, position=(0.0, 0.53, 0.0), spheres=False):
    """Default Cornell box + one of the small OBJ assets that ship in assets/."""
    import os
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.read_obj(os.path.join(engine.ASSET_DIR, obj), engine.placement(position=position, scale=scale, samplerIndex=1), material)
    if spheres:
        s.set_sphere(0, (0.55, 0.2, -0.5), 0.25, 5)
    return s


def seeded_rays(n, seed, box=1.2):
    """Rays that start inside/around the Cornell box and point everywhere."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-box, box, size=(n, 3)).astype(np.float32)
    o[:, 1] -= 0.5
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a few axis-aligned and zero-component directions (inf in invDir)
    d[0] = (0, 0, 1); d[1] = (1, 0, 0); d[2] = (0, -1, 0); d[3] = (0, 1, 0)
    o[4] = (0, -0.5, -3.5); d[4] = (0, 0, 1)
    return o, d.astype(np.float32)


def assert_hits_equal(a, b):
    """Two RtHit dict-of-arrays must agree bit for bit."""
    for k in a:
        x, y = a[k], b[k]
        if x.dtype == np.float32:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"field {k} differs at {np.argwhere(x.view(np.uint32) != y.view(np.uint32))[:5].tolist()}"
        else:
            assert np.array_equal(x, y), f"field {k} differs at {np.argwhere(x != y)[:5].tolist()}"


class EditedScene:
    """A Scene whose materials / spheres / objects arrays have been edited in place, the way the reference's ImGui panels
    edit the host vectors before "Update Buffer" (src/vk_engine.cpp:1536-1618): ctypes copies of the three arrays that the
    test mutates, handed to rt_update_* and, as the same RtSceneArrays, to the oracle."""

    def __init__(self, scene):
        import ctypes as C
        from ray_tracer_amd._capi import RayMaterial, RenderObject, Sphere
        self.scene = scene
        a = scene.arrays()
        self.materials = (RayMaterial * max(a.materialCount, 1))()
        self.objects = (RenderObject * max(a.objectCount, 1))()
        self.spheres = (Sphere * max(a.sphereCount, 1))()
        C.memmove(self.materials, a.materials, C.sizeof(RayMaterial) * a.materialCount)
        C.memmove(self.objects, a.objects, C.sizeof(RenderObject) * a.objectCount)
        C.memmove(self.spheres, a.spheres, C.sizeof(Sphere) * a.sphereCount)
        self.nMaterials, self.nObjects, self.nSpheres = a.materialCount, a.objectCount, a.sphereCount

    def arrays(self):
        import ctypes as C
        from ray_tracer_amd._capi import RayMaterial, RenderObject, Sphere
        a = self.scene.arrays()
        a.materials = C.cast(self.materials, C.POINTER(RayMaterial))
        a.objects = C.cast(self.objects, C.POINTER(RenderObject))
        a.spheres = C.cast(self.spheres, C.POINTER(Sphere))
        return a

    def counts(self):
        return self.scene.counts()

    def set_transform(self, i, placement):
        """objects[i].transformMatrix = T*Rx*Ry*Rz*S of the placement (src/vk_engine.cpp:1597-1601)."""
        import ctypes as C
        from ray_tracer_amd import _capi
        _capi.lib().rt_transform_matrix(C.byref(placement), self.objects[i].transformMatrix)

    def push(self, renderer, what):
        """update_buffer for one of the arrays (src/vk_engine.cpp:1545,1572,1603)."""
        l, h = renderer._l, renderer._h
        if what == "materials":
            renderer._check(l.rt_update_materials(h, self.materials, self.nMaterials), "rt_update_materials")
        elif what == "objects":
            renderer._check(l.rt_update_objects(h, self.objects, self.nObjects), "rt_update_objects")
        elif what == "spheres":
            renderer._check(l.rt_update_spheres(h, self.spheres, self.nSpheres), "rt_update_spheres")
        else:
            raise KeyError(what)


def random_emitter_scene(seed):
    """The bare Cornell box plus random small meshes and spheres, several of them emissive, under random placements: the
    stress scenes of the light queries (emitters in front of, behind and inside other things, emissive spheres, more than one
    emissive mesh). Returns (scene, push constants, width, height)."""
    rng = np.random.default_rng(4000 + seed)
    s = cornell_scene(False)
    mats = [0, 1, 2, 4, 5]
    for _ in range(2):
        mats.append(s.add_material(engine.default_material(albedo=tuple(rng.random(3)), emissionColor=tuple(rng.random(3)),
                                                            emissionStrength=float(rng.random() * 3 + 0.1))))
    for k in range(int(rng.integers(2, 6))):
        n = int(rng.integers(1, 9))
        tri = rng.uniform(-0.8, 0.8, (n, 3, 3)).astype(np.float32)
        tri[:, :, 1] -= 0.5
        tri[:, 1:, :] = tri[:, :1, :] + rng.uniform(-0.35, 0.35, (n, 2, 3)).astype(np.float32)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-9)
        pl = engine.placement(rotation=tuple(rng.uniform(-40, 40, 3)), scale=tuple(rng.uniform(0.6, 1.2, 3))) if rng.random() < 0.5 else engine.placement()
        s.add_mesh(f"e{seed}_{k}", tri, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32), pl, int(rng.choice(mats)))
    for i in range(int(rng.integers(0, 5))):
        s.set_sphere(i, tuple(rng.uniform(-0.7, 0.7, 3)), float(rng.uniform(0.05, 0.3)), int(rng.choice(mats)))
    W, H = 80, 60
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2, bounceLimit=int(rng.integers(2, 9)), environmentOn=bool(seed % 2))
    return s, pc, W, H
