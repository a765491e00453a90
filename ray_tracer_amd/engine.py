"""Host-side mirror of the reference engine's path-tracing surface.

`Scene` keeps the reference's scene vectors and builders (VulkanEngine::
read_obj / read_mtl / cornell_box / prepare_storage_buffers,
src/vk_engine.cpp:638-758, 800-1167) behind the C ABI's scene half.
`Renderer` is the device half: `upload_scene` = the copy_buffer calls
(src/vk_engine.cpp:753-757), `update_*` = update_buffer (:1446-1475),
`run_compute` = run_compute (:1623-1676) with the same frame semantics
(`totalSamples < sampleLimit`, single render = one dispatch of sampleLimit spp,
otherwise raysPerPixel spp per dispatch and frameCount advancing only when
progressive, :1782,1812-1814).

Everything numeric happens in librt_amd.so; this file only moves pointers.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (BVHNode, PushConstants, RayMaterial, RenderObject, RtCounters, RtHit, RtPlacement,
                    RtSceneArrays, RtTexture, Sphere, Triangle, TrianglePoint)

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


class RtError(RuntimeError):
    pass


def placement(position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), samplerIndex=0, frontOnly=False):
    """ImGuiObject's placement fields (src/vk_engine.h:125-132)."""
    p = RtPlacement()
    _capi.lib().rt_placement_default(C.byref(p))
    if np.isscalar(scale):
        scale = (scale,) * 3
    p.position[:] = [float(x) for x in position]
    p.rotation[:] = [float(x) for x in rotation]
    p.scale[:] = [float(x) for x in scale]
    p.samplerIndex = int(samplerIndex)
    p.frontOnly = 1 if frontOnly else 0
    return p


def default_material(**kw):
    m = RayMaterial()
    _capi.lib().rt_material_default(C.byref(m))
    for k, v in kw.items():
        if k in ("albedo", "emissionColor"):
            getattr(m, k)[:] = [float(x) for x in v]
        else:
            setattr(m, k, v)
    return m


def push_constants(width, height, **params):
    """PushConstants with the reference's defaults (src/vk_engine.h:145-171,325).
    Keyword names are the reference's field names; `cameraAngles` (degrees)
    recomputes cameraRotation as run_compute does."""
    pc = PushConstants()
    l = _capi.lib()
    l.rt_push_constants_default(C.byref(pc), width, height)
    for k, v in params.items():
        if k == "cameraAngles":
            ang = (C.c_float * 3)(*[float(x) for x in v])
            l.rt_camera_rotation(ang, pc.camInfo.cameraRotation)
        elif k in ("pos",):
            pc.camInfo.pos[:] = [float(x) for x in v]
        elif k in ("nearPlane", "aspectRatio", "fov"):
            setattr(pc.camInfo, k, float(v))
        elif k in ("horizonColor", "zenithColor", "groundColor", "lightDir"):
            getattr(pc.environment, k)[:] = [float(x) for x in v]
        elif k == "environmentOn":
            pc.environment.lightDir[3] = 1.0 if v else 0.0
        elif k == "frameCount":
            pc.frameCount = int(v)
        elif hasattr(pc.rayTraceParams, k):
            setattr(pc.rayTraceParams, k, int(v))
        else:
            raise KeyError(k)
    return pc


class Scene:
    """The reference's CPU-side scene: spheres, rayMaterials, triPoints,
    triangles, objects, bvhNodes (src/vk_engine.h:270-283)."""

    def __init__(self):
        self._l = _capi.lib()
        h = C.c_void_p()
        if self._l.rt_scene_create(C.byref(h)) != 0:
            raise RtError("rt_scene_create failed")
        self._h = h

    def close(self):
        if self._h:
            self._l.rt_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise RtError(f"{what}: {self._l.rt_scene_last_error(self._h).decode()}")
        return rc

    # -- builders ---------------------------------------------------------
    def add_material(self, m):
        return self._check(self._l.rt_scene_add_material(self._h, C.byref(m)), "add_material")

    def set_sphere(self, i, position, radius, materialIndex):
        pos = (C.c_float * 3)(*[float(x) for x in position])
        self._check(self._l.rt_scene_set_sphere(self._h, i, pos, float(radius), int(materialIndex)), "set_sphere")

    def read_obj(self, filePath, imGuiObj=None, material=0):
        p = imGuiObj if imGuiObj is not None else placement()
        return self._check(self._l.rt_scene_read_obj(self._h, os.fsencode(filePath), C.byref(p), int(material)),
                           "read_obj")

    def read_mtl(self, filePath):
        return self._check(self._l.rt_scene_read_mtl(self._h, os.fsencode(filePath)), "read_mtl")

    def add_mesh(self, key, positions, normals, imGuiObj=None, material=0, uvs=None):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 9)
        assert pos.shape == nrm.shape
        fp = C.POINTER(C.c_float)
        uvp = None
        if uvs is not None:
            uvs = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 6)
            uvp = uvs.ctypes.data_as(fp)
        p = imGuiObj if imGuiObj is not None else placement()
        return self._check(self._l.rt_scene_add_mesh(self._h, key.encode(), pos.ctypes.data_as(fp),
                                                     nrm.ctypes.data_as(fp), uvp, pos.shape[0], C.byref(p),
                                                     int(material)), "add_mesh")

    def cornell_box(self, assetDir=ASSET_DIR):
        self._check(self._l.rt_scene_cornell_box(self._h, os.fsencode(assetDir)), "cornell_box")

    def prepare_storage_buffers(self, assetDir=ASSET_DIR):
        self._check(self._l.rt_scene_prepare_default(self._h, os.fsencode(assetDir)), "prepare_storage_buffers")

    def use_device_bvh(self, renderer):
        """Meshes added from now on get their BVH from the GPU builder of `renderer` (rt_bvh_hook):
        the same nodes, numbering and triangle order as the host builder. None switches back."""
        if renderer is None:
            self._check(self._l.rt_scene_set_bvh_hook(self._h, None, None), "rt_scene_set_bvh_hook")
        else:
            fn = C.cast(self._l.rt_bvh_hook, C.c_void_p)
            self._check(self._l.rt_scene_set_bvh_hook(self._h, fn, renderer._h), "rt_scene_set_bvh_hook")
        self._bvh_renderer = renderer  # keep it alive as long as the hook points at it

    def texture_paths(self):
        """Image file of every texture slot the MTL files (or add_texture) claimed, in slot order."""
        n = self._l.rt_scene_texture_count(self._h)
        return [os.fsdecode(self._l.rt_scene_texture_path(self._h, i)) for i in range(n)]

    def add_texture(self, path):
        """Claims the next texture slot for an image file; bind it through a material's albedoIndex."""
        return self._check(self._l.rt_scene_add_texture(self._h, os.fsencode(path)), "add_texture")

    def set_material(self, i, m):
        self._check(self._l.rt_scene_set_material(self._h, int(i), C.byref(m)), "set_material")

    def material(self, i):
        a = self.arrays()
        m = RayMaterial()
        C.memmove(C.byref(m), C.byref(a.materials[i]), C.sizeof(RayMaterial))
        return m

    def find_material(self, key):
        return self._l.rt_scene_find_material(self._h, key.encode())

    def last_bvh_stats(self):
        v = [C.c_uint32() for _ in range(4)]
        self._l.rt_scene_last_bvh_stats(self._h, *[C.byref(x) for x in v])
        return dict(zip(("nodeCount", "maxDepth", "minDepth", "maxTri"), [x.value for x in v]))

    # -- views ------------------------------------------------------------
    def arrays(self):
        a = RtSceneArrays()
        self._check(self._l.rt_scene_get_arrays(self._h, C.byref(a)), "get_arrays")
        return a

    def _view(self, ptr, n, ctype):
        if n == 0:
            return np.zeros((0,), dtype=np.uint8)
        buf = (ctype * n).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=np.uint8).reshape(n, C.sizeof(ctype))

    def numpy(self):
        """Raw byte views (copied) of the six scene arrays, for tests."""
        a = self.arrays()
        return {
            "spheres": self._view(a.spheres, a.sphereCount, Sphere).copy(),
            "materials": self._view(a.materials, a.materialCount, RayMaterial).copy(),
            "triPoints": self._view(a.triPoints, a.triPointCount, TrianglePoint).copy(),
            "triangles": self._view(a.triangles, a.triangleCount, Triangle).copy(),
            "objects": self._view(a.objects, a.objectCount, RenderObject).copy(),
            "bvhNodes": self._view(a.bvhNodes, a.bvhNodeCount, BVHNode).copy(),
        }

    def counts(self):
        a = self.arrays()
        return dict(spheres=a.sphereCount, materials=a.materialCount, triPoints=a.triPointCount,
                    triangles=a.triangleCount, objects=a.objectCount, bvhNodes=a.bvhNodeCount)


class Renderer:
    """One MI355X: device buffers + the wavefront pipeline (rt_ctx)."""

    def __init__(self, device=0):
        self._l = _capi.lib()
        h = C.c_void_p()
        rc = self._l.rt_create(int(device), C.byref(h))
        if rc != 0:
            raise RtError(f"rt_create(device={device}) failed with {rc}: no usable HIP device "
                          "(this path has no CPU fallback)")
        self._h = h
        self.device = device
        self.totalSamples = 0
        self._frameNumber = 0
        self._scene = None
        self._shape = None

    def close(self):
        if getattr(self, "_h", None):
            self._l.rt_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {self._l.rt_last_error(self._h).decode()}")

    def set_stream(self, hip_stream_ptr):
        self._check(self._l.rt_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "rt_set_stream")

    def upload_scene(self, scene):
        a = scene.arrays()
        self._check(self._l.rt_upload_scene(self._h, C.byref(a)), "rt_upload_scene")
        self._scene = scene
        self._counts = scene.counts()

    def upload_textures(self, images):
        """The scene's texture table: a list of uint8 arrays [h, w, 4] in slot order (rt_upload_textures); [] removes it."""
        arr = (RtTexture * max(len(images), 1))()
        keep = []
        for i, im in enumerate(images):
            im = np.ascontiguousarray(im, dtype=np.uint8)
            if im.ndim != 3 or im.shape[2] != 4:
                raise ValueError("textures are [height, width, 4] uint8 (RGBA)")
            keep.append(im)
            arr[i].width, arr[i].height = im.shape[1], im.shape[0]
            arr[i].rgba8 = im.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self._l.rt_upload_textures(self._h, arr, len(images)), "rt_upload_textures")

    def update_materials(self, scene):
        a = scene.arrays()
        self._check(self._l.rt_update_materials(self._h, a.materials, a.materialCount), "rt_update_materials")

    def update_spheres(self, scene):
        a = scene.arrays()
        self._check(self._l.rt_update_spheres(self._h, a.spheres, a.sphereCount), "rt_update_spheres")

    def update_objects(self, scene):
        a = scene.arrays()
        self._check(self._l.rt_update_objects(self._h, a.objects, a.objectCount), "rt_update_objects")

    def fill_counts(self, pc):
        """rayTracerParams.sphereCount/objectCount as run_compute sets them (:1655-1656)."""
        pc.rayTraceParams.sphereCount = self._counts["spheres"]
        pc.rayTraceParams.objectCount = self._counts["objects"]
        return pc

    def render(self, pc, width, height, row0=0, rowStride=1, nRows=None, out_ptr=None, sync=True):
        """One dispatch. Returns an (nRows, width, 4) float32 array unless the
        caller supplied a device pointer."""
        if nRows is None:
            nRows = (height - row0 + rowStride - 1) // rowStride
        self.fill_counts(pc)
        self._check(self._l.rt_render(self._h, C.byref(pc), width, height, row0, rowStride, nRows,
                                      C.c_void_p(out_ptr) if out_ptr else None), "rt_render")
        self._shape = (nRows, width, 4)
        if not sync:
            return None
        self.sync()
        if out_ptr:
            return None
        return self.read_rgba()

    def render_frames(self, pc, width, height, nFrames, row0=0, rowStride=1, nRows=None, out_ptr=None, sync=True):
        """nFrames progressive dispatches starting at pc.frameCount (rt_render_frames): the frames share launches where one
        frame would leave the GPU short of pixels. Same pixels as nFrames render() calls."""
        if nRows is None:
            nRows = (height - row0 + rowStride - 1) // rowStride
        self.fill_counts(pc)
        self._check(self._l.rt_render_frames(self._h, C.byref(pc), width, height, row0, rowStride, nRows, int(nFrames),
                                             C.c_void_p(out_ptr) if out_ptr else None), "rt_render_frames")
        self._shape = (nRows, width, 4)
        if not sync:
            return None
        self.sync()
        return None if out_ptr else self.read_rgba()

    def run_compute(self, pc, width, height, frames=1, **tile):
        """Frame semantics of draw()/run_compute (src/vk_engine.cpp:1782,1812-1814); `tile` = row0 / rowStride / nRows
        of rt_render when the frame is split over several GPUs. `frames` > 1 (progressive accumulation only): that many of
        the frames the loop would dispatch one after the other go in one rt_render_frames call — the same image, sooner."""
        t = pc.rayTraceParams
        if self.totalSamples >= t.sampleLimit:
            return None
        pc.frameCount = self._frameNumber
        n = 1
        if frames > 1 and t.progressive and not t.singleRender and t.raysPerPixel > 0:
            left = (t.sampleLimit - self.totalSamples + t.raysPerPixel - 1) // t.raysPerPixel
            n = max(1, min(int(frames), int(left)))
        img = self.render(pc, width, height, **tile) if n == 1 else self.render_frames(pc, width, height, n, **tile)
        if t.singleRender:
            self.totalSamples = t.sampleLimit
        else:
            self.totalSamples += t.raysPerPixel * n
        if t.progressive:
            self._frameNumber += n
        return img

    def sync(self):
        self._check(self._l.rt_sync(self._h), "rt_sync")

    def clear_framebuffer(self):
        """A new progressive history: the next render into the context's own image starts from zeros."""
        self._check(self._l.rt_clear_framebuffer(self._h), "rt_clear_framebuffer")

    def read_rgba(self):
        out = np.empty(self._shape, dtype=np.float32)
        self._check(self._l.rt_read_rgba_f32(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                    "rt_read_rgba_f32")
        return out

    def read_rgba8_srgb(self):
        out = np.empty(self._shape, dtype=np.uint8)
        self._check(self._l.rt_read_rgba8_srgb(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size),
                    "rt_read_rgba8_srgb")
        return out

    def trace_rays(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        hits = (RtHit * o.shape[0])()
        fp = C.POINTER(C.c_float)
        self._check(self._l.rt_trace_rays(self._h, o.shape[0], o.ctypes.data_as(fp), d.ctypes.data_as(fp), hits),
                    "rt_trace_rays")
        return hits

    def counters(self):
        c = RtCounters()
        self._check(self._l.rt_get_counters(self._h, C.byref(c)), "rt_get_counters")
        return {n: getattr(c, n) for n, _ in RtCounters._fields_}

    def reset_counters(self):
        self._check(self._l.rt_reset_counters(self._h), "rt_reset_counters")

    def set_profiling(self, on):
        self._check(self._l.rt_set_profiling(self._h, 1 if on else 0), "rt_set_profiling")

    def trace_time_ms(self):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._l.rt_get_trace_time_ms(self._h, C.byref(ms), C.byref(n)), "rt_get_trace_time_ms")
        return ms.value, n.value

    def trace_busy_ms(self):
        """Time during which at least one traversal launch ran (union of the launches' spans) since set_profiling(True)."""
        ms = C.c_double()
        self._check(self._l.rt_get_trace_busy_ms(self._h, C.byref(ms)), "rt_get_trace_busy_ms")
        return ms.value

    def set_tuning(self, key, value):
        self._check(self._l.rt_set_tuning(self._h, key.encode(), int(value)), "rt_set_tuning")

    def last_kernel(self):
        """Traversal kernel instantiation of the last launch, e.g. 'k_trace_pw<20, false, false, false, false, 144, 5>'."""
        return self._l.rt_last_kernel(self._h).decode()

    def last_parts(self):
        """Parts (streams) the last multi-kernel dispatch ran in."""
        return self._l.rt_last_parts(self._h)

    def last_pipeline(self):
        """0 = multi-kernel wavefront pipeline, 1 = wave-private fused pipeline."""
        return self._l.rt_last_pipeline(self._h)

    def ray_cost(self):
        """Measured box tests per executed ray of the uploaded scene (< 0: not measured yet)."""
        return self._l.rt_ray_cost(self._h)

    def bvh_last_build_ms(self):
        return self._l.rt_bvh_last_build_ms(self._h)

    def selftest(self):
        b = C.c_uint32()
        self._check(self._l.rt_device_selftest(self._h, C.byref(b)), "rt_device_selftest")
        return b.value

    # -- several GPUs, one process each: the final gather over RCCL through the C ABI (rt_comm_*) --------------
    @staticmethod
    def comm_unique_id():
        """On rank 0: the RT_COMM_ID_BYTES every rank passes to comm_init (hand them over by any means)."""
        buf = C.create_string_buffer(128)
        if _capi.lib().rt_comm_unique_id(buf) != 0:
            raise RtError("rt_comm_unique_id failed: RCCL could not be loaded")
        return buf.raw

    def comm_init(self, unique_id, n_ranks, rank):
        self._check(self._l.rt_comm_init(self._h, C.c_char_p(unique_id), int(n_ranks), int(rank)), "rt_comm_init")

    def comm_destroy(self):
        self._check(self._l.rt_comm_destroy(self._h), "rt_comm_destroy")

    def gather_strips(self, strip_ptr, width, height, root=0, frame_ptr=None):
        """This rank's strip (device pointer) -> the whole frame on `root` (device pointer there, None elsewhere)."""
        self._check(self._l.rt_gather_strips(self._h, C.c_void_p(strip_ptr), width, height, int(root),
                                             C.c_void_p(frame_ptr) if frame_ptr else None), "rt_gather_strips")

    def deinterleave_strips(self, strips_ptr, width, height, n_ranks, frame_ptr):
        """Strips of ranks 0..n_ranks-1 stored one after the other (device pointer) -> the frame's rows (device pointer)."""
        self._check(self._l.rt_deinterleave_strips(self._h, C.c_void_p(strips_ptr), width, height, int(n_ranks), C.c_void_p(frame_ptr)),
                    "rt_deinterleave_strips")

    def deinterleave_strips_host(self, strips, n_ranks):
        """[H, W, 4] float32 strips of ranks 0..n_ranks-1 one after the other -> the [H, W, 4] frame (through the device kernel)."""
        strips = np.ascontiguousarray(strips, np.float32)
        out = np.empty_like(strips)
        self._check(self._l.rt_deinterleave_strips_host(self._h, strips.ctypes.data_as(C.c_void_p), strips.shape[1], strips.shape[0], int(n_ranks),
                                                        out.ctypes.data_as(C.c_void_p)), "rt_deinterleave_strips_host")
        return out

    def math_probe(self, inputs):
        """include/rt_probe.h evaluated on the device: [n, 32] float32 -> [n, 64] float32."""
        x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 32)
        out = np.zeros((x.shape[0], 64), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self._l.rt_device_math_probe(self._h, x.shape[0], x.ctypes.data_as(fp), out.ctypes.data_as(fp)),
                    "rt_device_math_probe")
        return out

    def copy_bandwidth_gbps(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        self._check(self._l.rt_measure_copy_bandwidth(self._h, nbytes, iters, C.byref(g)), "copy bandwidth")
        return g.value


def hits_to_numpy(hits):
    """RtHit array -> dict of numpy arrays."""
    n = len(hits)
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(n, C.sizeof(RtHit))
    f = raw.view(np.float32).reshape(n, -1)
    u = raw.view(np.uint32).reshape(n, -1)
    return dict(dst=f[:, 0].copy(), didHit=u[:, 1].copy(), isSphere=u[:, 2].copy(), objectHitIndex=u[:, 3].copy(),
                triHitIndex=u[:, 4].copy(), materialIndex=u[:, 5].copy(), frontFace=u[:, 6].copy(),
                hitPoint=f[:, 7:10].copy(), normal=f[:, 10:13].copy(), boxTests=u[:, 13].copy(),
                triTests=u[:, 14].copy())


def load_textures(scene):
    """Decodes the scene's texture files to RGBA8 the way the reference does (stbi_load(..., STBI_rgb_alpha),
    src/vk_textures.cpp:103-113: 4 channels, rows top to bottom), with PIL. A missing file raises (the reference exits)."""
    from PIL import Image
    out = []
    for path in scene.texture_paths():
        with Image.open(path) as im:
            out.append(np.asarray(im.convert("RGBA"), dtype=np.uint8))
    return out
