"""ctypes binding of librt_amd.so (the C ABI declared in include/rt_amd.h).

The structs mirror the reference's host structs byte for byte
(src/vk_engine.h:49-79,117-123,145-189; sizes checked in tests/test_layout.py).
The library is built in-tree by `__graft_entry__.build()`; importing this
module never falls back to another implementation: if the .so is missing the
import raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: another build of the same ABI, for A/B runs of two builds on one box (tools/)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")


class Sphere(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("materialIndex", C.c_uint32),
                ("_pad", C.c_uint32 * 3)]


class Triangle(C.Structure):
    _fields_ = [("v0", C.c_uint32), ("v1", C.c_uint32), ("v2", C.c_uint32), ("frontOnly", C.c_uint32),
                ("binormal", C.c_float * 3), ("_pad0", C.c_float), ("tangent", C.c_float * 3), ("_pad1", C.c_float)]


class TrianglePoint(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("normal", C.c_float * 4)]


class RayMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("_pad0", C.c_float), ("emissionColor", C.c_float * 3),
                ("emissionStrength", C.c_float), ("reflectance", C.c_float), ("ior", C.c_float),
                ("albedoIndex", C.c_int32), ("metalnessIndex", C.c_int32), ("alphaIndex", C.c_int32),
                ("bumpIndex", C.c_int32), ("_pad1", C.c_uint32 * 2)]


class RenderObject(C.Structure):
    _fields_ = [("transformMatrix", C.c_float * 16), ("smoothShade", C.c_uint32), ("bvhIndex", C.c_uint32),
                ("materialIndex", C.c_uint32), ("samplerIndex", C.c_uint32)]


class BVHNode(C.Structure):
    _fields_ = [("boundsX", C.c_float * 2), ("boundsY", C.c_float * 2), ("boundsZ", C.c_float * 2),
                ("index", C.c_uint32), ("triCount", C.c_uint32)]


class CameraInfo(C.Structure):
    _fields_ = [("cameraRotation", C.c_float * 16), ("pos", C.c_float * 3), ("nearPlane", C.c_float),
                ("aspectRatio", C.c_float), ("fov", C.c_float), ("_pad", C.c_float * 2)]


class EnvironmentData(C.Structure):
    _fields_ = [("horizonColor", C.c_float * 4), ("zenithColor", C.c_float * 4), ("groundColor", C.c_float * 3),
                ("_pad0", C.c_float), ("lightDir", C.c_float * 4)]


class RayTracerData(C.Structure):
    _fields_ = [("progressive", C.c_uint32), ("singleRender", C.c_uint32), ("debug", C.c_int32),
                ("raysPerPixel", C.c_uint32), ("bounceLimit", C.c_uint32), ("sphereCount", C.c_uint32),
                ("objectCount", C.c_uint32), ("triangleCap", C.c_uint32), ("boxCap", C.c_uint32),
                ("sampleLimit", C.c_uint32)]


class PushConstants(C.Structure):
    _fields_ = [("camInfo", CameraInfo), ("environment", EnvironmentData), ("rayTraceParams", RayTracerData),
                ("frameCount", C.c_uint32), ("_pad", C.c_uint32)]


class RtPlacement(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 3), ("scale", C.c_float * 3),
                ("samplerIndex", C.c_uint32), ("frontOnly", C.c_uint32)]


class RtSceneArrays(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("sphereCount", C.c_uint32),
                ("materials", C.POINTER(RayMaterial)), ("materialCount", C.c_uint32),
                ("triPoints", C.POINTER(TrianglePoint)), ("triPointCount", C.c_uint32),
                ("triangles", C.POINTER(Triangle)), ("triangleCount", C.c_uint32),
                ("objects", C.POINTER(RenderObject)), ("objectCount", C.c_uint32),
                ("bvhNodes", C.POINTER(BVHNode)), ("bvhNodeCount", C.c_uint32)]


class RtTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.POINTER(C.c_uint8))]


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("boxTests", "triTests", "raysTraced", "raysHit", "raysReference",
                                          "paths", "segments", "traceLaunches", "emitterTests", "skippedBoxTests")]


class RtHit(C.Structure):
    _fields_ = [("dst", C.c_float), ("didHit", C.c_uint32), ("isSphere", C.c_uint32),
                ("objectHitIndex", C.c_uint32), ("triHitIndex", C.c_uint32), ("materialIndex", C.c_uint32),
                ("frontFace", C.c_uint32), ("hitPoint", C.c_float * 3), ("normal", C.c_float * 3),
                ("boxTests", C.c_uint32), ("triTests", C.c_uint32)]


# every symbol include/rt_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
SYMBOLS = {
    "rt_scene_create": (C.c_int, [_P(_vp)]),
    "rt_scene_destroy": (None, [_vp]),
    "rt_scene_last_error": (C.c_char_p, [_vp]),
    "rt_scene_add_material": (C.c_int, [_vp, _P(RayMaterial)]),
    "rt_material_default": (None, [_P(RayMaterial)]),
    "rt_scene_set_sphere": (C.c_int, [_vp, C.c_uint32, _P(C.c_float), C.c_float, C.c_uint32]),
    "rt_placement_default": (None, [_P(RtPlacement)]),
    "rt_scene_read_obj": (C.c_int, [_vp, C.c_char_p, _P(RtPlacement), C.c_int]),
    "rt_scene_read_mtl": (C.c_int, [_vp, C.c_char_p]),
    "rt_scene_add_mesh": (C.c_int, [_vp, C.c_char_p, _P(C.c_float), _P(C.c_float), _P(C.c_float), C.c_uint32,
                                    _P(RtPlacement), C.c_int]),
    "rt_scene_cornell_box": (C.c_int, [_vp, C.c_char_p]),
    "rt_scene_prepare_default": (C.c_int, [_vp, C.c_char_p]),
    "rt_scene_get_arrays": (C.c_int, [_vp, _P(RtSceneArrays)]),
    "rt_scene_texture_count": (C.c_uint32, [_vp]),
    "rt_scene_texture_path": (C.c_char_p, [_vp, C.c_uint32]),
    "rt_scene_add_texture": (C.c_int, [_vp, C.c_char_p]),
    "rt_scene_set_material": (C.c_int, [_vp, C.c_uint32, _P(RayMaterial)]),
    "rt_scene_find_material": (C.c_int, [_vp, C.c_char_p]),
    "rt_scene_last_bvh_stats": (C.c_int, [_vp, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
    "rt_scene_set_bvh_hook": (C.c_int, [_vp, _vp, _vp]),
    "rt_camera_rotation": (None, [_P(C.c_float), _P(C.c_float)]),
    "rt_push_constants_default": (None, [_P(PushConstants), C.c_uint32, C.c_uint32]),
    "rt_transform_matrix": (None, [_P(RtPlacement), _P(C.c_float)]),
    "rt_device_count": (C.c_int, [_P(C.c_int)]),
    "rt_create": (C.c_int, [C.c_int, _P(_vp)]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_set_stream": (C.c_int, [_vp, _vp]),
    "rt_upload_scene": (C.c_int, [_vp, _P(RtSceneArrays)]),
    "rt_upload_textures": (C.c_int, [_vp, _P(RtTexture), C.c_uint32]),
    "rt_update_materials": (C.c_int, [_vp, _P(RayMaterial), C.c_uint32]),
    "rt_update_spheres": (C.c_int, [_vp, _P(Sphere), C.c_uint32]),
    "rt_update_objects": (C.c_int, [_vp, _P(RenderObject), C.c_uint32]),
    "rt_render": (C.c_int, [_vp, _P(PushConstants), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "rt_render_frames": (C.c_int, [_vp, _P(PushConstants), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "rt_sync": (C.c_int, [_vp]),
    "rt_clear_framebuffer": (C.c_int, [_vp]),
    "rt_read_rgba_f32": (C.c_int, [_vp, _P(C.c_float), C.c_size_t]),
    "rt_read_rgba8_srgb": (C.c_int, [_vp, _P(C.c_uint8), C.c_size_t]),
    "rt_trace_rays": (C.c_int, [_vp, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(RtHit)]),
    "rt_get_counters": (C.c_int, [_vp, _P(RtCounters)]),
    "rt_reset_counters": (C.c_int, [_vp]),
    "rt_set_profiling": (C.c_int, [_vp, C.c_int]),
    "rt_get_trace_time_ms": (C.c_int, [_vp, _P(C.c_double), _P(C.c_uint64)]),
    "rt_get_trace_busy_ms": (C.c_int, [_vp, _P(C.c_double)]),
    "rt_last_kernel": (C.c_char_p, [_vp]),
    "rt_last_parts": (C.c_int, [_vp]),
    "rt_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "rt_last_pipeline": (C.c_int, [_vp]),
    "rt_ray_cost": (C.c_double, [_vp]),
    "rt_bvh_build": (C.c_int, [_vp, _P(TrianglePoint), C.c_uint32, _P(Triangle), _P(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, _P(BVHNode), C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "rt_bvh_hook": (C.c_int, [_vp, _P(TrianglePoint), C.c_uint32, _P(Triangle), _P(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, _P(BVHNode), C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "rt_bvh_last_build_ms": (C.c_double, [_vp]),
    "rt_comm_unique_id": (C.c_int, [_vp]),
    "rt_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rt_comm_destroy": (C.c_int, [_vp]),
    "rt_gather_strips": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_int, _vp]),
    "rt_deinterleave_strips": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_int, _vp]),
    "rt_deinterleave_strips_host": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_int, _vp]),
    "rt_device_selftest": (C.c_int, [_vp, _P(C.c_uint32)]),
    "rt_host_selftest": (C.c_uint32, []),
    "rt_device_math_probe": (C.c_int, [_vp, C.c_uint32, _P(C.c_float), _P(C.c_float)]),
    "rt_measure_copy_bandwidth": (C.c_int, [_vp, C.c_size_t, C.c_int, _P(C.c_double)]),
    "rt_version": (C.c_char_p, []),
}

_lib = None


def lib():
    """Load librt_amd.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU or PyTorch fallback for the HIP path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
