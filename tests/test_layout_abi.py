"""Struct layouts and the C ABI surface: every function that
include/rt_amd.h declares is exported by librt_amd.so and bound in _capi."""
import ctypes as C
import os
import re
import subprocess

from ray_tracer_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_sizes_match_reference_std140():
    # sizes verified against the reference's host structs (src/vk_engine.h:49-189)
    assert C.sizeof(_capi.Sphere) == 32 and _capi.Sphere.radius.offset == 12 and _capi.Sphere.materialIndex.offset == 16
    assert C.sizeof(_capi.Triangle) == 48 and _capi.Triangle.binormal.offset == 16 and _capi.Triangle.tangent.offset == 32
    assert C.sizeof(_capi.TrianglePoint) == 32
    m = _capi.RayMaterial
    assert C.sizeof(m) == 64
    assert (m.emissionColor.offset, m.emissionStrength.offset, m.reflectance.offset, m.ior.offset) == (16, 28, 32, 36)
    assert (m.albedoIndex.offset, m.metalnessIndex.offset, m.alphaIndex.offset, m.bumpIndex.offset) == (40, 44, 48, 52)
    o = _capi.RenderObject
    assert C.sizeof(o) == 80 and (o.smoothShade.offset, o.bvhIndex.offset, o.materialIndex.offset, o.samplerIndex.offset) == (64, 68, 72, 76)
    assert C.sizeof(_capi.BVHNode) == 32 and _capi.BVHNode.index.offset == 24 and _capi.BVHNode.triCount.offset == 28
    cam = _capi.CameraInfo
    assert C.sizeof(cam) == 96 and (cam.pos.offset, cam.nearPlane.offset, cam.aspectRatio.offset, cam.fov.offset) == (64, 76, 80, 84)
    assert C.sizeof(_capi.EnvironmentData) == 64 and _capi.EnvironmentData.lightDir.offset == 48
    assert C.sizeof(_capi.RayTracerData) == 40
    pc = _capi.PushConstants
    assert C.sizeof(pc) == 208 and (pc.environment.offset, pc.rayTraceParams.offset, pc.frameCount.offset) == (96, 160, 200)


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_functions()
    assert len(declared) >= 35
    assert sorted(_capi.SYMBOLS) == declared, set(declared) ^ set(_capi.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(rt_[a-z0-9_]+)", out))
    missing = [n for n in declared if n not in exported]
    assert not missing, missing
    lib = _capi.lib()  # loads without a GPU and resolves everything
    for n in declared:
        assert getattr(lib, n)


def test_library_loads_without_gpu_and_fails_loudly():
    """No compute without a GPU: rt_create must fail (nonzero), never fall back."""
    lib = _capi.lib()
    assert lib.rt_host_selftest() == 0x0F
    assert lib.rt_version().startswith(b"ray_tracer_amd")
    n = C.c_int(-1)
    lib.rt_device_count(C.byref(n))
    if n.value <= 0:
        h = C.c_void_p()
        assert lib.rt_create(0, C.byref(h)) != 0
        assert not h.value
        from ray_tracer_amd import engine
        import pytest
        with pytest.raises(engine.RtError):
            engine.Renderer(0)


def test_gfx950_code_object_is_embedded():
    """The shipped library carries a gfx950 device image with the four pipeline kernels."""
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for k in (b"k_raygen", b"k_trace", b"k_shade", b"k_resolve"):
        assert k in blob
