"""TEST INFRASTRUCTURE — ctypes loader for the scalar CPU oracle
(oracle/raytrace_oracle.cpp, the restatement of shaders/raytrace.comp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module. The product (ray_tracer_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ray_tracer_amd._capi import PushConstants, RtHit, RtSceneArrays, RtTexture

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "boxTestsReference", "triTestsReference", "raysReference", "raysHitReference",
        "boxTests", "triTests", "raysTraced", "raysHit", "paths", "segments", "stackOverflow", "emitterTests",
        "lightQueryMismatch")]


def effective_cpus():
    """CPUs this process may actually use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.POINTER(RtSceneArrays), C.POINTER(PushConstants), C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                    C.POINTER(OracleCounters), C.c_int]
        l.oracle_trace_rays.restype = C.c_int
        l.oracle_trace_rays.argtypes = [C.POINTER(RtSceneArrays), C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(RtHit)]
        l.oracle_set_light_queries.restype = None
        l.oracle_set_light_queries.argtypes = [C.c_int]
        l.oracle_set_camera_reuse.restype = None
        l.oracle_set_camera_reuse.argtypes = [C.c_int]
        l.oracle_set_textures.restype = None
        l.oracle_set_textures.argtypes = [C.POINTER(RtTexture), C.c_uint32]
        l.oracle_random.restype = C.c_float
        l.oracle_random.argtypes = [C.POINTER(C.c_uint32)]
        l.oracle_math_probe.restype = None
        l.oracle_math_probe.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        l.oracle_mat4_inverse.restype = None
        l.oracle_mat4_inverse.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.oracle_glsl_probe.restype = None
        l.oracle_glsl_probe.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.oracle_selftest.restype = C.c_uint32
        l.oracle_hardware_threads.restype = C.c_uint
        _lib = l
    return _lib


def render(scene, pc, width, height, row0=0, rowStride=1, nRows=None, threads=None, prev=None):
    """One dispatch of the megakernel on the CPU. Returns (rgba[nRows,width,4], counters dict)."""
    if nRows is None:
        nRows = (height - row0 + rowStride - 1) // rowStride
    a = scene.arrays()
    pc.rayTraceParams.sphereCount = a.sphereCount
    pc.rayTraceParams.objectCount = a.objectCount
    out = np.zeros((nRows, width, 4), dtype=np.float32) if prev is None else np.ascontiguousarray(prev, np.float32).copy()
    cnt = OracleCounters()
    if threads is None:
        threads = effective_cpus()
    rc = lib().oracle_render(C.byref(a), C.byref(pc), width, height, row0, rowStride, nRows,
                             out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cnt), int(threads))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return out, {n: getattr(cnt, n) for n, _ in OracleCounters._fields_}


def trace_rays(scene, origins, dirs):
    a = scene.arrays()
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    hits = (RtHit * o.shape[0])()
    fp = C.POINTER(C.c_float)
    rc = lib().oracle_trace_rays(C.byref(a), a.sphereCount, a.objectCount, o.shape[0], o.ctypes.data_as(fp),
                                 d.ctypes.data_as(fp), hits)
    if rc != 0:
        raise RuntimeError(f"oracle_trace_rays failed: {rc}")
    return hits


def random(state):
    s = C.c_uint32(state)
    r = lib().oracle_random(C.byref(s))
    return s.value, r


def math_probe(x, y):
    out = (C.c_float * 8)()
    lib().oracle_math_probe(float(x), float(y), out)
    return np.array(list(out), dtype=np.float32)


def glsl_probe(inputs):
    """include/rt_probe.h through the oracle's build: inputs [n, 32] float32 -> [n, 64] float32."""
    x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 32)
    out = np.zeros((x.shape[0], 64), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().oracle_glsl_probe(x.shape[0], x.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def set_textures(images):
    """The scene's texture table for the next render() calls: a list of uint8 arrays [h, w, 4] (R8G8B8A8_SRGB), [] = none."""
    arr = (RtTexture * max(len(images), 1))()
    keep = []
    for i, im in enumerate(images):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        assert im.ndim == 3 and im.shape[2] == 4
        keep.append(im)
        arr[i].width, arr[i].height = im.shape[1], im.shape[0]
        arr[i].rgba8 = im.ctypes.data_as(C.POINTER(C.c_uint8))
    lib().oracle_set_textures(arr, len(images))
