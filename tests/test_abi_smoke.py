"""The C ABI from compiled hosts.

* include/rt_amd.h is a plain C header (gcc -std=c99) as well as a C++ one;
* its structs are the reference's structs: src/vk_engine.h's own definitions (read from /root/reference at test time, where
  that exists — it does not travel to the GPU box) compiled with the vendored glm next to rt_amd.h, every size and every
  field offset compared by the compiler;
* tests/abi_smoke.cpp, a C++ program that follows INTEGRATION.md's call sequence (no Python, no torch), gives the frames
  the ctypes path gives, bit for bit, and runs the multi-GPU calls (rt_comm_* / rt_gather_strips over RCCL) on one rank."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

STRUCTS = {   # struct -> fields whose offsets are compared
    "Sphere": ["position", "radius", "materialIndex"],
    "Triangle": ["v0", "v1", "v2", "frontOnly", "binormal", "tangent"],
    "TrianglePoint": ["position", "normal"],
    "RayMaterial": ["albedo", "emissionColor", "emissionStrength", "reflectance", "ior", "albedoIndex", "metalnessIndex", "alphaIndex", "bumpIndex"],
    "RenderObject": ["transformMatrix", "smoothShade", "bvhIndex", "materialIndex", "samplerIndex"],
    "BVHNode": ["boundsX", "boundsY", "boundsZ", "index", "triCount"],
    "CameraInfo": ["cameraRotation", "pos", "nearPlane", "aspectRatio", "fov"],
    "EnvironmentData": ["horizonColor", "zenithColor", "groundColor", "lightDir"],
    "RayTracerData": ["progressive", "singleRender", "debug", "raysPerPixel", "bounceLimit", "sphereCount", "objectCount", "triangleCap", "boxCap", "sampleLimit"],
    "PushConstants": ["camInfo", "environment", "rayTraceParams", "frameCount"],
}


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "c99.c"
    src.write_text('#include "rt_amd.h"\nint main(void) { PushConstants pc; rt_push_constants_default(&pc, 8, 8); return (int)sizeof(RtCounters) == 0; }\n')
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "vk_engine.h")), reason="the reference tree is not on this machine")
def test_structs_equal_the_references_own_header(tmp_path):
    """The reference's struct definitions, taken from its header as it lies there, against include/rt_amd.h."""
    text = open(os.path.join(REF, "src", "vk_engine.h"), encoding="utf-8-sig").read()
    bodies = []
    for name in STRUCTS:
        m = re.search(r"^struct %s \{.*?^\};" % name, text, flags=re.S | re.M)
        assert m, f"struct {name} not found in the reference's header"
        bodies.append(m.group(0))
    checks = []
    for name, fields in STRUCTS.items():
        checks.append(f'static_assert(sizeof(ref::{name}) == sizeof(::{name}), "sizeof {name}");')
        for f in fields:
            checks.append(f'static_assert(offsetof(ref::{name}, {f}) == offsetof(::{name}, {f}), "{name}.{f}");')
    src = tmp_path / "layout.cpp"
    src.write_text("#include <sys/types.h>\n#include <cstddef>\n#include <glm/glm.hpp>\n#include \"rt_amd.h\"\nnamespace ref {\n"
                   + "\n".join(bodies) + "\n}\n" + "\n".join(checks) + "\nint main() { return 0; }\n")
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wno-invalid-offsetof", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(REF, "third_party", "glm"), str(src)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.gpu
def test_cpp_host_follows_integration_md(tmp_path, renderer):
    exe, out = tmp_path / "abi_smoke", tmp_path / "frames.bin"
    libdir = os.path.join(ROOT, "ray_tracer_amd")
    p = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                        os.path.join(ROOT, "tests", "abi_smoke.cpp"), "-L", libdir, "-lrt_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    from ray_tracer_amd import engine
    p = subprocess.run([str(exe), engine.ASSET_DIR, str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "abi_smoke ok" in p.stdout, (p.stdout + p.stderr)[-3000:]
    W, H = 96, 57
    frames = np.fromfile(out, dtype=np.float32).reshape(2, H, W, 4)
    # the same two frames through ctypes
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.set_sphere(0, (0.0, 0.1, -0.3), 0.4, 5)
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=3)
    renderer.set_tuning("pipeline", -1)
    renderer.upload_scene(s)
    a = renderer.render(pc, W, H)
    s.set_sphere(0, (0.2, 0.0, -0.2), 0.35, 4)
    renderer.update_spheres(s)
    b = renderer.render(pc, W, H)
    assert np.array_equal(frames[0].view(np.uint32), a.view(np.uint32))
    assert np.array_equal(frames[1].view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,height", [(1, 5), (2, 7), (3, 10), (3, 2), (8, 1080), (8, 1083), (5, 3)])
def test_strips_of_n_ranks_become_the_frame(renderer, n_ranks, height):
    """rt_gather_strips' second half (k_deinterleave_rows) with more than one rank's strips, heights the rank count does not divide
    and fewer rows than ranks: rank r's strip holds rows r, r + N, ... and the strips lie rank after rank (the RCCL
    gather itself has only ever run with one rank, where this step is the identity)."""
    import numpy as np
    width = 37
    frame = np.arange(height * width * 4, dtype=np.float32).reshape(height, width, 4)
    strips = np.concatenate([frame[r::n_ranks] for r in range(n_ranks)], axis=0)
    assert strips.shape == frame.shape
    # (host buffers through the library's own copies: a second HIP runtime in this process — torch's, or libamdhip64 loaded by
    # hand — does not see the device the library's runtime holds)
    assert np.array_equal(renderer.deinterleave_strips_host(strips, n_ranks), frame)
