"""The host scene surface (OBJ / MTL loaders, BVH builder, argument checks) under AddressSanitizer + UBSan on the CPU
build: shipped assets, a few hundred mutated OBJ / MTL files, argument errors (tools/sanitize_scene.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_scene_surface_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_scene")
    cc = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
          "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray_tracer_amd", "csrc"),
          os.path.join(ROOT, "ray_tracer_amd", "csrc", "scene.cpp"), os.path.join(ROOT, "tools", "sanitize_scene.cpp"), "-o", exe]
    b = subprocess.run(cc, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "asan" in b.stderr.lower():
        pytest.skip("libasan not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    p = subprocess.run([exe, os.path.join(ROOT, "assets"), "600"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "none out of bounds" in p.stdout and "argument errors reported: 7 of 7" in p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_oracle_under_asan_ubsan(tmp_path):
    """The checker itself: Cornell + spheres + klein bottle in every debug mode, environment on and off, and a ray batch
    with axis-parallel and zero directions (tests/sanitize_oracle.cpp)."""
    exe = str(tmp_path / "sanitize_oracle")
    cc = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
          "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray_tracer_amd", "csrc"),
          "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ray_tracer_amd", "csrc", "scene.cpp"),
          os.path.join(ROOT, "oracle", "raytrace_oracle.cpp"), os.path.join(ROOT, "tests", "sanitize_oracle.cpp"), "-o", exe]
    b = subprocess.run(cc, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "asan" in b.stderr.lower():
        pytest.skip("libasan not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    p = subprocess.run([exe, os.path.join(ROOT, "assets")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "oracle under sanitizers ok" in p.stdout
