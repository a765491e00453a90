"""bench.py prints exactly one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "2",
                        "--width", "480", "--height", "270", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # the kernel is bound by the vector-memory pipeline's gather rate, not by HBM: `frac` is that rate against the measured gather
    # peak and cannot exceed 1; SURVEY 8(d)'s algorithmic rate (which can) and the HBM counters ride beside it under their own names
    assert "vector-memory" in r["bound"] and r["unit"] == "GB/s" and r["peak"] > 8000.0 and r["achieved"] > 0
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_gbps"] > 0 and abs(r["algorithmic_frac"] - r["algorithmic_gbps"] / 8000.0) < 1e-9 and "traffic" in r
    assert d["per_step_dispatch"]["ms_per_step"] > 0 and d["unique_mrays_per_s"] > 0
    p = d["parity_check"]
    assert p["equal"] is True and p["max_rel"] == 0.0 and p["rows"] >= 4
    assert d["in_flight_check"]["equal"] is True and d["in_flight_check"]["frames"] == 3 + 1 + d["per_step_dispatch"]["steps"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


@pytest.mark.gpu
def test_one_rank_job_through_rccl_stitches_the_single_process_frame():
    """The N > 1 code path of bench.py (RCCL process group, strip gather overlapped with the next render, the
    reductions of the counters, the closing barrier) on the one GPU of the box: a one-rank job forced through the
    collectives must give the frame a plain single-process render gives (--check)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--spp", "2",
                        "--scene", "cornell", "--width", "320", "--height", "181", "--cpu-seconds", "0",
                        "--force-collective", "--check"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["check_tiled_equals_single"] is True
    assert d["n_gpus"] == 1 and d["value"] > 0


@pytest.mark.gpu
def test_four_rank_rehearsal_on_one_gpu_stitches_the_single_process_frame():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), rehearsed on the one GPU
    of the box: four ranks over gloo (RCCL cannot put two ranks on one device; the box admits at most six processes with the
    GPU open, and the test runner is one of them), each renders rows r, r+4, ... of every step, up to four steps in flight per rank
    (rt_render_frames: groups of 4 + 3), one gather per group. 181 rows do not divide by four and seven steps do not divide by the group size. --check: rank 0 re-renders all
    steps alone, one dispatch per step, and the stitched frame must equal it bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "1",
                        "--spp", "2", "--scene", "bunny", "--width", "320", "--height", "181", "--cpu-seconds", "0",
                        "--backend", "gloo", "--frames-in-flight", "4", "--check"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["check_tiled_equals_single"] is True
    assert d["n_gpus"] == 4 and d["steps"] == 7 and d["config"]["frames_in_flight"] == 4 and d["value"] > 0


def test_scripts_parse():
    """bench.py, __graft_entry__.py and the tools compile (CPU): a syntax slip in one of them must not wait for the GPU box to show."""
    import glob
    import py_compile
    for f in [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + glob.glob(os.path.join(ROOT, "tools", "*.py")):
        py_compile.compile(f, doraise=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode == 0 and "--per-step-dispatches" in p.stdout
