#!/usr/bin/env python3
"""Reduces the PMC passes of tools/profile_round.sh to the per-launch figures bench.py quotes in `roofline`.

usage: pmc_roofline.py <dir with pass_*/ subdirs and bench_pass.json> <kernel substring> <out.json> [note]

Every pass is its own `rocprofv3 --pmc ... --kernel-trace` run of the same bench command (MI355X_MICROARCH.md: separate
passes; FETCH_SIZE and WRITE_SIZE do not fit one pass). Counters are summed over the dispatches of the kernel and divided
by their number, the kernel's duration comes from the kernel trace of the same pass.

  traffic            = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes; FETCH_SIZE reports half the bytes on gfx950, guide section HBM;
                       uncalibrated for 16-byte gathers, Infinity-Cache hits included)
  cycles             = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs by rocprofv3) -> effective clock = cycles / duration
  ta_busy_frac       = TA_TA_BUSY_sum / (256 CUs * cycles)
  valu_issue_frac    = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * cycles)   (CDNA4's SIMDs are 32 lanes wide: a wave64 VALU instruction issues
                       over 2 cycles, MI355X_MICROARCH.md "Wave scheduling"; 4 only for a wave alone on its SIMD. Round 2 charged 4 and
                       reported twice the true figure. Against the 0.38 instructions per clock and SIMD that tools/micro/pk_rate.hip
                       reaches with a scalar-fp32 stream: valu_issue_frac_of_measured)
  active_lane_frac   = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) (lanes switched on per VALU instruction)
  l1_lookups         = TCP_TOTAL_CACHE_ACCESSES_sum (tag look-ups of the vector L1)
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["ray_tracer_amd/csrc/rt_kernels.hip.h", "ray_tracer_amd/csrc/rt_device.hip", "include/rt_det_math.h"]


def source_sha():
    h = hashlib.sha256()
    for f in SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    src, kernel, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    counters, launches, dur_ns = {}, {}, []
    for d in sorted(glob.glob(os.path.join(src, "pass_*"))):
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not cc:
            continue
        df = pd.read_csv(cc[0])
        df = df[df.Kernel_Name.str.contains(kernel, regex=False)]
        # every launch of the kernel counts (the multi-kernel pipeline's rounds shrink as paths end: the per-launch figures are
        # averages over all of them, like bench.py's own launch time); only launches below 1 % of the longest are left out
        # (the ray-cost probe's few rows, had it not been switched off)
        if kt:
            t = pd.read_csv(kt[0])
            t = t[t.Kernel_Name.str.contains(kernel, regex=False)]
            t["dur"] = t.End_Timestamp - t.Start_Timestamp
            big = t[t.dur > 0.01 * t.dur.max()]
            dur_ns.append(float(big.dur.mean()))
            keep = set(big.Dispatch_Id) if "Dispatch_Id" in big else None
            if keep is not None and "Dispatch_Id" in df:
                df = df[df.Dispatch_Id.isin(keep)]
        for name, g in df.groupby("Counter_Name"):
            n = g.Dispatch_Id.nunique() if "Dispatch_Id" in g else len(g)
            counters[name] = float(g.Counter_Value.sum()) / max(n, 1)
            launches[name] = int(n)
    bench = {}
    bp = os.path.join(src, "bench_pass.json")
    if os.path.exists(bp):
        with open(bp) as f:
            bench = json.load(f)
    ms = sum(dur_ns) / max(len(dur_ns), 1) / 1e6
    res = {"kernel": kernel, "note": note, "counters_per_launch": counters, "launches_per_pass": launches,
           "kernel_ms_under_profiler": ms, "source_sha": source_sha(),
           "git_head": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()}
    c = counters
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res["fetch_bytes_per_launch_raw"] = c["FETCH_SIZE"] * 1024.0
        res["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
        res["traffic_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
    if "GRBM_GUI_ACTIVE" in c and ms > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        res["cycles_per_launch"] = cyc
        res["effective_clock_ghz"] = cyc / (ms * 1e6)
        if "TA_TA_BUSY_sum" in c:
            res["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (256.0 * cyc)
        if "SQ_INSTS_VALU" in c:
            res["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
            res["valu_issue_frac_of_measured"] = c["SQ_INSTS_VALU"] / (1024.0 * cyc) / 0.38
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        res["active_lane_frac"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAIT_ANY") is not None and c["SQ_WAVE_CYCLES"]:
        res["wave_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        res["l1_lookups_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if bench:
        steps = max(bench.get("steps", 1), 1)
        launches = max(bench.get("roofline", {}).get("launches") or steps, 1)   # a launch may render several steps (frames in flight)
        res["steps_per_launch"] = steps / launches
        res["rays_traced_per_launch"] = bench["unique_mrays_per_s"] * 1e6 * bench["ms_per_step"] * 1e-3 * steps / launches
        res["box_tests_per_ray"] = bench.get("box_tests_per_ray")
        if "l1_lookups_per_launch" in res:
            res["l1_lookups_per_ray"] = res["l1_lookups_per_launch"] / res["rays_traced_per_launch"]
        res["bench_command"] = bench.get("command", "")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
