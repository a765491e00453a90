#!/usr/bin/env python3
"""bench.py — Mrays/s of the wavefront path tracer on the BASELINE.json workload.

A "step" is one dispatch of the hot path (rt_render = the reference's run_compute + raytrace.comp) over the whole 1920x1080
Sponza frame at --spp samples per pixel, with frameCount advancing per step as in the reference's progressive mode
(src/vk_engine.cpp:1812-1814). With N GPUs the framebuffer is tiled by interleaved rows (rank r renders rows r, r+N, ...),
the scene is replicated, there is no mid-frame traffic, and the fp32 strips are gathered to rank 0 over RCCL (total work fixed
=> "strong" scaling). Steps are independent until they are blended, so a rank submits --frames-in-flight of them at once
(rt_render_frames: their pixels share a launch, the blends follow in frame order — the same bits as one dispatch per step).

value    = reference-semantics rays / s: 1 ray = 1 calculateIntersections call of shaders/raytrace.comp (4 per diffuse
           segment). `unique_mrays_per_s` beside it counts only the scene queries the GPU actually executed (the
           duplicate NEE query merged, light queries answered from the emitter list not traced, the camera ray traced once per
           pixel and dispatch) — the roofline uses only executed work.
roofline : the dominant kernel (k_trace_pw) is bound by no bandwidth and no ALU: the counters of the HEAD binary (separate rocprofv3
           --pmc passes, profiles/counters_*.json, stamped with a hash of the kernel sources) show its busiest unit to be the
           vector-memory pipeline that serves its 16-byte gathers (texture addresser / L1 tag look-ups), with waves parked on
           s_waitcnt more than half of their cycles. So: bound = that pipeline; achieved = L1 look-ups per second (look-ups per
           executed ray from the counters x rays per second of kernel time, timed live with HIP events) x 16 B; peak = the gather
           rate tools/gather_bench.hip measures on this chip (1.39 look-ups per clock and CU); frac = achieved / peak <= 1.
           Beside it: algorithmic_gbps / algorithmic_frac (32 B per box test + 36 B per triangle test + 100 B per hit,
           counted by the kernel, / launch time, against 8 TB/s — a cache-served rate that may exceed 1), traffic and
           hbm_counter_frac (bytes that reach the fabric, PMC), ta_busy_frac, valu_issue_frac, wave_waiting_frac, active_lane_frac.
cpu_baseline: the scalar oracle (oracle/, a port of the shader) timed on this host's cores on a bounded sample of the same frame;
parity_check: those very rows rendered by the GPU and compared bit for bit (the oracle is the checker, never the product).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # see ray_tracer_amd.tiling.prepare_rccl_env (set before torch loads RCCL)
# A dispatch runs in three parts on three HIP streams (rt_amd.h, "lanes"); ROCm maps a process's streams onto GPU_MAX_HW_QUEUES
# hardware queues (4 by default) and two streams on one queue run one after the other: with torch's and RCCL's own streams in the
# process the parts could end up sharing (measured with a fourth part: 103.7 ms per step on 4 queues, 82.2 on 8). Read at HIP's start-up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


GATHER_PEAK_LOOKUPS_PER_CLK_CU = 1.39   # tools/gather_bench.hip (profiles/README.md, "what a vector load costs")

SCENE_NAMES = {"cornell": "Cornell", "bunny": "Cornell + bunny", "dragon": "Cornell + dragon", "sponza": "Sponza",
               "sponza_dragons": "Sponza + 16 dragon instances", "sponza_dragons_flat": "Sponza + 16 dragons (flattened)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="sponza", choices=["cornell", "bunny", "dragon", "sponza", "sponza_dragons", "sponza_dragons_flat"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket traversal launches with HIP events")
    ap.add_argument("--no-build", action="store_true", help="do not run __graft_entry__.build() first (under rocprofv3: build beforehand, so that no compiler or make process is started from the profiled one)")
    ap.add_argument("--per-step-dispatches", type=int, default=5, help="after the timed region: this many steps dispatched one by one (one dispatch and, on N GPUs, one gather per step, as the reference's draw() loop presents every frame), reported as `per_step_dispatch`; 0 = skip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default); gloo only to rehearse N ranks on ONE GPU (strips gathered through host memory)")
    ap.add_argument("--tune", default="", help="comma-separated rt_set_tuning knobs, e.g. blocks_per_cu=2,refill=8")
    ap.add_argument("--force-collective", action="store_true",
                    help="send a one-rank job through the process group, gather and reductions too (RCCL smoke test on a one-GPU box)")
    ap.add_argument("--no-in-flight-check", action="store_true", help="one GPU: skip re-rendering all steps one dispatch at a time for the comparison with the timed image")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the whole frame alone and compares it with the gathered one")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="most steps (= progressive frames) a rank submits at once through rt_render_frames; 0 = 10 N on N GPUs. A pixel's "
                         "samples are serial, so what fills a GPU is paths: N frames of a 1/N tile are one frame's worth, and the "
                         "traversal gets cheaper per ray the more paths share a dispatch (Sponza 1080p, ms per step with 1 / 2 / 4 / 10 / 20 "
                         "frames in one dispatch: 117 / 113.5 / 103.9 / 99.4 / 98.8). The steps of a run are split into equal groups of "
                         "at most that many. 1 = every step its own dispatch and its own gather")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not args.no_build:
        ge.build()  # mtime check under a file lock: one rank compiles if anything is stale, the others wait for it
    from ray_tracer_amd import engine, scenes, tiling

    rehearsal = args.backend == "gloo"
    if rehearsal:
        local = 0  # every rank shares GPU 0; RCCL cannot run two ranks on one device
    torch.cuda.set_device(local)
    multi = world > 1 or args.force_collective
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            tiling.prepare_rccl_env()
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    W, H = args.width, args.height
    scene, label = scenes.CONFIGS[args.scene]()
    cam = scenes.sponza_camera if args.scene.startswith("sponza") else engine.push_constants
    pc = cam(W, H, raysPerPixel=args.spp, progressive=1, singleRender=0)

    r = engine.Renderer(local)
    if r.selftest() != 0x0F:
        raise SystemExit("device deterministic-math self-test failed")
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        r.set_tuning(k, int(v))
    r.upload_scene(scene)
    rows = tiling.rows_of_rank(H, rank, world)
    strip = torch.zeros((len(rows), W, 4), dtype=torch.float32, device=f"cuda:{local}")
    fdev = "cpu" if rehearsal else f"cuda:{local}"
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=fdev) if rank == 0 else None

    # N > 1: the strip of step i is copied aside and gathered while step i+1 renders (the progressive blend keeps
    # its history in `strip`, so the renderer goes on in place); the gather's kernels fill the tail of the render.
    sendbuf = torch.zeros_like(strip) if multi else None
    pending = [False]

    # Steps are progressive frames: independent until they are blended in order. A rank may therefore submit a group of
    # them at once (rt_render_frames: their pixels share a launch, the blends follow in frame order — the same bits as
    # one dispatch per step); the strips are gathered once per group.
    fif = args.frames_in_flight if args.frames_in_flight > 0 else 10 * world

    def launch(i, n):
        pc.frameCount = i
        if n == 1:
            r.render(pc, W, H, row0=rank, rowStride=world, nRows=len(rows), out_ptr=strip.data_ptr(), sync=False)
        else:
            r.render_frames(pc, W, H, n, row0=rank, rowStride=world, nRows=len(rows), out_ptr=strip.data_ptr(), sync=False)

    def finish_previous():
        if pending[0]:
            tiling.gather_frame(sendbuf.cpu() if rehearsal else sendbuf, frame, H, world, rank, force_collective=multi)
            pending[0] = False

    def step(i, n=1):
        launch(i, n)
        if multi:
            finish_previous()      # gather of step i-1 overlaps the render of step i
        r.sync()
        if multi:
            sendbuf.copy_(strip)
            torch.cuda.current_stream().synchronize()
            pending[0] = True

    def fence():
        if multi:
            finish_previous()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def groups(count):
        """`count` steps in the fewest groups of at most `fif`, sized evenly (24 steps, at most 10 at once: 8 + 8 + 8, not
        10 + 10 + 4 — a small last group would leave the GPU short of paths, which is what the groups are there to avoid)."""
        if count <= 0:
            return []
        k = (count + fif - 1) // fif
        return [count // k + (1 if j < count % k else 0) for j in range(k)]

    def run(first, count):
        i = first
        for n in groups(count):
            step(i, n)
            i += n

    run(0, args.warmup)
    r.reset_counters()
    r.set_profiling(not args.no_profile)
    fence()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    trace_ms, trace_launches = r.trace_time_ms()
    busy_ms = r.trace_busy_ms() if not args.no_profile else 0.0
    r.set_profiling(False)
    cnt = r.counters()
    timed_pipeline = r.last_pipeline()   # of the timed region (the legs below may run in the other one)

    # The same steps, one dispatch (and one gather) per step: what a host that presents every frame gets.
    per_step = None
    frames_rendered = args.warmup + args.steps   # progressive frames the image holds by now
    if args.per_step_dispatches > 0:
        k2 = args.per_step_dispatches
        fif_saved, fif = fif, 1
        run(args.warmup + args.steps, 1)   # untimed: the first dispatch of another shape (allocations, pipeline choice)
        r.reset_counters()
        fence()
        t1 = time.perf_counter()
        run(args.warmup + args.steps + 1, k2)
        fence()
        dt1 = time.perf_counter() - t1
        c1 = r.counters()
        fif = fif_saved
        frames_rendered += 1 + k2
        v1 = torch.tensor([float(c1["raysReference"]), float(c1["raysTraced"]), dt1], dtype=torch.float64, device=fdev)
        if multi:
            m1 = v1.clone()
            dist.all_reduce(v1, op=dist.ReduceOp.SUM)
            dist.all_reduce(m1, op=dist.ReduceOp.MAX)
            dt1 = float(m1[2])
        per_step = {"steps": k2, "ms_per_step": dt1 / k2 * 1e3, "value": float(v1[0]) / dt1 / 1e6, "unique_mrays_per_s": float(v1[1]) / dt1 / 1e6,
                    "pipeline": ["multi-kernel", "fused"][r.last_pipeline()],
                    "what": "one dispatch" + (" and one RCCL gather" if world > 1 else "") + " per step (frames_in_flight 1), outside the timed region"}

    keys = ["boxTests", "triTests", "raysTraced", "raysHit", "raysReference", "paths", "segments", "skippedBoxTests"]
    vec = torch.tensor([float(cnt[k]) for k in keys] + [dt, trace_ms, float(trace_launches), busy_ms], dtype=torch.float64,  # (the indices below follow len(keys))
                       device=fdev)
    if multi:
        mx = vec.clone()
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx[len(keys)])
    tot = {k: float(vec[i]) for i, k in enumerate(keys)}
    sum_trace_ms, sum_launches, sum_busy_ms = float(vec[len(keys) + 1]), float(vec[len(keys) + 2]), float(vec[len(keys) + 3])

    if rank == 0:
        # executed tests only: what a ray is charged for objects it is taken past (two box tests each, never fetched) is left out
        alg_bytes = 32.0 * (tot["boxTests"] - tot["skippedBoxTests"]) + 36.0 * tot["triTests"] + 100.0 * tot["raysHit"]
        launches = max(sum_launches, 1.0)
        avg_launch_ms = sum_trace_ms / launches
        # per GPU: the kernel's algorithmic bytes per launch / its average launch duration
        alg_gbps = (alg_bytes / launches) / (avg_launch_ms * 1e-3) / 1e9 if sum_trace_ms > 0 else None
        tkern = ["k_trace_pw", "k_render_fused"][timed_pipeline]
        out = {
            "metric": f"Mrays/s (reference-semantics closest-hit queries) at {W}x{H} {SCENE_NAMES[args.scene]}",
            "value": tot["raysReference"] / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} ({label}), {W}x{H}, {args.spp} spp/step, bounceLimit 8, "
                                   f"rows interleaved over {world} GPU(s)" + f", up to {fif} steps in flight per rank (groups of {'+'.join(map(str, groups(args.steps)))})" + (", one RCCL gather per group" if world > 1 else ""),
                       "scene": args.scene, "assets": label, "width": W, "height": H, "spp_per_step": args.spp, "frames_in_flight": fif,
                       "pipeline": ["multi-kernel (k_trace_pw + k_shade per round)", "fused (k_render_fused)"][timed_pipeline]},
            "unique_mrays_per_s": tot["raysTraced"] / dt / 1e6,
            "value_counts": "calculateIntersections calls of the reference's shader (4 per diffuse segment); unique_mrays_per_s = scene queries the GPU executed",
            "spp_per_s": tot["paths"] / (W * H) / dt,
            "paths": tot["paths"], "segments": tot["segments"],
            "box_tests_per_ray": tot["boxTests"] / max(tot["raysTraced"], 1), "tri_tests_per_ray": tot["triTests"] / max(tot["raysTraced"], 1),
            "executed_box_tests_per_ray": (tot["boxTests"] - tot["skippedBoxTests"]) / max(tot["raysTraced"], 1),
        }
        if per_step:
            out["per_step_dispatch"] = per_step
        # The kernel's roofline. `bound` names the unit the counters show busiest; achieved / peak / frac are the rate of that unit.
        rf = {"kernel": tkern + ("" if tkern == "k_trace_pw" else " (traversal + shading in one kernel)"),
              "bound": "vector-memory pipeline (16-byte gathers through the texture addresser and the L1 tags); latency- and divergence-limited: neither HBM nor MFMA nor VALU is near its peak",
              "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None,
              "achieved_is": "L1 look-ups per second x 16 B: look-ups per executed ray (PMC counters of the same binary) x executed rays / time with a traversal launch running (HIP events, live)",
              "peak_is": "1.39 look-ups per clock and CU (tools/gather_bench.hip on this chip: a 64-lane dwordx4 gather costs the CU 46 clocks, L2-resident table) x 256 CUs x 2.4 GHz x 16 B",
              "launches": sum_launches, "avg_launch_ms": avg_launch_ms,
              "launch_overlap": sum_trace_ms / sum_busy_ms if sum_busy_ms > 0 else None,
              "trace_busy_share_of_step": (sum_busy_ms / world) / (dt * 1e3) if sum_busy_ms > 0 else None,
              "algorithmic_bytes_per_launch": alg_bytes / launches,
              "algorithmic_gbps": alg_gbps, "algorithmic_frac": (alg_gbps / 8000.0) if alg_gbps else None,
              "algorithmic_is": "SURVEY 8(d): 32 B per box test + 36 B per triangle test + 100 B per hit, counted by the kernel, per launch / average launch duration, against 8 TB/s of HBM. Mostly served by LDS / L1 / L2, so it can exceed 1: it says how much traversal work per second the kernel does, not how busy HBM is (hbm_counter_frac does). With the dispatch in parts on several streams the launches overlap (launch_overlap) and share the GPU, so a launch's duration is longer than the kernel would need alone"}
        out["roofline"] = rf
        # What the hardware counters say about the same kernel: separate rocprofv3 --pmc passes of this very command
        # (tools/profile_round.sh -> profiles/), since PMC counters cannot be read from inside the process. The file is
        # stamped with a hash of the kernel sources; a stamp that no longer matches the tree is flagged, not hidden.
        cfile = os.path.join(ROOT, "profiles", f"counters_{tkern}_{args.scene}_{W}x{H}_{args.spp}spp.json")
        if world == 1 and os.path.exists(cfile):
            with open(cfile) as f:
                pm = json.load(f)
            rf["counters_source"] = os.path.relpath(cfile, ROOT)
            rf["counters_stale"] = pm.get("source_sha") != kernel_source_sha()
            rf["traffic"] = pm.get("traffic_bytes_per_launch")
            if rf["traffic"] and pm.get("kernel_ms_under_profiler"):
                # both from the profiled run (its own launch count and durations), per launch like everything else here
                rf["hbm_counter_gbps"] = rf["traffic"] / (pm["kernel_ms_under_profiler"] * 1e-3) / 1e9
                rf["hbm_counter_frac"] = rf["hbm_counter_gbps"] / 8000.0
            for k in ("ta_busy_frac", "valu_issue_frac", "active_lane_frac", "wave_waiting_frac", "l1_lookups_per_ray", "effective_clock_ghz"):
                if k in pm:
                    rf[k] = pm[k]
            if pm.get("l1_lookups_per_ray") and sum_busy_ms > 0:
                lookups_per_s = pm["l1_lookups_per_ray"] * tot["raysTraced"] / (sum_busy_ms * 1e-3) / world
                rf["achieved"] = lookups_per_s * 16.0 / 1e9
                rf["achieved_from"] = "PMC look-ups per ray x live ray rate"
        rf["peak"] = GATHER_PEAK_LOOKUPS_PER_CLK_CU * 256 * 2.4e9 * 16.0 / 1e9
        if rf["achieved"] is None and sum_busy_ms > 0:
            # no counters of this workload in profiles/ (another scene, size or GPU count): the look-ups the traversal's own work
            # counters imply when none is served from LDS — four 16-byte loads per pair of child boxes, three per triangle, five per
            # ray (its record in, its hit out): an upper bound of the traversal's share, per GPU
            est = 2.0 * (tot["boxTests"] - tot["skippedBoxTests"]) + 3.0 * tot["triTests"] + 5.0 * tot["raysTraced"]
            rf["achieved"] = est / (sum_busy_ms * 1e-3) / world * 16.0 / 1e9
            rf["achieved_from"] = "estimate from the kernel's work counters (2 look-ups per box test, 3 per triangle test, 5 per ray; upper bound: the top levels' pairs come from LDS)"
        if rf["achieved"] is not None:
            rf["frac"] = rf["achieved"] / rf["peak"]
        if world == 1:
            # a streaming copy measured on this very box (1 GiB, read + write), next to the nominal 8 TB/s
            rf["measured_copy_gbps"] = r.copy_bandwidth_gbps(1 << 30, 5)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], ref_rows, tile = cpu_baseline(scene, pc, W, H, args)
            # the rows the oracle has just rendered are the rows of the bench frame at frameCount 0: render them on the
            # GPU with the same constants and compare, bit for bit (the oracle is the checker here, never the product)
            # ... through both pipelines (the timed groups run in one of them, small tiles in the other)
            same, max_rel = True, 0.0
            for pipe in (0, 1):
                r.set_tuning("pipeline", pipe)
                r.reset_counters()
                pc.frameCount = 0
                gpu_rows = r.render(pc, W, H, **tile)
                same = same and bool(np.array_equal(gpu_rows.view(np.uint32), ref_rows.view(np.uint32)))
                with np.errstate(all="ignore"):
                    rel = np.abs(gpu_rows - ref_rows) / np.maximum(np.abs(ref_rows), 1e-6)
                max_rel = max(max_rel, float(np.nanmax(rel)) if rel.size else 0.0)
            r.set_tuning("pipeline", -1)
            out["parity_check"] = {"rows": tile["nRows"], "pixels": int(tile["nRows"] * W), "spp": args.spp, "equal": same,
                                   "max_rel": max_rel, "pipelines": ["multi-kernel", "fused"],
                                   "against": "oracle (scalar restatement of raytrace.comp), same rows, frameCount 0"}
        if world == 1 and not multi and not args.no_in_flight_check:
            # the frame the timed groups left behind (warm-up + steps, several steps per dispatch) against the same steps
            # dispatched one by one: bit for bit, or the run fails
            r.reset_counters()
            solo = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")
            for i in range(frames_rendered):
                pc.frameCount = i
                r.render(pc, W, H, out_ptr=solo.data_ptr(), sync=True)
            same = bool(torch.equal(solo.view(torch.int32), strip.view(torch.int32)))
            out["in_flight_check"] = {"equal": same, "frames": frames_rendered,
                                      "what": "the progressive image after all steps, rendered in groups (steps in flight), equals the one rendered one dispatch per step"}
            del solo
        if args.check and multi:
            # the same frames rendered by one process must equal the stitched strips bit for bit
            r.reset_counters()
            solo = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")
            for i in range(frames_rendered):
                pc.frameCount = i
                r.render(pc, W, H, out_ptr=solo.data_ptr(), sync=True)
            same = bool(torch.equal(solo.cpu().view(torch.int32), frame.cpu().view(torch.int32)))
            out["check_tiled_equals_single"] = same
            if not same:
                raise SystemExit("tiled frame differs from the single-GPU frame")
        print(json.dumps(out), flush=True)
        if out.get("parity_check", {}).get("equal") is False:
            raise SystemExit("parity_check failed: the GPU rows differ from the oracle's")
        if out.get("in_flight_check", {}).get("equal") is False:
            raise SystemExit("in_flight_check failed: the image rendered with several steps in flight differs from the one rendered step by step")
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def kernel_source_sha():
    """Hash of the sources the traversal kernels are built from (the stamp tools/pmc_roofline.py puts into profiles/)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("ray_tracer_amd/csrc/rt_kernels.hip.h", "ray_tracer_amd/csrc/rt_device.hip", "include/rt_det_math.h"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, pc, W, H, args):
    """The scalar oracle on this host's cores, on every k-th row of the same frame."""
    from oracle import pyoracle
    threads = pyoracle.effective_cpus()
    # timed as the shader's own work only: without the oracle's second evaluation of every light query (its check of the
    # pipeline's shortcut, which the parity tests use); the pixels are the same either way
    pyoracle.lib().oracle_set_light_queries(0)
    pc.frameCount = 0
    # calibrate on 4 rows spread over the frame, then size the sample for ~cpu_seconds
    probe_rows = 4
    t = time.perf_counter()
    pyoracle.render(scene, pc, W, H, row0=H // 8, rowStride=H // probe_rows, nRows=probe_rows, threads=threads)
    per_row = max(time.perf_counter() - t, 1e-3) / probe_rows
    n_rows = int(max(probe_rows, min(H, args.cpu_seconds / per_row)))
    stride = max(1, H // n_rows)
    n_rows = min(n_rows, (H + stride - 1) // stride)
    t = time.perf_counter()
    rows, c = pyoracle.render(scene, pc, W, H, row0=0, rowStride=stride, nRows=n_rows, threads=threads)
    dt = time.perf_counter() - t
    return ({"value": c["raysReference"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{n_rows} of {H} rows (every {stride}th) of the same {W}x{H} frame at {args.spp} spp, "
                      f"{c['raysReference']} rays in {dt:.1f} s",
            "what": "scalar C++ restatement of raytrace.comp: every calculateIntersections call of the shader is executed (4 per diffuse segment)"},
            rows, dict(row0=0, rowStride=stride, nRows=n_rows))


if __name__ == "__main__":
    main()
