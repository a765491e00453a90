"""Framebuffer tiling across the GPUs of one node.

Each pixel depends only on its global index and the read-only scene
(raytrace.comp:563-564), so the frame shards with no mid-frame exchange:
rank r renders image rows r, r+N, r+2N, ... (fine interleave = even load),
seeds come from global pixel coordinates (rt_render's row0/rowStride), and one
gather of the fp32 strips to rank 0 ends the frame. Over RCCL every peer's
strip (3.1 MB at 1080p/8) goes over its own xGMI link; no ring is needed.
"""
import os

import torch
import torch.distributed as dist


def prepare_rccl_env():
    """Call before init_process_group("nccl"): this pool's host driver only supports dmabuf IPC, and without
    HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's peer buffers fail with `hipIpcGetMemHandle: invalid argument`.
    (Already exported on the GPU boxes; set here for any other launcher.)"""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the renderer's three streams next to torch's and RCCL's: more hardware queues than ROCm's default four, so that no two of them
    # share one (streams on one queue serialise; only effective when set before the process's first HIP call, as bench.py does)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def rows_of_rank(height, rank, world):
    """Image rows rendered by `rank`: rank, rank+world, ..."""
    return range(rank, height, world)


def max_rows(height, world):
    return (height + world - 1) // world


def gather_frame(strip, frame, height, world, rank, dst=0, force_collective=False):
    """Gather per-rank strips [(rows_r), W, 4] into frame [H, W, 4] on `dst`.

    `strip` holds this rank's rows in order; strips are padded to a common
    row count for the collective. Works with nccl (= RCCL) device tensors
    and with gloo CPU tensors. `force_collective` sends a one-rank job through
    the collective as well (the RCCL smoke test on a one-GPU box).
    """
    if world == 1 and not force_collective:
        if frame is not None:
            frame[:height].copy_(strip[:height])
        return frame
    mr = max_rows(height, world)
    if strip.shape[0] == mr:
        send = strip.contiguous()
    else:
        send = torch.zeros((mr,) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
        send[: strip.shape[0]].copy_(strip)
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=parts, dst=dst)
        for r in range(world):
            n = len(rows_of_rank(height, r, world))
            frame[r::world].copy_(parts[r][:n])
        return frame
    dist.gather(send, gather_list=None, dst=dst)
    return None
