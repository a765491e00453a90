"""N>1 path on CPU: two gloo ranks each produce their interleaved row strip
(with the CPU oracle standing in for the GPU renderer, tests may do that), the
product's tiling.gather_frame stitches them on rank 0, and the result equals
the full single-process frame bit for bit. Covers row assignment, padding of
unequal strips (odd heights) and the gather/de-interleave used by bench.py."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, outfile):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from ray_tracer_amd import engine, tiling
    from util import cornell_scene
    s = cornell_scene(True)
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=2)
    rows = tiling.rows_of_rank(H, rank, world)
    strip, _ = pyoracle.render(s, pc, W, H, row0=rank, rowStride=world, nRows=len(rows), threads=2)
    frame = torch.zeros((H, W, 4), dtype=torch.float32) if rank == 0 else None
    tiling.gather_frame(torch.from_numpy(strip), frame, H, world, rank)
    if rank == 0:
        np.save(outfile, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(world, W, H, tmp_path):
    out = str(tmp_path / f"frame_{world}.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, out), nprocs=world, join=True)
    return np.load(out)


def test_two_ranks_stitch_to_the_single_process_frame(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    from ray_tracer_amd import engine
    from util import cornell_scene
    W, H = 40, 27   # odd height: rank 0 has one row more than rank 1
    full, _ = pyoracle.render(cornell_scene(True), engine.push_constants(W, H, singleRender=1, sampleLimit=2), W, H)
    got = _run(2, W, H, tmp_path)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_eight_ranks_stitch_an_uneven_frame(tmp_path):
    """The 8-GPU partition (rows r, r+8, ...) on eight gloo ranks, 27 rows: three ranks get four rows, five get three."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    from ray_tracer_amd import engine
    from util import cornell_scene
    W, H = 32, 27
    full, _ = pyoracle.render(cornell_scene(True), engine.push_constants(W, H, singleRender=1, sampleLimit=2), W, H)
    got = _run(8, W, H, tmp_path)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_row_assignment_covers_every_row_once():
    from ray_tracer_amd import tiling
    for H in (1, 7, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            rows = sorted(y for r in range(world) for y in tiling.rows_of_rank(H, r, world))
            assert rows == list(range(H))
            assert max(len(tiling.rows_of_rank(H, r, world)) for r in range(world)) == tiling.max_rows(H, world)
    assert [len(tiling.rows_of_rank(1080, r, 8)) for r in range(8)] == [135] * 8
