"""The N > 1 path on CPU: cyclic band partition, gather to rank 0 (gloo, world_size 2 and 3) and frame reassembly.
The per-rank renderer is stubbed by the CPU oracle here (test infrastructure standing in for rt_render_rows, whose
own band selection is checked on the GPU in test_gpu_parity.py); the product code under test is raytracer_amd.bands."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import bands, scenes

HERE = os.path.dirname(os.path.abspath(__file__))


def test_band_partition_covers_every_row_once():
    for H, band, n in [(100, 8, 3), (2160, 16, 8), (17, 16, 2), (5, 8, 4), (64, 16, 1)]:
        seen = []
        for r in range(n):
            rows = bands.rank_rows(H, band, n, r)
            assert len(rows) <= bands.rows_per_rank(H, band, n)
            seen += rows
        assert sorted(seen) == list(range(H))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, H, W, band, q):
    sys.path.insert(0, HERE)
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.default4()
    samples, depth, seed = 2, 10, 7
    rows = bands.rank_rows(H, band, world, rank)
    rpr = bands.rows_per_rank(H, band, world)
    block = torch.zeros((rpr, W, 3), dtype=torch.uint8)
    # this rank's bands, band by band (what rt_render_rows does in one call with band_rows / band_stride)
    k = 0
    for b0 in range(rank * band, H, band * world):
        n = min(band, H - b0)
        rgb, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=b0, rows=n, want_emission=False)
        block[k:k + n] = torch.from_numpy(rgb); k += n
    assert k == len(rows)
    frame = bands.gather_frame(block, H, band)
    if rank == 0:
        want, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, want_emission=False)
        q.put(bool(np.array_equal(frame.numpy(), want)))
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band", [(2, 72, 8), (3, 50, 16)])
def test_gather_frame_gloo(world, H, band):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, 64, band, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


# ---- frames in flight (raytracer_amd/pipeline.py) ----------------------------------------------------------------------

def _pattern_rows(frame_index, ys, W):
    """a frame whose pixel (x, y) of frame k is a function of all three: any mix-up of slots, ranks or rows shows"""
    y = torch.tensor(ys, dtype=torch.int64).view(-1, 1, 1)
    x = torch.arange(W, dtype=torch.int64).view(1, -1, 1)
    c = torch.arange(3, dtype=torch.int64).view(1, 1, -1)
    return ((x * 7 + y * 13 + c * 29 + frame_index * 101) % 251).to(torch.uint8)


def _pipeline_run(rank, world, H, W, band, in_flight, n_frames, jitter, blocks_per_worker=2):
    import random
    import time
    from raytracer_amd.pipeline import FramePipeline
    rows = bands.rank_rows(H, band, world, rank)
    rpr = bands.rows_per_rank(H, band, world)
    blocks = [torch.zeros((rpr, W, 3), dtype=torch.uint8) for _ in range(in_flight * blocks_per_worker)]
    rnd = random.Random(1234 + rank)
    busy = [0] * in_flight

    def render(worker, block, k):
        assert busy[worker] == 0                    # a producer's context renders one frame at a time
        busy[worker] = 1
        if jitter:
            time.sleep(rnd.random() * 0.01)         # frames finish out of order
        block[:len(rows)] = _pattern_rows(k, rows, W)
        busy[worker] = 0

    seen = []
    pipe = FramePipeline(render, blocks, H, band, workers=in_flight)
    last = pipe.run(n_frames, on_frame=lambda k, f: seen.append((k, bool(torch.equal(f, _pattern_rows(k, list(range(H)), W))))), first_frame=5)
    if rank == 0:
        assert [k for k, _ in seen] == list(range(5, 5 + n_frames))
        assert all(ok for _, ok in seen)
        assert torch.equal(last, _pattern_rows(5 + n_frames - 1, list(range(H)), W))
        # a second run on the same pipeline reuses its buffers
        again = pipe.run(2)
        assert torch.equal(again, _pattern_rows(1, list(range(H)), W))
    else:
        assert last is None and not seen
        pipe.run(2)
    return True


@pytest.mark.parametrize("in_flight,n_frames,blocks_per_worker", [(1, 3, 1), (3, 7, 1), (3, 11, 2), (4, 2, 2)])
def test_frame_pipeline_single_rank(in_flight, n_frames, blocks_per_worker):
    assert _pipeline_run(0, 1, 40, 24, 8, in_flight, n_frames, jitter=True, blocks_per_worker=blocks_per_worker)


def test_frame_pipeline_hands_on_a_producer_error():
    from raytracer_amd.pipeline import FramePipeline
    blocks = [torch.zeros((8, 4, 3), dtype=torch.uint8) for _ in range(2)]

    def render(worker, block, k):
        if k == 3:
            raise RuntimeError("frame 3 failed")

    with pytest.raises(RuntimeError, match="frame 3 failed"):
        FramePipeline(render, blocks, 8, 8).run(6)


def _pipeline_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = _pipeline_run(rank, world, 72, 32, 8, 3, 8, jitter=True)
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_pipeline_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
