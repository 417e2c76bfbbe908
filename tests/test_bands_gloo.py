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
