"""The N>1 path on CPU: two gloo ranks, row-block sharding + all-gather of y.

The per-rank product is done by the oracle here (no GPU in this container); what is
under test is the partition, the independent per-block generation and the exchange.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
import smvp_toolkit_amd as sm
from smvp_toolkit_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "synthetic_equal_rows":
            M = 40_000
            bounds = sharding.equal_row_bounds(M, world)
            r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
            row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M, row_begin=r0, row_end=r1,
                                                 threads=1)
            x = np.random.default_rng(0).random(M)
        else:                       # a real matrix, partition balanced by bytes -> unequal blocks
            tc, M, N, coo = sm.mm_read_coo(ob.fixture_path("memplus.mtx"))
            rp, ci, v = sm.csr_from_coo(coo, M)
            bounds = sm.partition_rows(rp, world).astype(np.int64)
            r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
            row_ptr, col_ind, val = sharding.slice_csr(rp, ci, v, r0, r1)
            x = np.ones(N)
        y_full = torch.full((M,), float("nan"), dtype=torch.float64)
        if case == "synthetic_equal_rows":
            # like bench.py: the local block is a view of the full vector and is gathered in place
            y_local = y_full[r0:r1]
            y_local.copy_(torch.from_numpy(ob.csr_spmv(row_ptr, col_ind, val, x)))
        else:
            y_local = torch.from_numpy(ob.csr_spmv(row_ptr, col_ind, val, x))
        sharding.allgather_y(dist, y_local, y_full, bounds)
        if rank == 0:
            np.save(out, y_full.numpy())
        # every rank must hold the same full vector
        ref = y_full.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, y_full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["synthetic_equal_rows", "memplus_byte_balanced"])
def test_two_rank_row_block_spmv(case, tmp_path):
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), case, out), nprocs=2, join=True)
    y = np.load(out)
    if case == "synthetic_equal_rows":
        M = 40_000
        row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M, threads=1)
        x = np.random.default_rng(0).random(M)
    else:
        tc, M, N, coo = sm.mm_read_coo(ob.fixture_path("memplus.mtx"))
        row_ptr, col_ind, val = sm.csr_from_coo(coo, M)
        x = np.ones(N)
    assert np.array_equal(y, ob.csr_spmv(row_ptr, col_ind, val, x))


def _chunk_worker(rank, world, port, rows, chunks, overlap, out):
    """ChunkedExchange (block-cyclic row chunks, one all-gather per chunk) exactly as bench.py's config-4 leg uses it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = sharding.ChunkedExchange(torch, dist, rows, world, rank, chunks, "cpu")
        x = np.random.default_rng(7).random(rows)
        blocks = [sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 5, r0, r1, threads=1) for r0, r1 in ex.ranges]
        calls = []

        def product(c, out_view):
            r0, r1 = ex.ranges[c]
            calls.append(c)
            if r1 > r0:
                out_view[:r1 - r0].copy_(torch.from_numpy(ob.csr_spmv(*blocks[c], x)))

        for _ in range(2):                       # a second step must give the same vector (buffers are re-used)
            y = ex.step(product, overlap=overlap)
        assert calls == list(range(chunks)) * 2 and y.numel() == rows
        ref = y.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, y)               # every rank holds the same full vector
        if rank == 0:
            np.save(out, y.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,chunks,overlap", [(2, 1000, 1, False), (2, 1003, 4, True), (4, 4099, 4, True),
                                                        (4, 4099, 3, False), (4, 10, 4, True)])
def test_chunked_exchange_block_cyclic(world, rows, chunks, overlap, tmp_path):
    """world 2 and 4, chunk heights that do not divide the rows (short and empty last chunks), both step forms."""
    out = str(tmp_path / "y.npy")
    mp.spawn(_chunk_worker, args=(world, _free_port(), rows, chunks, overlap, out), nprocs=world, join=True)
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 5, threads=1)
    assert np.array_equal(np.load(out), ob.csr_spmv(rp, ci, v, np.random.default_rng(7).random(rows)))


def test_cyclic_chunk_rows_cover_every_row_once():
    for rows, world, chunks in ((103, 4, 3), (16, 8, 4), (1, 2, 2), (10_000_000, 8, 4), (0, 2, 2)):
        h, ranges = sharding.cyclic_chunk_rows(rows, world, chunks)
        seen = np.zeros(rows, dtype=np.int32)
        for r in range(world):
            for c, (a, b) in enumerate(ranges[r]):
                assert 0 <= a <= b <= rows and b - a <= h
                assert a == min(rows, (c * world + r) * h)      # where the all-gather of chunk c puts rank r's rows
                seen[a:b] += 1
        assert np.all(seen == 1)
    assert sharding.cyclic_chunk_rows(100, 4, 1)[1] == [[(0, 25)], [(25, 50)], [(50, 75)], [(75, 100)]]


def test_bounds_helpers():
    b = sharding.equal_row_bounds(10, 4)
    assert b.tolist() == [0, 2, 5, 7, 10]
    counts, pad = sharding.gather_counts(b)
    assert counts.tolist() == [2, 3, 2, 3] and pad == 3
    assert sharding.equal_row_bounds(1 << 24, 8).tolist() == [(1 << 21) * g for g in range(9)]


def test_choose_chunks_model():
    """Chunks per rank from measured product times and the two link models (or a measured gather): more chunks hide more of
    the exchange but the column sweep pays for each of them -- DESIGN 6's numbers for one rank's eighth of config 4."""
    pick = sharding.choose_chunks({1: 0.41, 2: 0.44, 4: 0.54}, 8 * 1.25e6, 8)
    assert pick["chosen"] == 2 and pick["inputs"]["gather_ms_by_chunks"] is None
    e = pick["estimates_ms"]
    assert e[1]["direct_links"] < e[4]["direct_links"] and e[4]["one_link_ring"] < e[1]["one_link_ring"]      # each model has its own favourite
    assert abs(e[1]["one_link_ring"] - (0.41 + 7 * 10e6 / 153e9 * 1e3)) < 1e-3
    # a measured gather decides alone; free chunks (a product that does not slow down) -> as many as offered when the gather is slow
    assert sharding.choose_chunks({1: 1.0, 2: 1.0, 4: 1.0}, 8e6, 2, {1: 0.8, 2: 0.4, 4: 0.2})["chosen"] == 4
    assert sharding.choose_chunks({1: 1.0, 2: 1.3, 4: 1.9}, 8e6, 2, {1: 0.1, 2: 0.05, 4: 0.03})["chosen"] == 1
    # one GPU: nothing to gather, one chunk
    assert sharding.choose_chunks({1: 0.4, 2: 0.5}, 8e6, 1)["chosen"] == 1
    assert sharding.overlapped_step_ms(0.4, 4, 0.3) == pytest.approx(0.1 + 4 * 0.3)       # gathers queue behind each other
    assert sharding.overlapped_step_ms(0.4, 4, 0.05) == pytest.approx(0.4 + 0.05)         # only the last one is exposed
