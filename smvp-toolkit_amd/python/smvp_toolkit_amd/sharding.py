"""Row-block sharding of one SpMV across the GPUs of a node.

Rank g owns rows [bounds[g], bounds[g+1]) of A -- its own slice of val/col_ind, a
rebased row_ptr, the whole operand x -- and produces that slice of y.  The one
exchange step is an all-gather of the y slices (RCCL over xGMI on the GPUs,
gloo in the CPU tests).  The reference has no counterpart: it is one thread.
"""
import numpy as np


def equal_row_bounds(rows_total, world):
    """Row blocks of (almost) equal height: good when row lengths are i.i.d."""
    return np.array([rows_total * g // world for g in range(world + 1)], dtype=np.int64)


def gather_counts(bounds):
    """(rows per rank, padded block height used on the wire)."""
    counts = np.diff(bounds)
    return counts, int(counts.max()) if len(counts) else 0


def allgather_y(dist, y_local, y_full, bounds, wire=None, group=None):
    """All-gather the row blocks of y into y_full (length bounds[-1]) on every rank.

    Equal blocks go straight into y_full with one all_gather_into_tensor.  Unequal
    blocks (nnz-balanced partitions) travel padded to the tallest block through
    `wire` (world * pad elements) and are compacted afterwards.
    """
    import torch

    counts, pad = gather_counts(bounds)
    world = len(counts)
    if world == 1 and not (hasattr(dist, "is_initialized") and dist.is_initialized()):
        y_full[:counts[0]].copy_(y_local[:counts[0]])
        return y_full
    if int(counts.min()) == pad:
        dist.all_gather_into_tensor(y_full[:pad * world], y_local[:pad], group=group)
        return y_full
    if wire is None:
        wire = torch.empty(world * pad, dtype=y_local.dtype, device=y_local.device)
    send = y_local
    if y_local.numel() != pad:
        send = torch.zeros(pad, dtype=y_local.dtype, device=y_local.device)
        send[:y_local.numel()].copy_(y_local)
    dist.all_gather_into_tensor(wire, send, group=group)
    for g in range(world):
        y_full[int(bounds[g]):int(bounds[g + 1])].copy_(wire[g * pad:g * pad + int(counts[g])])
    return y_full


def slice_csr(row_ptr, col_ind, val, r0, r1):
    """Rows [r0, r1) of a host CSR matrix with row_ptr rebased to 0."""
    a, b = int(row_ptr[r0]), int(row_ptr[r1])
    return (row_ptr[r0:r1 + 1] - row_ptr[r0]).astype(np.int32), col_ind[a:b], val[a:b]


def tile_block_diagonal(row_ptr, col_ind, val, cols, copy_begin, copy_end):
    """Rows of copies [copy_begin, copy_end) of kron(I_k, A): local row numbering, GLOBAL column numbering.

    Copy c holds A at rows [c*rows, (c+1)*rows) and columns [c*cols, (c+1)*cols); a rank that owns a run of
    copies therefore needs nothing from the others (row-block sharding at copy granularity).
    """
    n = copy_end - copy_begin
    nnz = int(row_ptr[-1])
    rp = (row_ptr[:-1][None, :].astype(np.int64) + (np.arange(n, dtype=np.int64) * nnz)[:, None]).reshape(-1)
    rp = np.concatenate([rp, [n * nnz]])
    if rp[-1] >= 2 ** 31:
        raise ValueError("block of %d copies holds %d entries: beyond 32-bit indices" % (n, rp[-1]))
    ci = (col_ind[None, :].astype(np.int64) +
          (np.arange(copy_begin, copy_end, dtype=np.int64) * cols)[:, None]).reshape(-1)
    if len(ci) and ci.max() >= 2 ** 31:
        raise ValueError("column index beyond 32 bits")
    return rp.astype(np.int32), ci.astype(np.int32), np.tile(val, n)


# ------------------------------------------------------------------ chunked exchange (overlap of product and all-gather)
def cyclic_chunk_rows(rows_total, world, chunks):
    """Block-cyclic row ownership: chunk c of rank r is the global row range [(c*world + r)*h, +h), h = ceil(rows /
    (world*chunks)), clipped to rows_total (the last ranges may be short or empty).

    With this layout the all-gather of chunk c of every rank lands as ONE contiguous run of the full vector,
    y_full[c*world*h : (c+1)*world*h], in natural row order: equal counts, no padding between ranks, no compaction --
    and the gather of chunk c can travel while chunk c+1 is being multiplied.  chunks = 1 is the plain row-block
    partition.  Returns (h, ranges) with ranges[r][c] = (row_begin, row_end).
    """
    h = max(1, -(-rows_total // (world * chunks)))
    ranges = [[(min(rows_total, (c * world + r) * h), min(rows_total, (c * world + r + 1) * h)) for c in range(chunks)]
              for r in range(world)]
    return h, ranges


class ChunkedExchange:
    """y blocks of one rank under cyclic_chunk_rows and their all-gather, plain or overlapped with the products.

    product(c, out) must enqueue chunk c's product into `out` (h elements; rows past the chunk's end are left
    untouched = the zero padding).  step(overlap): for every chunk, product then all-gather of that chunk.
    overlap=False waits for all products before the first gather starts (the un-overlapped form); overlap=True
    issues each gather asynchronously right behind its product (on the backend's own stream), so chunk c travels
    while chunk c+1 is computed; the caller's stream waits for all of them at the end.
    """

    def __init__(self, torch, dist, rows_total, world, rank, chunks, device, group=None):
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank, self.chunks, self.rows_total = world, rank, chunks, rows_total
        self.h, ranges = cyclic_chunk_rows(rows_total, world, chunks)
        self.ranges = ranges[rank]
        self.y_local = torch.zeros(chunks * self.h, dtype=torch.float64, device=device)
        self.y_full = torch.zeros(world * chunks * self.h, dtype=torch.float64, device=device)   # padded past rows_total

    def local(self, c):
        return self.y_local[c * self.h:(c + 1) * self.h]

    def _gather(self, c, async_op):
        dst = self.y_full[c * self.world * self.h:(c + 1) * self.world * self.h]
        if self.world == 1 and not (hasattr(self.dist, "is_initialized") and self.dist.is_initialized()):
            dst.copy_(self.local(c))
            return None
        return self.dist.all_gather_into_tensor(dst, self.local(c), group=self.group, async_op=async_op)

    def step(self, product, overlap, gather=True):
        works = []
        for c in range(self.chunks):
            product(c, self.local(c))
            if gather and overlap:
                works.append(self._gather(c, True))
        if gather and not overlap:
            for c in range(self.chunks):
                self._gather(c, False)
        for w in works:
            if w is not None:
                w.wait()
        return self.y_full[:self.rows_total]


# ------------------------------------------------------------------ how many chunks per rank?
XGMI_LINK_GBPS = 153.0   # one xGMI link, one direction; an MI355X has seven, one to every peer of an 8-GPU node


def gather_ms_bounds(piece_bytes, world):
    """(direct, ring) milliseconds for one all-gather in which every rank contributes `piece_bytes`: every GPU must
    receive world - 1 pieces -- each over its own link at best (direct), all through one link at worst (a ring)."""
    if world <= 1:
        return 0.0, 0.0
    direct = piece_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3
    return direct, direct * (world - 1)


def overlapped_step_ms(product_ms_total, chunks, gather_ms_per_chunk):
    """End of the last all-gather when chunk c's gather is issued right behind its product and the gathers run one
    after the other on the communication stream (what ChunkedExchange.step(overlap=True) and the C layer do)."""
    p = product_ms_total / chunks
    t = 0.0
    for c in range(chunks):
        t = max(t, (c + 1) * p) + gather_ms_per_chunk
    return t


def choose_chunks(product_ms, y_bytes_per_rank, world, gather_ms=None):
    """Chunks per rank for the overlapped product / all-gather step.

    product_ms: {chunks: MEASURED ms of this rank's local products when its block is cut into that many chunks} -- the
    column sweep pays for every chunk (each pulls all of x into the L2s again), so more chunks are not free;
    y_bytes_per_rank: 8 * rows per rank, what the rank contributes per product; gather_ms: {chunks: measured ms of ONE
    chunk's all-gather} where it could be measured (world > 1), else the two link models of gather_ms_bounds.
    With a measurement the choice is the shortest estimated step; without, the candidate with the smallest worst-case
    regret over the two link models.  Returns {"chosen", "estimates_ms": {chunks: {...}}, "inputs": {...}}.
    """
    cands = sorted(product_ms)
    est = {}
    for c in cands:
        if gather_ms and c in gather_ms:
            est[c] = {"measured_gather": overlapped_step_ms(product_ms[c], c, gather_ms[c])}
        else:
            d, r = gather_ms_bounds(y_bytes_per_rank / c, world)
            est[c] = {"direct_links": overlapped_step_ms(product_ms[c], c, d), "one_link_ring": overlapped_step_ms(product_ms[c], c, r)}
    models = sorted({m for e in est.values() for m in e})
    best = {m: min(est[c][m] for c in cands if m in est[c]) for m in models}
    regret = {c: max(est[c][m] - best[m] for m in models if m in est[c]) for c in cands}
    chosen = min(cands, key=lambda c: (regret[c], c))
    return {"chosen": chosen,
            "estimates_ms": {c: {m: round(v, 4) for m, v in est[c].items()} for c in cands},
            "inputs": {"product_ms_by_chunks": {c: round(product_ms[c], 4) for c in cands}, "y_bytes_per_rank": y_bytes_per_rank,
                       "world": world, "link_GBps": XGMI_LINK_GBPS,
                       "gather_ms_by_chunks": {c: round(v, 4) for c, v in gather_ms.items()} if gather_ms else None},
            "rule": "shortest estimated overlapped step" if gather_ms else
                    "smallest worst-case regret over the direct-links and one-link-ring models of the all-gather"}
