#!/usr/bin/env python3
"""tools/exp_near_window.py -- drive tools/near_window_bench.hip: the near part of the SURVEY 8(d) random model with a row
block's window of x in LDS, against the tile kernel on the same entries.

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/near_window_bench.hip -o tools/libnear_window_bench.so
    python tools/exp_near_window.py [--rows-log2 24] [--cap 16] [--steps 50]

The plan (rows of a block of 8192 sorted by length, slices of 64 rows stored column-major, long rows apart) is built here with
numpy.  Development aid, not part of the library or the tests.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))
RB, BAND, WAVES = 8192, 4096, 16        # (--rb / --waves change the first and the last)
VALID, END = np.uint16(0x8000), np.uint16(0x4000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-log2", type=int, default=24)
    ap.add_argument("--cap", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--lib", default="libnear_window_bench.so")
    ap.add_argument("--rb", type=int, default=8192)
    ap.add_argument("--waves", type=int, default=16)
    ap.add_argument("--xcd", type=int, default=0)
    ap.add_argument("--longcap", type=int, default=1024)
    ap.add_argument("--balance", action="store_true", help="deal a block's long rows to its wavefronts longest first")
    ap.add_argument("--emulate", action="store_true", help="no GPU: walk the plan the way the kernel does, in Python (small --rows-log2)")
    args = ap.parse_args()
    global RB, WAVES
    RB, WAVES = args.rb, args.waves
    if not args.emulate:
        import torch                       # before the library: one HIP runtime in the process
    import smvp_toolkit_amd as sm
    rows = 1 << args.rows_log2
    assert rows % RB == 0
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, threads=16)
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    near = np.abs(ci.astype(np.int64) - row_of) <= BAND
    nci, nv, nrow = ci[near].astype(np.int64), v[near], row_of[near]
    lens = np.bincount(nrow, minlength=rows)
    nrp = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=nrp[1:])
    short = lens <= args.cap
    slen = np.where(short, lens, 0)
    nblocks = rows // RB
    block = np.arange(rows) // RB
    order = np.lexsort((np.arange(rows), -slen, block))        # inside a block: longest first, ties in row order
    sl = slen[order]
    nss = rows // 64                                            # short slices, in block order
    s_width = sl.reshape(nss, 64).max(axis=1).astype(np.int64)
    s_block = np.arange(nss) // (RB // 64)
    s_k = np.arange(nss) % (RB // 64)
    blk_short = np.bincount(s_block[s_width > 0], minlength=nblocks).astype(np.int32)
    pos_of_row = np.empty(rows, dtype=np.int64)
    pos_of_row[order] = np.arange(rows)
    perm16 = (order - block[order] * RB).astype(np.uint16)
    perm16[~short[order]] = 0xFFFF                              # long rows: written by their wavefront instead
    long_rows = np.nonzero(~short)[0]
    if args.balance:                                            # inside a block: most steps first
        long_rows = long_rows[np.lexsort((long_rows, -((lens[long_rows] + 63) // 64), long_rows // RB))]
    l_block = long_rows // RB
    blk_long_ptr = np.searchsorted(l_block, np.arange(nblocks + 1)).astype(np.int32)
    l_q = np.arange(long_rows.size) - blk_long_ptr[l_block]
    l_width = (lens[long_rows] + 63) // 64
    # all slices: short ones, then long rows; a wavefront's run = its short slices, then its long rows
    a_block = np.concatenate([s_block, l_block])
    a_wave = np.concatenate([s_k % WAVES, l_q % WAVES])
    a_sec = np.concatenate([np.zeros(nss, dtype=np.int64), np.ones(long_rows.size, dtype=np.int64)])
    a_ord = np.concatenate([s_k // WAVES, l_q // WAVES])
    a_width = np.concatenate([s_width, l_width])
    srt = np.lexsort((a_ord, a_sec, a_wave, a_block))
    steps_off = np.zeros(srt.size + 1, dtype=np.int64)
    np.cumsum(a_width[srt], out=steps_off[1:])
    off = np.empty(srt.size, dtype=np.int64)
    off[srt] = steps_off[:-1]
    gw = (a_block * WAVES + a_wave)
    first = np.searchsorted(gw[srt], np.arange(nblocks * WAVES))
    wave_ptr = steps_off[first]
    wave_n1 = np.bincount(gw[a_sec == 0], weights=a_width[a_sec == 0], minlength=nblocks * WAVES).astype(np.int32)
    wave_n2 = np.bincount(gw[a_sec == 1], weights=a_width[a_sec == 1], minlength=nblocks * WAVES).astype(np.int32)
    total = int(steps_off[-1]) * 64
    sval = np.zeros(total)
    sword = np.zeros(total, dtype=np.uint16)
    wbase = np.maximum(0, block * RB - BAND)
    es = short[nrow]                                            # entries of short rows
    k = (np.arange(nci.size) - nrp[:-1][nrow])
    p = pos_of_row[nrow[es]]
    dest = (off[p // 64] + k[es]) * 64 + p % 64
    sval[dest] = nv[es]
    sword[dest] = (nci[es] - wbase[nrow[es]]).astype(np.uint16) | VALID
    el = ~es
    which = np.full(rows, -1, dtype=np.int64)
    which[long_rows] = np.arange(long_rows.size)
    lidx = which[nrow[el]]                                      # which long row
    dest = (off[nss + lidx] + k[el] // 64) * 64 + k[el] % 64
    sval[dest] = nv[el]
    sword[dest] = (nci[el] - wbase[nrow[el]]).astype(np.uint16) | VALID
    nz = a_width > 0
    last = ((off[nz] + a_width[nz] - 1) * 64)[:, None] + np.arange(64)[None, :]
    sword[last.ravel()] |= END
    print("near part: %d entries; %d in %d short rows (<= %d), %d in %d long rows; %d slots (+%.1f %%)" % (
        nci.size, int(es.sum()), int(short.sum()), args.cap, int(el.sum()), long_rows.size, total, 100.0 * (total / nci.size - 1)))

    x = sm.vector_random(rows)
    steps_total = total // 64
    assert np.all(wave_ptr + wave_n1 + wave_n2 <= steps_total) and perm16[perm16 != 0xFFFF].max() < RB
    wl = np.minimum(rows, (np.arange(nblocks) + 1) * RB + BAND) - np.maximum(0, np.arange(nblocks) * RB - BAND)
    step_block = np.repeat(a_block[srt], a_width[srt])
    assert np.all((sword & 0x3FFF).reshape(-1, 64).max(axis=1) < wl[step_block])
    prod = nv * x[nci]
    starts = np.minimum(nrp[:-1], nci.size - 1)
    ref = np.add.reduceat(prod, starts) * (lens > 0)
    scale = np.add.reduceat(np.abs(prod), starts) * (lens > 0)
    if args.emulate:
        y = np.full(rows, np.nan)
        lr16 = long_rows - l_block * RB
        for b in range(nblocks):
            R0, wb = b * RB, max(0, b * RB - BAND)
            pm = perm16[b * RB:(b + 1) * RB]
            for pp in range(blk_short[b] * 64, RB):
                if pm[pp] != 0xFFFF:
                    y[R0 + pm[pp]] = 0.0
            for w in range(WAVES):
                g = b * WAVES + w
                acc = np.zeros(64)
                ks = kl = 0
                for j in range(wave_n1[g] + wave_n2[g]):
                    at = (wave_ptr[g] + j) * 64
                    wd = sword[at:at + 64]
                    ok = (wd & 0x8000) != 0
                    acc[ok] += sval[at:at + 64][ok] * x[wb + (wd[ok] & 0x3FFF)]
                    if wd[0] & 0x4000:
                        if j < wave_n1[g]:
                            r = pm[(w + WAVES * ks) * 64:(w + WAVES * ks) * 64 + 64]
                            y[R0 + r[r != 0xFFFF]] = acc[r != 0xFFFF]
                            ks += 1
                        else:
                            y[R0 + lr16[blk_long_ptr[b] + w + WAVES * kl]] = acc.sum()
                            kl += 1
                        acc[:] = 0
        err = np.abs(y - ref) / np.maximum(scale, 1e-300)
        print("emulated: max normwise error %.3g, nan %d" % (np.nanmax(err), int(np.isnan(y).sum())))
        return
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", args.lib))
    lib.near_window_run.restype = ctypes.c_float
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d = dict(x=dev(x), y=torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda"), wave_ptr=dev(wave_ptr), wave_n1=dev(wave_n1),
             wave_n2=dev(wave_n2), blk_short=dev(blk_short), perm16=dev(perm16.view(np.int16)), sval=dev(sval), sword=dev(sword.view(np.int16)),
             blk_long_ptr=dev(blk_long_ptr), long_row16=dev(np.concatenate([(long_rows - l_block * RB).astype(np.uint16), np.zeros(1, np.uint16)]).view(np.int16)))
    P = lambda name: ctypes.c_void_p(d[name].data_ptr())

    def run(reps, variant=0):
        return lib.near_window_run(P("x"), P("y"), rows, rows, P("wave_ptr"), P("wave_n1"), P("wave_n2"), P("blk_short"), P("perm16"), P("sval"),
                                   P("sword"), P("blk_long_ptr"), P("long_row16"), nblocks, reps, variant, args.xcd)

    assert run(0) == 0.0
    torch.cuda.synchronize()
    got = d["y"].cpu().numpy()
    err = np.abs(got - ref) / np.maximum(scale, 1e-300)
    print("max normwise error vs numpy: %.3g (nan: %d)" % (np.nanmax(err), int(np.isnan(got).sum())))
    assert not np.isnan(got).any() and err.max() < 1e-12
    ms = run(args.steps)
    moved = total * 10 + rows * (8 + 2) + 2 * rows * 8
    print("near_window: %.4f ms per product; %.0f entries/us; %.2f GB of streams + window + y = %.0f GB/s" % (
        ms, nci.size / ms / 1e3, moved / 1e9, moved / ms / 1e6), flush=True)
    print("near_window, y stored in slice order (timing only): %.4f ms" % run(args.steps, 1), flush=True)
    print("near_window, ordinary y stores: %.4f ms" % run(args.steps, 2), flush=True)
    assert np.diff(blk_long_ptr).max() <= args.longcap
    d["y"].fill_(float("nan"))
    run(0, 3)
    torch.cuda.synchronize()
    got3 = d["y"].cpu().numpy()
    print("near_window, y through LDS: bit-equal to the first form: %s" % bool(np.array_equal(got3, got)))
    print("near_window, y through LDS: %.4f ms" % run(args.steps, 3), flush=True)
    print("   ... without the window's load (timing only): %.4f ms" % run(args.steps, 4), flush=True)
    print("   ... without the y phase (timing only): %.4f ms" % run(args.steps, 5), flush=True)

    # the tile kernel on the same entries
    A = sm.CsrMatrix(rows, rows, dev(nrp.astype(np.int32)), dev(nci.astype(np.int32)), dev(nv))
    A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
    dy = torch.empty(rows, dtype=torch.float64, device="cuda")
    for _ in range(5):
        A.spmv(d["x"], dy)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        A.spmv(d["x"], dy)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.4f ms per product" % (A.describe()[0], e0.elapsed_time(e1) / args.steps))
    print("largest difference between the two: %.3g (normwise)" % np.max(np.abs(dy.cpu().numpy() - got) / np.maximum(scale, 1e-300)))


if __name__ == "__main__":
    main()
