#!/usr/bin/env python3
"""tests/golden/make_arrays.py -- golden integer arrays and full-precision y vectors for the sample matrices.

SURVEY 8(c) lists these fixtures (row_ptr/col_ind, perm/start_pos/row_ind, y at full precision, the reference's
defective TJDS y).  The reference program cannot be run in this image (main-cli.c needs libpopt), so they are
produced by the CPU oracle (oracle/smvp_oracle.c) AFTER it has been pinned against every committed reference report
(tests/test_oracle_golden.py): y_csr and y_tjds_refquirks printed with "%g" equal those reports line for line.
The arrays are committed so that the converters and kernels stay pinned even if the oracle is edited later.

    python tests/golden/make_arrays.py        # rewrites tests/golden/arrays/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

REPORTS = {"ibm32.mtx": ("1615284655", "1615284655"), "memplus.mtx": ("1615284663", "1615284665"),
           "pwt.mtx": ("1615284671", "1615284679"), "curtis54.mtx": ("1615284695", "1615284695"),
           "pdp08-pg4.mtx": ("1619162887", None)}

for name, (csr_stamp, tjds_stamp) in REPORTS.items():
    rc, tc, m, n, coo = ob.mm_read_coo(ob.fixture_path(name))
    assert rc == 0
    row_ptr, col_ind, val = ob.csr_build(coo, m)
    y_csr = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    assert ob.fmt_g(y_csr) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % csr_stamp))
    t = ob.tjds_build(coo, m, n)
    y_q = ob.tjds_spmv(t, np.ones(n), refquirks=True)
    if tjds_stamp:
        assert ob.fmt_g(y_q) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_TJDS_%s.txt" % tjds_stamp))
    out = os.path.join(HERE, "arrays", name.replace(".mtx", ".npz"))
    np.savez_compressed(out, rows=m, cols=n, typecode=tc, row_ptr=row_ptr, col_ind=col_ind, perm=t.perm,
                        start_pos=t.start_pos, row_ind=t.row_ind, num_diag=t.num_diag,
                        ref_num_tjdiag=t.ref_num_tjdiag, last_diag_single=t.last_diag_single,
                        y_csr=y_csr, y_tjds_refquirks=y_q)
    print("%-16s %8d bytes" % (os.path.basename(out), os.path.getsize(out)))
