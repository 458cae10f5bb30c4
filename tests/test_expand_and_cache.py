"""optional symmetric expansion (off by default: the reference multiplies the stored triangle,
main-cli.c:1427-1441) and the binary cache of a loaded matrix.  CPU only; the GPU product of an expanded matrix is in
test_gpu_parity.py."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm


def full_product_numpy(coo, rows, sign):
    """A_full . 1 with numpy only: every stored entry once, every off-diagonal one mirrored."""
    y = np.zeros(rows)
    np.add.at(y, coo["row"], coo["val"])
    off = coo["row"] != coo["col"]
    np.add.at(y, coo["col"][off], sign * coo["val"][off])
    return y


def test_expand_pwt_mirrors_the_stored_triangle():
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("pwt.mtx"))
    assert tc == "MCPS" and len(coo) == 181313                       # pattern, symmetric, lower triangle stored
    full = sm.mm_expand_symmetric(tc, coo, m, n)
    ndiag = int((coo["row"] == coo["col"]).sum())
    assert len(full) == 2 * len(coo) - ndiag
    assert np.array_equal(full[:len(coo)], coo)                       # stored entries first, untouched
    mirrored = full[len(coo):]
    off = coo[coo["row"] != coo["col"]]
    assert np.array_equal(mirrored["row"], off["col"]) and np.array_equal(mirrored["col"], off["row"])
    assert np.array_equal(mirrored["val"], off["val"])
    # the expanded matrix is symmetric and its product with ones is what numpy gets from the fixture
    rp, ci, v = sm.csr_from_coo(full, m)
    y = ob.csr_spmv(rp, ci, v, np.ones(n))
    assert np.array_equal(y, full_product_numpy(coo, m, 1.0))
    rpT, ciT, vT = sm.csr_from_coo(sm.make_coo(full["col"], full["row"], full["val"]), m)
    assert np.array_equal(rp, rpT) and np.array_equal(ci, ciT) and np.array_equal(v, vT)
    # default path: the stored triangle only, like the reference's committed report
    rp0, ci0, v0 = sm.csr_from_coo(coo, m)
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284671.txt"))
    assert ob.fmt_g(ob.csr_spmv(rp0, ci0, v0, np.ones(n))) == want


def test_expand_general_skew_and_edge_cases():
    coo = sm.make_coo([0, 1, 2, 2], [0, 0, 1, 2], [1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(sm.mm_expand_symmetric("MCRG", coo, 3, 3), coo)                  # general: a copy
    skew = sm.mm_expand_symmetric("MCRK", coo, 3, 3)
    assert skew["row"].tolist() == [0, 1, 2, 2, 0, 1] and skew["col"].tolist() == [0, 0, 1, 2, 1, 2]
    assert skew["val"].tolist() == [1.0, 2.0, 3.0, 4.0, -2.0, -3.0]
    herm = sm.mm_expand_symmetric("MCRH", coo, 3, 3)                                        # real field: like symmetric
    assert herm["val"].tolist() == [1.0, 2.0, 3.0, 4.0, 2.0, 3.0]
    assert len(sm.mm_expand_symmetric("MCRS", coo[:0], 3, 3)) == 0
    with pytest.raises(sm.SmvpError):
        sm.mm_expand_symmetric("MCRS", coo, 3, 4)                                           # rectangular cannot be symmetric


def test_cache_round_trip_and_staleness(tmp_path):
    mtx = str(tmp_path / "m.mtx")
    shutil.copy(ob.fixture_path("memplus.mtx"), mtx)
    tc, m, n, coo = sm.mm_read_coo(mtx)
    rp, ci, v = sm.csr_from_coo(coo, m)
    cache = mtx + ".smvpbin"
    with pytest.raises(sm.SmvpError) as e:
        sm.cache_read_csr(cache, mtx)
    assert e.value.code == sm.ERR_IO                                   # no cache yet
    sm.cache_write_csr(cache, mtx, tc, m, n, rp, ci, v)
    tc2, expanded, m2, n2, rp2, ci2, v2 = sm.cache_read_csr(cache, mtx)
    assert (tc2, expanded, m2, n2) == (tc, False, m, n)
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and v2.tobytes() == v.tobytes()
    back = sm.coo_from_csr(m, rp2, ci2, v2)
    assert np.array_equal(np.sort(back, order=["row", "col"]), np.sort(coo, order=["row", "col"]))
    assert os.path.getsize(cache) == 64 + 4 * (m + 1) + 12 * len(coo)
    # one changed byte in the .mtx: the cache is refused
    with open(mtx, "r+b") as f:
        f.seek(os.path.getsize(mtx) - 2)
        f.write(b"7")
    with pytest.raises(sm.SmvpError) as e:
        sm.cache_read_csr(cache, mtx)
    assert e.value.code == sm.ERR_INVALID
    sm.cache_read_csr(cache, None)                                     # without the source check it still opens
    # a damaged payload is caught by its own checksum
    with open(cache, "r+b") as f:
        f.seek(64 + 4 * (m + 1) + 40)
        f.write(b"\x01\x02\x03")
    with pytest.raises(sm.SmvpError):
        sm.cache_read_csr(cache, None)
    # not a cache at all
    junk = str(tmp_path / "junk.smvpbin")
    open(junk, "wb").write(b"hello" * 40)
    with pytest.raises(sm.SmvpError) as e:
        sm.cache_read_csr(junk, None)
    assert e.value.code == sm.ERR_INVALID


def _fnv1a(data, h=0xcbf29ce484222325):
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


def test_crafted_cache_with_a_matching_checksum_is_refused(tmp_path):
    """The payload checksum is no integrity guarantee: a cache whose row pointer decreases, or whose column index lies
    outside the matrix, is refused even when its checksum has been recomputed to match (it would otherwise make
    smvp_coo_from_csr write, and the kernels gather, out of bounds)."""
    import struct

    mtx = str(tmp_path / "m.mtx")
    shutil.copy(ob.fixture_path("ibm32.mtx"), mtx)
    tc, m, n, coo = sm.mm_read_coo(mtx)
    rp, ci, v = sm.csr_from_coo(coo, m)
    cache = mtx + ".smvpbin"
    sm.cache_write_csr(cache, mtx, tc, m, n, rp, ci, v)
    good = open(cache, "rb").read()

    def crafted(rp2, ci2):
        payload = rp2.astype("<i4").tobytes() + ci2.astype("<i4").tobytes() + v.astype("<f8").tobytes()
        head = bytearray(good[:64])
        head[48:56] = struct.pack("<Q", _fnv1a(payload))          # payload_fnv1a, right behind mtx_bytes / mtx_fnv1a
        return bytes(head) + payload

    open(cache, "wb").write(crafted(rp, ci))
    sm.cache_read_csr(cache, mtx)                                       # the helper reproduces a valid file
    bad_rp = rp.copy()
    bad_rp[3], bad_rp[4] = rp[4] + 5, rp[3]                             # decreasing, still 0 ... nnz at the ends
    bad_ci = ci.copy()
    bad_ci[7] = n + 1000
    neg_ci = ci.copy()
    neg_ci[0] = -1
    for rp2, ci2 in ((bad_rp, ci), (rp, bad_ci), (rp, neg_ci)):
        open(cache, "wb").write(crafted(rp2, ci2))
        with pytest.raises(sm.SmvpError) as e:
            sm.cache_read_csr(cache, mtx)
        assert e.value.code == sm.ERR_INVALID
    with pytest.raises(sm.SmvpError):                                   # and the converter itself refuses such a row pointer
        sm.coo_from_csr(m, bad_rp, ci, v)
    # the CLI falls back to parsing the .mtx
    open(cache, "wb").write(crafted(rp, bad_ci))
    p = subprocess.run([sm.CLI_PATH, "-c", "--cache", "-d", str(tmp_path), mtx], capture_output=True, text=True)
    assert "taken from the binary cache" not in p.stdout and "Non-zero numbers contained in matrix" in p.stdout


def test_cli_cache_and_expand_flags_reach_the_loader(tmp_path):
    """No GPU here: the run stops at device selection, after the matrix has been loaded / cached."""
    mtx = str(tmp_path / "pwt.mtx")
    shutil.copy(ob.fixture_path("pwt.mtx"), mtx)
    p = subprocess.run([sm.CLI_PATH, "-c", "-n", "1", "-d", str(tmp_path), "--cache", "--expand-symmetric", mtx],
                       capture_output=True, text=True)
    assert "Symmetric storage expanded: 181313 stored entries -> 326107." in p.stdout
    assert "Binary cache written:" in p.stdout and os.path.exists(mtx + ".smvpbin")
    tc, expanded, m, n, rp, ci, v = sm.cache_read_csr(mtx + ".smvpbin", mtx)
    assert expanded and int(rp[-1]) == 326107
    p = subprocess.run([sm.CLI_PATH, "-c", "-n", "1", "-d", str(tmp_path), "--cache", "--expand-symmetric", mtx],
                       capture_output=True, text=True)
    assert "Matrix content taken from the binary cache" in p.stdout and "Non-zero numbers contained in matrix: \x1b[0m326107" in p.stdout
    # the same cache is NOT used when the expansion choice differs: the text is parsed again and the cache rewritten
    p = subprocess.run([sm.CLI_PATH, "-c", "-n", "1", "-d", str(tmp_path), "--cache", mtx], capture_output=True, text=True)
    assert "taken from the binary cache" not in p.stdout and "Non-zero numbers contained in matrix: \x1b[0m181313" in p.stdout
    assert not sm.cache_read_csr(mtx + ".smvpbin", mtx)[1]
