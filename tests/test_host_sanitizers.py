"""The host side of the library under sanitizers (SURVEY 5 "race detection / sanitizers").

`make -C smvp-toolkit_amd host-san` builds the seven host translation units (no HIP in them) + cli/host_san_driver.cpp with
g++ -fsanitize=address,undefined and with -fsanitize=thread.  The driver runs everything the host library does with an input
file -- both readers, COO -> CSR -> COO, COO -> TJDS, symmetric expansion, the binary cache (written, read, truncated,
bit-flipped), CISR with 1 / 16 / 3 slots, the report writer -- and the generators.  A sanitizer report ends the process with a
non-zero status and its text on stderr; an input the library REJECTS is a pass.  CPU only.
"""
import gzip
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import ROOT, SAMPLES
from test_host_library import BANNERS

PKG = os.path.join(ROOT, "smvp-toolkit_amd")
ASAN = os.path.join(PKG, "build", "san", "host_san_asan")
TSAN = os.path.join(PKG, "build", "san", "host_san_tsan")


@pytest.fixture(scope="module")
def san_builds():
    p = subprocess.run(["make", "-C", PKG, "host-san"], capture_output=True, text=True)
    if p.returncode != 0:
        if "cannot find" in p.stderr and ("asan" in p.stderr or "tsan" in p.stderr or "ubsan" in p.stderr):
            pytest.skip("this toolchain has no sanitizer runtime: " + p.stderr[-300:])
        raise AssertionError(p.stderr[-3000:])
    return ASAN, TSAN


def run(binary, args, env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
             TSAN_OPTIONS="halt_on_error=1")
    e.update(env or {})
    p = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0 and "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, \
        "%s %s\n%s\n%s" % (os.path.basename(binary), args, p.stdout[-1500:], p.stderr[-4000:])
    return p.stdout


def plain_sample(name, tmp_path):
    src = ob.fixture_path(name)
    return src


@pytest.mark.parametrize("name", SAMPLES + ["badfile.mtx"])
def test_sample_files_under_asan_and_ubsan(san_builds, tmp_path, name):
    out = run(ASAN, ["file", plain_sample(name, tmp_path), tmp_path])
    if name == "badfile.mtx":
        assert "header: rc 12" in out          # MM_PREMATURE_EOF like the reference (mmio.c:96)
    else:
        assert "csr_from_coo: rc 0" in out and "tjds_from_coo: rc 0" in out and "cisr 16 slots: rc 0" in out
        assert out.count("crafted cache") == 4 and "cisr 1 slots: rc 6" in out      # the reference's "overran" exit (main-cli.c:596-600)


@pytest.mark.parametrize("i", range(len(BANNERS)))
def test_malformed_banners_and_size_lines_under_asan(san_builds, tmp_path, i):
    text, code, tc = BANNERS[i]
    f = tmp_path / "b.mtx"
    f.write_text(text)
    out = run(ASAN, ["file", f, tmp_path])
    assert ("header: rc %d " % code) in out


def test_more_malformed_inputs_under_asan(san_builds, tmp_path):
    """Entries outside the matrix, a size line that promises more than the file holds, huge counts, a line of 5000 characters,
    non-numbers where numbers belong, a symmetric file with entries above the diagonal, duplicates."""
    head = "%%MatrixMarket matrix coordinate real general\n"
    cases = [
        head + "3 3 2\n1 1 1.0\n9 9 2.0\n",
        head + "3 3 5\n1 1 1.0\n",
        head + "3 3 2147483647\n1 1 1.0\n",
        head + "-3 3 1\n1 1 1.0\n",
        head + "3 3 2\n1 x 1.0\n2 2 abc\n",
        head + "%" + "c" * 5000 + "\n2 2 1\n1 1 1\n",
        head + "2 2 1\n" + "1" * 400 + " 1 1.0\n",
        "%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 3 1.0\n3 1 2.0\n2 2 3.0\n",
        head + "2 2 4\n1 1 1\n1 1 2\n1 1 3\n1 1 4\n",
        "%%MatrixMarket matrix coordinate pattern general\n4 4 3\n1 1\n2 2 7\n3 3\n",
        head + "0 0 0\n",
        head + "1 1 1\n1 1 1e999\n",
    ]
    for k, text in enumerate(cases):
        f = tmp_path / ("m%d.mtx" % k)
        f.write_text(text)
        run(ASAN, ["file", f, tmp_path])
        run(ASAN, ["file", f, tmp_path], env={"SMVP_MM_THREADS": "4"})


def _fnv1a(data):
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


def test_crafted_cache_files_under_asan(san_builds, tmp_path):
    """Cache files whose checksum MATCHES but whose arrays do not hold together (a decreasing row pointer, a column outside
    the matrix, a negative one): refused by the reader; nothing reads or writes out of bounds on the way."""
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
        head[48:56] = struct.pack("<Q", _fnv1a(payload))
        return bytes(head) + payload

    assert "cache arrays: rc 0" in run(ASAN, ["cache", cache, mtx])
    bad_rp = rp.copy()
    bad_rp[3], bad_rp[4] = rp[4] + 5, rp[3]
    bad_ci = ci.copy()
    bad_ci[7] = n + 1000
    neg_ci = ci.copy()
    neg_ci[0] = -1
    huge_rp = rp.copy()
    huge_rp[5] = 2 ** 31 - 1
    for rp2, ci2 in ((bad_rp, ci), (rp, bad_ci), (rp, neg_ci), (huge_rp, ci)):
        open(cache, "wb").write(crafted(rp2, ci2))
        assert "cache arrays: rc 1" in run(ASAN, ["cache", cache, mtx])
    # a header that lies about the sizes
    for off, val in ((16, 2 ** 31 - 1), (24, -5), (16, 0)):
        head = bytearray(good)
        head[off:off + 4] = struct.pack("<i", val)
        open(cache, "wb").write(bytes(head))
        run(ASAN, ["cache", cache, mtx])


def test_generators_and_edge_cases_under_asan_and_tsan(san_builds, tmp_path):
    assert "version 0.6.4" in run(ASAN, ["synth", tmp_path])
    assert "version 0.6.4" in run(TSAN, ["synth", tmp_path])        # smvp_synth_fill with 4 threads


def test_parallel_tokeniser_under_tsan(san_builds, tmp_path):
    """The parallel Matrix Market tokeniser (files >= 8 MB, or SMVP_MM_THREADS): every thread parses its own stretch of the
    text into the shared entry array -- neighbouring threads write different FIELDS of one entry at a chunk seam.  Under
    ThreadSanitizer on memplus.mtx (forced to 2 / 7 / 16 threads), on a pattern file, and on a generated 9 MB file that
    takes the parallel path by its size alone."""
    for name, counts in (("memplus.mtx", (2, 16)), ("pwt.mtx", (7,))):
        for threads in counts:
            out = run(TSAN, ["file", ob.fixture_path(name), tmp_path], env={"SMVP_MM_THREADS": str(threads)})
            assert "entries: rc 0" in out and "csr_from_coo: rc 0" in out
    rng = np.random.default_rng(3)
    n, nnz = 200_000, 560_000
    r, c, v = rng.integers(1, n + 1, nnz), rng.integers(1, n + 1, nnz), rng.random(nnz)
    big = tmp_path / "big.mtx"
    with open(big, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (n, n, nnz))
        f.write("".join("%d %d %.9f\n" % t for t in zip(r, c, v)))
    assert os.path.getsize(big) >= 8 << 20
    out = run(TSAN, ["file", big, tmp_path])
    assert "entries: rc 0" in out and "entries inside the matrix: 1" in out
    out = run(ASAN, ["file", big, tmp_path])
    assert "entries: rc 0" in out
