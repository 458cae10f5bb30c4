"""CISR .coe export (main-cli.c:473-729), host only.

PARITY UNPINNED: the reference commits no .coe output and main-cli.c cannot be built in this image (libpopt), so the
checks are (1) a known answer traced by hand through the reference's loops, (2) the product's cursor-based
generator against the oracle's literal restatement (slot-group table) on every sample matrix and on random ones,
(3) structural properties of the file, (4) the CLI surface (-g, -s)."""
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import SAMPLES

HEADER = ("\n;*********************************************\n;* CISR COE File for Vivado Single-Port BRAM *"
          "\n;*********************************************\n\n;Generated with a slot/channel count of: %d\n\n"
          "memory_initialization_radix=16;\nmemory_initialization_vector=\n00aaaaaaaa,\n")


def test_hand_traced_known_answer(tmp_path):
    """3 x 3, rows (2 3 .)(. 5 .)(7 1 4), 2 slots.  Traced through main-cli.c:524-607: slot 0 walks row 0 then idles,
    slot 1 walks row 1 then row 2; five groups (the last all padding); row lengths 2,1 | 3 behind the first two words."""
    coo = sm.make_coo([2, 0, 1, 0, 2, 2], [0, 0, 1, 2, 2, 1], [7.0, 2.0, 5.0, 3.0, 4.0, 1.0])
    want = HEADER % 2 + "".join(w + ",\n" for w in (
        "0100200000", "0210021001", "0100500101", "0210030000",      # group 0: (2,c0,s0) lens 2,1 | (5,c1,s1) len 3
        "0100300200", "0100700001",                                    # group 1: (3,c2,s0) | (7,c0,s1)
        "0100000000", "0100100101",                                    # group 2: padding   | (1,c1,s1)
        "0100000000", "0100400201",                                    # group 3: padding   | (4,c2,s1)
        "0100000000", "0100000001")) + "03ffffffff;\n\n"               # group 4: padding x2 (slot number stays)
    assert sm.cisr_coegen(coo, 3, 2, str(tmp_path / "a.coe")) == want
    assert ob.cisr_coegen(coo, 3, 2, str(tmp_path / "b.coe")) == (0, want)


@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("slots", [2, 16, 32])
def test_product_matches_oracle_on_samples(name, slots, tmp_path):
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    rc, want = ob.cisr_coegen(coo, m, slots, str(tmp_path / "o.coe"))
    assert rc == 0
    got = sm.cisr_coegen(coo, m, slots, str(tmp_path / "p.coe"))
    assert got == want
    # structure: header, one 01 word per (group, slot), ceil(rows / 2) row-length words, terminator
    assert got.startswith(HEADER % slots) and got.endswith("03ffffffff;\n\n")
    words = got[len(HEADER % slots):].split("\n")
    n01 = sum(w.startswith("01") for w in words)
    assert n01 % slots == 0 and sum(w.startswith("02") for w in words) == (m + 1) // 2
    # every stored entry appears exactly once (no sample has an empty row): for the small pattern matrices the value
    # field (1 << 20) is not overlapped by the column field (col << 8, main-cli.c:703), so the words can be counted
    if tc[2] == "P" and n < 4096:
        assert sum(w.startswith("01") and int(w[2:10], 16) >> 20 == 1 for w in words) == len(coo)
    # row lengths travel in row order
    rp, _, _ = sm.csr_from_coo(coo, m)
    lens = []
    for w in words:
        if w.startswith("02"):
            v = int(w[2:10], 16)
            lens.append((v >> 16) & 0xFFF)
            if (v >> 12) & 1:
                lens.append(v & 0xFFF)
    assert lens == np.diff(rp).tolist()


def test_one_slot_is_the_references_overrun_exit(tmp_path):
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("ibm32.mtx"))
    assert ob.cisr_coegen(coo, m, 1, str(tmp_path / "o.coe"))[0] == 1
    with pytest.raises(sm.SmvpError) as e:
        sm.cisr_coegen(coo, m, 1, str(tmp_path / "p.coe"))
    assert e.value.code == sm.ERR_UNSUPPORTED and "overran" in str(e.value)


@pytest.mark.parametrize("seed", range(12))
def test_product_matches_oracle_on_random_matrices(seed, tmp_path):
    """Random shapes incl. empty rows (where a slot emits the following row's first entry, main-cli.c:541), negative and
    large values (the (int) cast, :703), more slots than rows."""
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    lens = np.minimum(rng.integers(0, 7, rows), cols)
    r = np.repeat(np.arange(rows), lens)
    c = np.concatenate([rng.choice(cols, l, replace=False) for l in lens] + [np.zeros(0, int)])
    v = np.round(rng.uniform(-3000, 3000, len(r)), int(rng.integers(0, 3)))
    coo = sm.make_coo(r, c, v)[rng.permutation(len(r))]
    for slots in (2, 3, 8, 70):
        rc, want = ob.cisr_coegen(coo, rows, slots, str(tmp_path / "o.coe"))
        if rc == 1:
            with pytest.raises(sm.SmvpError):
                sm.cisr_coegen(coo, rows, slots, str(tmp_path / "p.coe"))
        else:
            assert sm.cisr_coegen(coo, rows, slots, str(tmp_path / "p.coe")) == want


def test_cli_cisr_flags(tmp_path):
    """-g prints the image to stdout after the [INFO] line (main-cli.c:490,689-728); -s sets the channel count; no GPU needed."""
    p = subprocess.run([sm.CLI_PATH, "-g", "-s", "4", "-d", str(tmp_path), ob.fixture_path("ibm32.mtx")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-500:]
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("ibm32.mtx"))
    assert "[INFO]\tConverting loaded content to CISR format." in p.stdout
    assert sm.cisr_coegen(coo, m, 4, str(tmp_path / "p.coe")) in p.stdout
    assert "[STOP]" in p.stdout.split("03ffffffff;")[1]
    p16 = subprocess.run([sm.CLI_PATH, "--cisr-gen", ob.fixture_path("ibm32.mtx")], capture_output=True, text=True)
    assert ";Generated with a slot/channel count of: 16\n" in p16.stdout            # default -s 16, main-cli.c:1228
    p1 = subprocess.run([sm.CLI_PATH, "-g", "-s", "1", ob.fixture_path("ibm32.mtx")], capture_output=True, text=True)
    assert p1.returncode == 1 and "[ERROR]\tslot_group_iter overran fInputNonZeros!" in p1.stdout
