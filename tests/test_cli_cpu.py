"""Command-line surface of smvp-toolkit-cli that needs no GPU: flag parsing, error texts, exit codes.

Each expectation is the reference's behaviour at the cited main-cli.c lines.
"""
import subprocess

import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm


def run(*args):
    p = subprocess.run([sm.CLI_PATH, *args], capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


def test_no_arguments_prints_usage_to_stderr():          # main-cli.c:1267-1271
    rc, out, err = run()
    assert rc == 1 and err.startswith("Usage:") and out == ""


@pytest.mark.parametrize("args", [["-a", "-c", "x.mtx"], ["-c", "-a", "x.mtx"], ["-t", "--all-algs", "x.mtx"],
                                  ["--all-algs", "-g", "x.mtx"]])
def test_all_algs_is_exclusive(args):                    # main-cli.c:1279-1321
    rc, out, _ = run(*args)
    assert rc == 1 and "[ERROR]\tCombining [-a|--all] with other algorithm flags is not supported." in out


def test_iteration_count_validation():                   # main-cli.c:1323-1331, 1376-1381
    assert "[ERROR]\tInvalid number of algorithm iterations specified." in run("-n", "0", "x")[1]
    assert "[ERROR]\tInvalid number of algorithm iterations specified." in run("--number=-3", "x")[1]
    assert "[ERROR]\tArgument for iteration count contains non-number characters." in run("-n", "12x", "x")[1]
    assert "[ERROR]\tInvalid number of CISR slots specified." in run("-s", "0", "x")[1]
    assert "[ERROR]\tOne or more options missing a required argument." in run("-c", "-n")[1]


def test_report_dir_must_exist(tmp_path):                # main-cli.c:1344-1355
    rc, out, _ = run("-c", "-d", str(tmp_path / "nope"), "x.mtx")
    assert rc == 1 and "[ERROR]\tReport output folder not found." in out


def test_single_positional_file_required():              # main-cli.c:1389-1393
    for args in (["-c"], ["-c", "a.mtx", "b.mtx"]):
        rc, out, err = run(*args)
        assert rc == 1 and "Must specify a single input file" in err
    # POSIXMEHARDER: an option after the file is a second positional
    rc, out, err = run(ob.fixture_path("ibm32.mtx"), "-c")
    assert rc == 1 and "Must specify a single input file" in err


def test_missing_input_file():                           # main-cli.c:1394-1398
    rc, out, _ = run("-c", "/no/such/file.mtx")
    assert rc == 1 and "[ERROR]\tSpecified input file not found." in out


def test_unknown_option():
    rc, out, err = run("--frobnicate", "x")
    assert rc == 1 and "unknown option" in err


def test_help_and_usage():
    rc, out, _ = run("--help")
    assert rc == 0 and "--all-algs" in out and "Enable CSR SMVP algorithm." in out
    rc, out, _ = run("--usage")
    assert rc == 0 and out.startswith("Usage:")


def test_empty_file_is_premature_eof():                  # sample-data/badfile.mtx, main-cli.c:146-150
    rc, out, _ = run("-c", ob.fixture_path("badfile.mtx"))
    assert rc == 1 and "[START]\tExecuting smvp-toolbox-cli v0.6.4" in out
    assert "Required parameters not present on first line of file." in out


def test_dense_array_files_are_refused(tmp_path):        # main-cli.c:1410-1414
    p = tmp_path / "dense.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n1 1\n1.0\n")
    rc, out, _ = run("-c", str(p))
    assert rc == 1 and "only supports sparse matricies" in out


def test_header_errors(tmp_path):                        # main-cli.c:151-160
    p = tmp_path / "m.mtx"
    p.write_text("hello world this is text\n")
    assert "Required header is missing" in run("-c", str(p))[1]
    p.write_text("%%MatrixMarket matrix coordinate quaternion general\n1 1 1\n1 1 1\n")
    assert "Matrix content description not parseable" in run("-c", str(p))[1]


def test_stdout_tags_up_to_the_compute_step():
    """Without a GPU the CLI must stop at device selection with an error -- never compute on the CPU."""
    if sm.device_count() > 0:
        pytest.skip("a GPU is visible")
    rc, out, _ = run("-c", "-n", "3", ob.fixture_path("ibm32.mtx"))
    assert rc == 1
    for tag in ("[START]\tExecuting smvp-toolbox-cli v0.6.4", "[FILE]\tInput matrix file name: ",
                "[INFO]\tLoading matrix content from source file.",
                "[DATA]\tNon-zero numbers contained in matrix: ", "126",
                "Ones vector with dimensions [32, 1]", "[ERROR]"):
        assert tag in out
    assert "no HIP device" in out


def test_cisr_runs_without_a_gpu():                      # main-cli.c:1473-1476: host-only work (tests/test_cisr.py)
    rc, out, _ = run("-g", ob.fixture_path("ibm32.mtx"))
    assert rc == 0 and "memory_initialization_vector=" in out and "03ffffffff;" in out


def test_x_and_dump_arrays_flags(tmp_path):
    """--x ones|random and --dump-arrays are parsed (additive flags); without a GPU the run stops at device
    selection, after the operand line.  The random operand is a documented pure function of (seed, index)."""
    import numpy as np

    f = ob.fixture_path("ibm32.mtx")
    rc, out, _ = run("-c", "--x", "bogus", f)
    assert rc == 1 and "[ERROR]\tUnknown operand (use ones or random)." in out
    rc, out, _ = run("-c", "--x=random", "--dump-arrays", "-d", str(tmp_path), f)
    assert "Random vector (uniform [0, 1), seed 67890) with dimensions [32, 1]" in out
    rc, out, _ = run("-c", "--x", "ones", "-d", str(tmp_path), f)
    assert "Ones vector with dimensions [32, 1]" in out
    assert "--dump-arrays" in run("--help")[1] and "--x=ones" in run("--help")[1]
    # the generator: top 53 bits of splitmix64's finaliser of seed + i, times 2^-53
    x = sm.vector_random(1000, 67890)
    z = (np.uint64(67890) + np.arange(1000, dtype=np.uint64) + np.uint64(0x9e3779b97f4a7c15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    z = z ^ (z >> np.uint64(31))
    assert np.array_equal(x, (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53)
    assert 0.0 <= x.min() and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.05
