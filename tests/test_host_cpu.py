"""CPU-only checks of libpsa_hip: the library loads, exports every symbol that
include/psa.h declares, and its host-side helpers (no GPU work) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def psa():
    import psa_amd
    psa_amd.lib()
    return psa_amd


def test_exports_every_declared_symbol(psa):
    hdr = open(os.path.join(ROOT, "include", "psa.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(psa_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = psa.lib()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_result_struct_layout(psa):
    assert C.sizeof(psa.Result) == 96


def test_host_helpers_match_oracle(psa, oracle):
    for a in range(26):
        for b in range(26):
            assert psa.residue_class(chr(65 + a), chr(65 + b)) == oracle.lib().psa_oracle_class(65 + a, 65 + b)
    for w in ([2, 1.5, 1.1, 1.3], [2.5, 1.5, 1.1, 1.3], [0, 0, 0, 0]):
        w = np.array(w, dtype=np.float32)
        assert (psa.createCompMatrix(w).view(np.uint32) == oracle.comp_matrix(w).view(np.uint32)).all()
    for L2 in (2, 3, 5, 64, 138, 2000, 8000):
        M = L2 * (L2 - 1) // 2
        for t in sorted(set([1, 2, L2 - 1, L2, L2 + 1, M // 2, M - 1, M]) & set(range(1, M + 1))):
            assert psa.CPUGetNK(t, L2) == oracle.get_nk(t, L2)
            n, k = psa.CPUGetNK(t, L2)
            assert psa.mutant_number(n, k, L2) == t
    rng = np.random.default_rng(0)
    for _ in range(2000):
        L2 = int(rng.integers(2, 8192))
        t = int(rng.integers(1, L2 * (L2 - 1) // 2 + 1))
        assert psa.CPUGetNK(t, L2) == oracle.get_nk(t, L2)


def test_format_and_portion(psa):
    r = psa.Result()
    r.mutant_num, r.n, r.k, r.offset, r.score = 84, 1, 86, 364, 272.0
    assert psa.format_result(r) == "mutant num: 84, MS(1,86), score: 272.00, offset: 364\n"
    assert psa.calcPortion(7) == (3, 4) and psa.calcPortion(4) == (2, 2)


def test_partition_balances_cost(psa):
    rng = np.random.default_rng(1)
    len2 = rng.integers(500, 2001, size=4096).astype(np.int32)
    L1 = 10000
    for parts in (1, 2, 4, 8):
        for stages in (0, psa.STAGE_RESOLVE):
            owner = psa.partition(len2, L1, parts, stages)
            cost = np.array([psa.cost_estimate(int(L), L1, stages) for L in len2])
            loads = np.array([cost[owner == p].sum() for p in range(parts)])
            assert loads.min() > 0 and loads.max() / loads.mean() <= 1.02
    # the cost follows the candidate loop's tile count, not the raw candidate count: candidates per
    # predicted cycle rise with the Seq2 length (1024 candidates per 32 cycles at the bound of a tile)
    def eff(L2):
        cand = (L1 - L2 + 3) * (L2 * (L2 - 1) // 2)
        return cand / 1024.0 * 32.0 / psa.cost_estimate(L2, L1, 0)
    assert 0.45 < eff(512) < 0.56 and 0.63 < eff(2000) < 0.74 and eff(512) < eff(1000) < eff(2000) < eff(8000) < 0.85
    assert psa.cost_estimate(1, L1, 0) == 0.0 and psa.cost_estimate(L1 + 3, L1, 0) == 0.0
    assert psa.cost_estimate(700, L1, psa.STAGE_RESOLVE) == (L1 - 700 + 3) * 700.0


def test_bench_tile_count_matches_cost_model(psa):
    """The tile count bench.py prices the candidate loop with is the one the library's cost model uses."""
    import importlib.util, pathlib
    spec = importlib.util.spec_from_file_location("bench_mod", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    L1 = 9000
    for L2 in (2, 31, 32, 33, 64, 65, 255, 256, 257, 500, 1000, 2000, 4097, 8191):
        S = L1 - L2 + 3
        levels = (64 * ((L2 + 63) // 64) + 255) // 256
        tiles = (psa.cost_estimate(L2, L1, 0) / S - 1300.0 * levels) / 40.0
        assert tiles == bench.tiles_per_offset(L2), L2
        # every candidate pair (p < q) lies in a counted tile
        assert bench.tiles_per_offset(L2) * 1024 >= L2 * (L2 - 1) // 2


def test_combine_results_tie_order(psa):
    def rec(score, n, k, off, cand=10):
        r = psa.Result()
        r.score, r.n, r.k, r.offset, r.candidates = score, n, k, off, cand
        r.enum_frac_bits, r.enum_n, r.enum_k, r.enum_offset, r.enum_score = 3, n, k, off, int(score * 8)
        return r
    out = psa.combine_results([rec(5.0, 3, 4, 9), rec(7.5, 2, 9, 100), rec(7.5, 2, 9, 40), rec(7.5, 2, 10, 1)])
    assert (out.n, out.k, out.offset, out.score) == (2, 9, 40, 7.5)
    assert (out.enum_n, out.enum_k, out.enum_offset) == (2, 9, 40) and out.candidates == 40


def test_no_gpu_means_loud_failure(psa):
    """Without a GPU the product path must raise, never fall back to CPU code."""
    if psa.lib().psa_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(psa.PsaError):
        psa.Context(0)


def _mpiexec():
    for cand in ("/opt/conda/bin/mpiexec", "mpiexec"):
        from shutil import which
        if which(cand):
            return which(cand)
    return None


@pytest.mark.parametrize("np_ranks", [1, 2, 3])
def test_mpi_host_split_dry_run(psa, np_ranks, tmp_path):
    """psa_mpi without a GPU: rank 0 reads, all ranks agree on the cost-balanced split, rank 0 reports
    it in input order (the MPI plumbing of the multi-node host program, world sizes 1 to 3)."""
    import subprocess
    exe = os.path.join(ROOT, "parallel-sequence-alignment_amd", "psa_mpi")
    mpiexec = _mpiexec()
    if mpiexec is None or not os.path.exists(exe):
        pytest.skip("no MPI installation: psa_mpi is not built")
    inp = os.path.join(ROOT, "tests", "golden", "inputs", "ref_input.txt")
    r = subprocess.run([mpiexec, "-np", str(np_ranks), exe, "--dry-run", inp, str(tmp_path / "unused.txt")],
                       capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "my rank is 0 and i'm currently reading from the file"  # P/main.c:37
    lines = lines[1:]
    assert lines[0] == "num of seq2 = 4"  # P/main.c:51
    got = [re.match(r"Seq2 #(\d+) \((\d+) characters\) -> rank (\d+)", l).groups() for l in lines[1:5]]
    lens = [138, 1202, 466, 377]
    assert [int(g[0]) for g in got] == [1, 2, 3, 4] and [int(g[1]) for g in got] == lens
    owner = psa.partition(lens, 1489, np_ranks, psa.STAGE_RESOLVE)
    assert [int(g[2]) for g in got] == list(owner)
    assert lines[5].startswith("Total Time: ")
    assert not (tmp_path / "unused.txt").exists()


def test_mpi_host_more_ranks_than_seq2(psa, tmp_path):
    """Three ranks, two Seq2: the rank without work still takes part in every collective."""
    import subprocess
    exe = os.path.join(ROOT, "parallel-sequence-alignment_amd", "psa_mpi")
    mpiexec = _mpiexec()
    if mpiexec is None or not os.path.exists(exe):
        pytest.skip("no MPI installation: psa_mpi is not built")
    inp = os.path.join(ROOT, "tests", "golden", "inputs", "ref_p_input1.txt")
    r = subprocess.run([mpiexec, "-np", "3", exe, "--dry-run", inp], capture_output=True, text=True, timeout=60,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[1] == "num of seq2 = 2" and len([l for l in lines if l.startswith("Seq2 #")]) == 2
