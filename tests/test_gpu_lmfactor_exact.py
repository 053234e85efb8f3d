"""The exact policy's factorisation on its own (nlh_lmfactor_exact = nlh_qrx.hip: lmfactor + Q^T f of
src/nonlin_least_squares.f90:569-667 / :241-253 in the reference's operation order) against the CPU oracle, BIT FOR BIT,
on matrices built to reach the corners the random LM problems rarely visit: graded rows (a new running maximum inside
every NORM2 run), graded and permuted columns (non-trivial pivoting), duplicate columns (ties: lowest index wins), zero
and dependent columns (zero reflectors), exact zeros, sizes on both sides of the kernels' internal limits (more than 256
candidate columns, columns longer than one NORM2 chunk), and the same matrix in batches that select each of the three
forms of the trailing pass (the wide sixteen-wave and the four-wave row-parallel form, one wave per window)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _qtf_reference(a, rdiag, f):
    """Q^T f as lss_solve forms it (:241-253): ascending dot product, no fused operations (Python floats)."""
    m, n = a.shape
    w = [float(v) for v in f]
    qtf = np.zeros(n)
    for j in range(n):
        ajj = float(a[j, j])
        if ajj != 0.0:
            col = a[j:, j]
            s = 0.0
            for i in range(m - j):
                s = s + float(col[i]) * w[j + i]
            t = -s / ajj
            for i in range(m - j):
                w[j + i] = w[j + i] + float(col[i]) * t
        qtf[j] = w[j]
    return qtf, np.array(w)


def _matrix(kind, m, n, rng):
    a = rng.standard_normal((m, n))
    if kind == "random":
        pass
    elif kind == "graded_rows":              # |row i| grows: every element is a new maximum of the running NORM2
        a *= (1.0 + 1e-3) ** np.arange(m)[:, None]
    elif kind == "graded_rows_down":
        a *= (1.0 + 1e-3) ** (-np.arange(m))[:, None]
    elif kind == "graded_cols":              # column norms spread over 12 decades, shuffled
        a *= 10.0 ** (-12.0 * rng.permutation(n) / max(n - 1, 1))[None, :]
    elif kind == "duplicates":               # pairs of identical columns: ties in the pivot search, dependent columns
        a[:, 1::2] = a[:, 0:n - 1:2][:, : a[:, 1::2].shape[1]]
    elif kind == "zero_cols":
        a[:, rng.choice(n, size=max(1, n // 5), replace=False)] = 0.0
    elif kind == "sparse":                   # three quarters exact zeros, a few all-zero rows
        a[rng.random((m, n)) < 0.75] = 0.0
        a[rng.choice(m, size=m // 10, replace=False), :] = 0.0
    elif kind == "rank_one":
        a = np.outer(rng.standard_normal(m), rng.standard_normal(n))
    elif kind == "quantized":                # multiples of 1/64: the squared ratios of NORM2 land on exact ties of its running sum
        a = np.round(a * 64.0) / 64.0
    elif kind == "plus_minus_one":           # every ratio exactly one
        a = np.sign(a)
    elif kind == "heavy_tail":               # a few huge entries late in the column: new maxima far down, tiny ratios before
        a = a * (1.0 + 1e6 * (rng.random((m, n)) < 2e-4))
    else:
        raise ValueError(kind)
    return np.asfortranarray(a)


def _check(ds, oracle, a, f, copies=1):
    m, n = a.shape
    J = torch.tensor(np.ascontiguousarray(a.T)[None].repeat(copies, axis=0), device="cuda:0")
    F = torch.tensor(np.ascontiguousarray(f)[None].repeat(copies, axis=0), device="cuda:0")
    R, ipvt, rdiag, acnorm, qtf, wa4 = ds.lmfactor_exact(J, F)
    ao, ip, rd, acn = oracle.lmfactor(a)
    qt, w = _qtf_reference(ao, rd, f)
    for p in sorted({0, copies // 2, copies - 1}):
        assert np.array_equal(ipvt[p].cpu().numpy(), ip)
        assert np.array_equal(rdiag[p].cpu().numpy(), rd)
        assert np.array_equal(acnorm[p].cpu().numpy(), acn)
        Rg = R[p].cpu().numpy().T                                  # column-major n x n
        up = np.triu_indices(n, 1)
        assert np.array_equal(Rg[up], ao[:n, :n][up])
        assert np.array_equal(np.diag(Rg), rd)
        assert np.array_equal(qtf[p].cpu().numpy(), qt)
        assert np.array_equal(wa4[p].cpu().numpy(), w)
    if copies > 1:                                                 # every copy the same bits
        for t in (R, ipvt, rdiag, acnorm, qtf, wa4):
            assert bool((t == t[0:1]).all())


KINDS = ["random", "graded_rows", "graded_rows_down", "graded_cols", "duplicates", "zero_cols", "sparse", "rank_one"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("m,n", [(300, 37), (130, 129), (64, 64), (21, 4)])
def test_lmfactor_exact_bitwise_adversarial(ds, oracle, kind, m, n):
    rng = np.random.default_rng(hash((kind, m, n)) % (2 ** 31))
    a = _matrix(kind, m, n, rng)
    f = rng.standard_normal(m)
    _check(ds, oracle, a, f)


@pytest.mark.parametrize("kind", ["graded_rows", "duplicates", "graded_cols"])
@pytest.mark.parametrize("m,n", [(4500, 40), (700, 300)])
def test_lmfactor_exact_bitwise_beyond_internal_limits(ds, oracle, kind, m, n):
    """m - j > 4096: the pivot column does not fit one NORM2 chunk; n - j > 256: more candidate columns than threads."""
    rng = np.random.default_rng(7 + m + n)
    a = _matrix(kind, m, n, rng)
    f = rng.standard_normal(m)
    _check(ds, oracle, a, f)


@pytest.mark.parametrize("kind", ["random", "graded_rows", "graded_rows_down", "duplicates", "zero_cols", "sparse"])
@pytest.mark.parametrize("m,n,copies", [(9001, 24, 1), (12290, 17, 3), (4101, 30, 2), (8200, 12, 40), (2100, 33, 1), (4096, 70, 3)])
def test_lmfactor_exact_bitwise_long_columns(ds, oracle, kind, m, n, copies):
    """Columns of several NORM2 chunks.  A handful of problems: the workgroup-per-column sweep with the update one step
    behind (k_qrx_pass_col: one pending reflector, a bank switch per step, chain wave + preparing waves) and the
    pipelined NORM2 of the pivot kernel (graded rows: a new maximum in every run, the general recurrence in every chunk;
    graded down: the maximum is the first element; m not a multiple of 8, m - n on both sides of 4096; single-chunk columns
    of more than 2048 rows take the same sweep).  Forty problems:
    the lane-per-column passes with up to nine pending reflectors under the long-column pivot kernel."""
    rng = np.random.default_rng(11 + m + n)
    a = _matrix(kind, m, n, rng)
    f = rng.standard_normal(m)
    _check(ds, oracle, a, f, copies=copies)


@pytest.mark.parametrize("kind", ["random", "quantized", "plus_minus_one", "heavy_tail", "sparse", "graded_rows_down"])
@pytest.mark.parametrize("m,n", [(40000, 6), (70001, 4)])
def test_lmfactor_exact_long_columns_chain_free_norm2(ds, oracle, kind, m, n):
    """Columns of a dozen and more NORM2 chunks: from the third chunk on the running sum of squared ratios is formed
    WITHOUT the serial chain (ordered_possum_wave_int, nlh_common.h: per-binade exact additions with a two-state tie rule),
    falling back to the chain where the sum crosses a binade or a new maximum appears.  Quantized data puts the ratios on
    exact ties, +-1 makes every ratio one, heavy tails move the maximum late; every bit must be the oracle's."""
    rng = np.random.default_rng(101 + m + n)
    a = _matrix(kind, m, n, rng)
    f = rng.standard_normal(m)
    _check(ds, oracle, a, f)


@pytest.mark.parametrize("copies", [1, 20, 40, 60, 100, 120, 200, 300, 1100])
def test_lmfactor_exact_every_pass_form(ds, oracle, copies):
    """The same graded 520 x 70 matrix (two 64-column windows) in batches of 1 / 20 (<= 1536 (problem, column) pairs: the
    workgroup-per-column sweep), 40, 60, 100 and 120 (80 / 120 / 200 / 240 (problem, window) pairs: the wide row-parallel form, sixteen
    waves, producers reading whole sectors per lane quad, on 32-column half windows while those are at most 256 -- 60 copies
    all the way, 100 once one window is left --; at most 256 pairs), 200 (400: four-wave; at most 512), 300 and 1100
    (one wave per window, separate and as the waves of one workgroup); all the same bits."""
    rng = np.random.default_rng(99)
    a = _matrix("graded_rows", 520, 70, rng)
    a[:, 5] = a[:, 3]
    a[:, 11] = 0.0
    f = rng.standard_normal(520)
    _check(ds, oracle, a, f, copies=copies)
