"""GPU parity tests, one kernel family at a time, against the CPU oracle on the same seeded
inputs.  Everything goes through the C ABI (nonlin_amd.device -> libnonlin_hip.so)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (2048, 128), (64, 64)])
def test_generator_matches_oracle_bitwise(ds, oracle, m, n):
    nprob = 3
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=777, gamma=0.5, sigma=1e-3, spread=0.3,
                               square_shift=(m == n))
    for p in range(nprob):
        Ah, bh, xth, x0h = oracle.dq_generate(777 + p, m, n, square_shift=(m == n))
        assert np.array_equal(A[p].cpu().numpy().T, Ah)
        assert np.array_equal(xt[p].cpu().numpy(), xth)
        assert np.array_equal(x0[p].cpu().numpy(), x0h)
        assert np.array_equal(b[p].cpu().numpy(), bh)      # needs the bit-exact residual kernel


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (4096, 256), (1, 1)])
def test_residual_bit_exact(ds, oracle, m, n):
    """vecfcn of the device model: every residual equals the CPU value bit for bit."""
    nprob = 2
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=4242)
    f = ds.residual(A, b, 0.5, x0)
    for p in range(nprob):
        fo = oracle.dq_residual(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5,
                                x0[p].cpu().numpy())
        assert np.array_equal(f[p].cpu().numpy(), fo)


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (2048, 128), (130, 130)])
def test_fd_jacobian_bit_exact(ds, oracle, m, n):
    """vfh_jac_fcn: the n perturbed evaluations and the (f1 - f0)/h column write, bit for bit;
    includes an exactly-zero x_j (h = sqrt(eps) branch)."""
    nprob = 2
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=99)
    x0[:, 0] = 0.0
    f0 = ds.residual(A, b, 0.5, x0)
    P = ds.fd_panel(A, b, 0.5, x0)
    J = ds.fd_jacobian_panel(P, f0, x0)
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        Jo = oracle.dq_fd_jacobian(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), fv=f0[p].cpu().numpy())
        assert np.array_equal(J[p].cpu().numpy().T, Jo)


def test_analytic_jacobian_bit_exact(ds, oracle):
    A, b, xt, x0 = ds.generate(2, 96, 96, seed0=5, square_shift=True)
    J = ds.jacobian(A, 0.5, x0)
    for p in range(2):
        Jo = oracle.dq_jacobian(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy())
        assert np.array_equal(J[p].cpu().numpy().T, Jo)


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (4096, 256), (1000, 130)])
def test_gram_mfma(ds, m, n):
    """J^T J and J^T f (fp64 MFMA, split-K) against a float64 reference; exact symmetry and
    run-to-run reproducibility.  Asymmetric random J catches a transposed C/D map."""
    g = torch.Generator(device="cpu").manual_seed(1)
    J = torch.randn((2, n, m), dtype=torch.float64, generator=g).cuda()
    f = torch.randn((2, m), dtype=torch.float64, generator=g).cuda()
    G, gv = ds.gram(J, f)
    G2, gv2 = ds.gram(J, f)
    Gref = torch.matmul(J, J.transpose(1, 2))          # [p, a, b] = sum_i J(i,a) J(i,b)
    gref = torch.matmul(J, f.unsqueeze(-1)).squeeze(-1)
    scale = float(Gref.abs().max())
    assert float((G - Gref).abs().max()) <= 1e-13 * scale * np.sqrt(m)
    assert float((gv - gref).abs().max()) <= 1e-13 * float(gref.abs().max()) * np.sqrt(m)
    assert torch.equal(G, G.transpose(1, 2))
    assert torch.equal(G, G2) and torch.equal(gv, gv2)


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (2048, 128)])
def test_chol_factor_matches_lmfactor(ds, oracle, m, n):
    """Pivoted Cholesky of J^T J reproduces lmfactor's pivot order, |R|, acnorm and |qtf|
    (rows of R and entries of qtf are defined up to a common sign)."""
    A, b, xt, x0 = ds.generate(1, m, n, seed0=31)
    f0 = ds.residual(A, b, 0.5, x0)
    J = ds.fd_jacobian_panel(ds.fd_panel(A, b, 0.5, x0), f0, x0)
    G, g = ds.gram(J, f0)
    ipvt, acnorm, qtf, info = ds.chol_factor(G, g)
    assert int(info[0]) == 0
    Jh = np.asfortranarray(J[0].cpu().numpy().T)
    a, ip, rdiag, acn = oracle.lmfactor(Jh)
    assert np.array_equal(ipvt[0].cpu().numpy(), ip)
    np.testing.assert_allclose(acnorm[0].cpu().numpy(), acn, rtol=1e-13)
    R = np.triu(G[0].cpu().numpy().T)                   # G[p, c, r] -> R(r, c)
    Ro = np.triu(a[:n, :n], 1) + np.diag(rdiag)
    sgn = np.sign(rdiag)
    np.testing.assert_allclose(R, sgn[:, None] * Ro, rtol=0, atol=1e-11 * np.abs(Ro).max())
    # qtf: apply the reflectors to f on the host
    w = f0[0].cpu().numpy().copy()
    a2 = a.copy()
    for j in range(n):
        if a2[j, j] != 0.0:
            t = -np.dot(a2[j:, j], w[j:]) / a2[j, j]
            w[j:] += a2[j:, j] * t
    np.testing.assert_allclose(qtf[0].cpu().numpy(), sgn * w[:n], rtol=0, atol=1e-11 * np.abs(w).max())


@pytest.mark.parametrize("m,n", [(21, 4), (512, 64), (300, 37), (64, 64)])
def test_qr_factor_matches_lmfactor(ds, oracle, m, n):
    """The faithful lmfactor + Q^T f kernel against the oracle (same algorithm; only the
    reduction order of norms and dot products differs)."""
    A, b, xt, x0 = ds.generate(1, m, n, seed0=57, square_shift=(m == n))
    f0 = ds.residual(A, b, 0.5, x0)
    J = ds.fd_jacobian_panel(ds.fd_panel(A, b, 0.5, x0), f0, x0)
    Jh = np.asfortranarray(J[0].cpu().numpy().T)
    ipvt, rdiag, acnorm, qtf, wa4 = ds.qr_factor(J, f0)
    a, ip, rd, acn = oracle.lmfactor(Jh)
    assert np.array_equal(ipvt[0].cpu().numpy(), ip)
    np.testing.assert_allclose(rdiag[0].cpu().numpy(), rd, rtol=1e-12)
    np.testing.assert_allclose(acnorm[0].cpu().numpy(), acn, rtol=1e-13)
    Jg = J[0].cpu().numpy().T
    sc = np.abs(a).max()
    # below/above the diagonal the factored matrices agree; the diagonal was reset to rdiag (:251)
    mask = ~np.eye(m, n, dtype=bool)
    np.testing.assert_allclose(Jg[mask], a[mask], rtol=0, atol=1e-12 * sc)
    w = f0[0].cpu().numpy().copy()
    for j in range(n):
        if a[j, j] != 0.0:
            t = -np.dot(a[j:, j], w[j:]) / a[j, j]
            w[j:] += a[j:, j] * t
    np.testing.assert_allclose(qtf[0].cpu().numpy(), w[:n], rtol=0, atol=1e-12 * np.abs(w).max())
    np.testing.assert_allclose(wa4[0].cpu().numpy(), w, rtol=0, atol=1e-12 * np.abs(w).max())


@pytest.mark.parametrize("m,n,delta_scale", [(64, 16, 0.05), (512, 64, 0.3), (300, 37, 0.3), (64, 64, 0.02)])
def test_lmpar_binding_trust_region(ds, oracle, m, n, delta_scale):
    """lmpar with a binding trust region (forces the secular-equation loop, lmsolve and both
    deviations from MINPACK) on real QR factors, against the oracle."""
    A, b, xt, x0 = ds.generate(1, m, n, seed0=11, gamma=2.0, sigma=0.1, spread=2.0, square_shift=(m == n))
    f0 = ds.residual(A, b, 2.0, x0)
    J = ds.fd_jacobian_panel(ds.fd_panel(A, b, 2.0, x0), f0, x0)
    Jh = np.asfortranarray(J[0].cpu().numpy().T)
    a, ip, rd, acn = oracle.lmfactor(Jh)
    w = f0[0].cpu().numpy().copy()
    for j in range(n):
        if a[j, j] != 0.0:
            t = -np.dot(a[j:, j], w[j:]) / a[j, j]
            w[j:] += a[j:, j] * t
        a[j, j] = rd[j]
    qtf = w[:n].copy()
    diag = np.where(acn == 0.0, 1.0, acn)
    delta = delta_scale * np.linalg.norm(diag * x0[0].cpu().numpy())
    par_o, x_o, sdiag_o, _ = oracle.lmpar(a, ip, diag, qtf, delta, 0.0, w)
    assert par_o > 0.0            # the loop really ran
    R = torch.tensor(np.ascontiguousarray(a[:n, :n].T), device="cuda").unsqueeze(0)    # [1, n(col), n(row)]
    dev = "cuda"
    f64 = dict(dtype=torch.float64, device=dev)
    par, x, sdiag = ds.lmpar(R, torch.tensor(ip, dtype=torch.int32, device=dev).unsqueeze(0),
                             torch.tensor(diag, **f64).unsqueeze(0), torch.tensor(qtf, **f64).unsqueeze(0),
                             torch.tensor([float(delta)], **f64), torch.tensor([float(np.sum(w[n:] ** 2))], **f64),
                             torch.tensor([0.0], **f64))
    assert abs(float(par[0]) - par_o) <= 1e-10 * par_o
    np.testing.assert_allclose(x[0].cpu().numpy(), x_o, rtol=0, atol=1e-10 * np.abs(x_o).max())
    np.testing.assert_allclose(sdiag[0].cpu().numpy(), sdiag_o, rtol=1e-10)


@pytest.mark.parametrize("n", [2, 3, 17, 63, 64, 65, 95, 96, 127, 128, 129, 191, 200, 255, 256, 257])
@pytest.mark.parametrize("zero_diag", [False, True])
def test_lmsolve_on_chip_sweep_is_bitwise_the_global_wavefront(ds, n, zero_diag):
    """lmsolve's Givens sweeps (src/nonlin_least_squares.f90:717-765) for n <= 256 run out of LDS and registers (columns of S
    in a ring, working rows in the owning wave's registers: nlh_kernels_lm.h); NLH_LMSOLVE_GLOBAL=1 forces the global-memory
    wavefront that serves larger n.  Same rotations on the same data in the same order: par, x, sdiag and the S left in
    R's lower triangle must agree bit for bit -- including eliminations skipped because diag(l) == 0 (:721) and
    rotations skipped because the entry is already zero (:732).  (n = 257: both runs take the global form.)"""
    rng = np.random.default_rng(1000 + n)
    R = np.triu(rng.standard_normal((n, n)))
    R[np.arange(n), np.arange(n)] += np.sign(R[np.arange(n), np.arange(n)]) * 2.0
    if zero_diag and n > 4:
        R[2, 3:] = 0.0                                   # a zero entry for rotations to meet
    ip = rng.permutation(n).astype(np.int32)
    diag = np.abs(rng.standard_normal(n)) + 0.5
    if zero_diag:
        diag[rng.integers(0, n, size=max(1, n // 8))] = 0.0
    qtf = rng.standard_normal(n)
    xgn = np.empty(n)
    xgn[ip] = np.linalg.solve(R, qtf)                    # the Gauss-Newton step, x(ipvt(j)) = z(j)
    delta = 0.2 * np.linalg.norm(diag * xgn) + 1e-3
    f64 = dict(dtype=torch.float64, device="cuda")
    outs = []
    for env in ("1", "0"):
        os.environ["NLH_LMSOLVE_GLOBAL"] = env
        try:
            Rd = torch.tensor(np.ascontiguousarray(R.T), device="cuda").unsqueeze(0).repeat(3, 1, 1)
            par, x, sdiag = ds.lmpar(Rd, torch.tensor(ip, dtype=torch.int32, device="cuda").unsqueeze(0).repeat(3, 1),
                                     torch.tensor(diag, **f64).unsqueeze(0).repeat(3, 1), torch.tensor(qtf, **f64).unsqueeze(0).repeat(3, 1),
                                     torch.tensor([float(delta)] * 3, **f64), torch.tensor([0.0] * 3, **f64), torch.tensor([0.0] * 3, **f64))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("NLH_LMSOLVE_GLOBAL", None)
        outs.append((par.cpu().numpy(), x.cpu().numpy(), sdiag.cpu().numpy(), Rd.cpu().numpy()))
    assert outs[0][0][0] > 0.0                            # lmpar's iteration (and with it lmsolve) really ran
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    for q in range(1, 3):                                 # the problems of a launch are independent and identical
        assert np.array_equal(outs[1][3][q], outs[1][3][0]) and np.array_equal(outs[1][1][q], outs[1][1][0])


def test_lmsolve_on_chip_sweep_every_size_up_to_256(ds):
    """Every n from 2 to 256 (ring size, chunk count, register moves at the 64-element boundaries, thread count and the
    number of forming waves all depend on it), one problem each, every third with zero diagonal entries: on-chip form ==
    global-memory wavefront, bit for bit (par, x, sdiag, S)."""
    f64 = dict(dtype=torch.float64, device="cuda")
    for n in range(2, 257):
        rng = np.random.default_rng(5000 + n)
        R = np.triu(rng.standard_normal((n, n)))
        R[np.arange(n), np.arange(n)] += np.sign(R[np.arange(n), np.arange(n)]) * 2.0
        ip = rng.permutation(n).astype(np.int32)
        diag = np.abs(rng.standard_normal(n)) + 0.5
        if n % 3 == 0:
            diag[rng.integers(0, n, size=max(1, n // 8))] = 0.0
        qtf = rng.standard_normal(n)
        xgn = np.empty(n)
        xgn[ip] = np.linalg.solve(R, qtf)
        delta = 0.2 * np.linalg.norm(diag * xgn) + 1e-3
        outs = []
        for env in ("1", "0"):
            os.environ["NLH_LMSOLVE_GLOBAL"] = env
            try:
                Rd = torch.tensor(np.ascontiguousarray(R.T), device="cuda").unsqueeze(0)
                par, x, sdiag = ds.lmpar(Rd, torch.tensor(ip, dtype=torch.int32, device="cuda").unsqueeze(0), torch.tensor(diag, **f64).unsqueeze(0),
                                         torch.tensor(qtf, **f64).unsqueeze(0), torch.tensor([float(delta)], **f64), torch.tensor([0.0], **f64),
                                         torch.tensor([0.0], **f64))
                torch.cuda.synchronize()
            finally:
                os.environ.pop("NLH_LMSOLVE_GLOBAL", None)
            outs.append((par.cpu().numpy(), x.cpu().numpy(), sdiag.cpu().numpy(), Rd.cpu().numpy()))
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), n


@pytest.mark.parametrize("n", [2, 37, 130, 300, 600, 1024, 1100])
def test_lu_bit_exact(ds, oracle, n):
    """lu_factor / solve_lu stand-ins: same pivots, bit-identical factors and solution (n >= 128: the blocked path --
    register panels with 1, 2 and 4 rows per thread by panel height, the global-memory panel above 1024 rows)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    Ah = np.asfortranarray(rng.standard_normal((n, n)))
    bh = rng.standard_normal(n)
    lu = Ah.copy(order="F")
    ipo = np.zeros(n, dtype=np.int32)
    L = oracle.lib()
    L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    xo = bh.copy()
    L.nlo_lu_solve(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)),
                   xo.ctypes.data_as(C.POINTER(C.c_double)))
    Ad = torch.tensor(np.ascontiguousarray(Ah.T), device="cuda").unsqueeze(0)
    bd = torch.tensor(bh, device="cuda").unsqueeze(0)
    ipvt, info = ds.lu_factor(Ad)
    ds.lu_solve(Ad, ipvt, bd)
    assert int(info[0]) == 0
    assert np.array_equal(ipvt[0].cpu().numpy(), ipo)
    assert np.array_equal(Ad[0].cpu().numpy().T, lu)
    assert np.array_equal(bd[0].cpu().numpy(), xo)


def test_lu_batch_of_problems_bit_exact(ds, oracle):
    """Several problems in one call (the lock-step Newton batches factor this way): every problem's factors are the
    CPU loop's."""
    import ctypes as C
    n, nprob = 200, 5
    rng = np.random.default_rng(11)
    Ah = rng.standard_normal((nprob, n, n))
    L = oracle.lib()
    Ad = torch.tensor(np.ascontiguousarray(np.transpose(Ah, (0, 2, 1))), device="cuda")
    ipvt, info = ds.lu_factor(Ad)
    for p in range(nprob):
        lu = np.asfortranarray(Ah[p])
        ipo = np.zeros(n, dtype=np.int32)
        L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
        assert int(info[p]) == 0
        assert np.array_equal(ipvt[p].cpu().numpy(), ipo)
        assert np.array_equal(Ad[p].cpu().numpy().T, lu)


def _lapack_lu_names():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lapack_vectors.npz"))
    return [str(v) for v in g["lu_names"]]


@pytest.mark.parametrize("name", _lapack_lu_names())
def test_lu_follows_lapack_dgetrf(ds, name):
    """The device factorisation against LAPACK itself (tests/golden/lapack_vectors.npz: scipy's DGETRF on seeded
    matrices): the interchange sequence exactly, the factors to max(64, 4 n) ulp of the column scale.  linalg's lu_factor IS
    DGETRF (call site src/nonlin_solve.f90:570), whose bits depend on the LAPACK build: this is the tightest pin the
    reference's own arithmetic allows."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lapack_vectors.npz"))
    a = g[f"lu_{name}_a"]
    n = a.shape[0]
    ref, piv, rinfo = g[f"lu_{name}_lu"], g[f"lu_{name}_piv"], int(g[f"lu_{name}_info"])
    Ad = torch.tensor(np.ascontiguousarray(a.T), device="cuda").unsqueeze(0)
    ipvt, info = ds.lu_factor(Ad)
    lu = Ad[0].cpu().numpy().T
    ip = ipvt[0].cpu().numpy()
    assert int(info[0]) == rinfo
    eps = np.finfo(float).eps
    if name == "singular12":                       # equal columns: pivots agree up to the dependent column
        assert np.array_equal(ip[:9], piv[:9])
        assert np.abs(lu[:9, :] - ref[:9, :]).max() <= 64 * eps * np.abs(ref).max()
        return
    assert np.array_equal(ip, piv)
    scale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), np.abs(a).max(axis=0, keepdims=True))
    scale[scale == 0.0] = 1.0                      # (a zero column stays exactly zero)
    assert (np.abs(lu - ref) / scale).max() <= max(64, 4 * n) * eps


@pytest.mark.parametrize("n", [40, 130, 300, 700])
def test_lu_singular_and_tied_pivots(ds, oracle, n):
    """A zero pivot column (info = its 1-based index, the update still runs) and exact ties in the pivot search
    (the first maximum wins): the factors stay bit-identical to the CPU loop."""
    import ctypes as C
    rng = np.random.default_rng(n)
    Ah = np.asfortranarray(rng.integers(-3, 4, size=(n, n)).astype(np.float64))     # many ties
    Ah[:, 5] = 0.0                                                                    # structurally singular
    lu = Ah.copy(order="F")
    ipo = np.zeros(n, dtype=np.int32)
    L = oracle.lib()
    rc = L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    Ad = torch.tensor(np.ascontiguousarray(Ah.T), device="cuda").unsqueeze(0)
    ipvt, info = ds.lu_factor(Ad)
    assert int(info[0]) == rc and rc > 0
    assert np.array_equal(ipvt[0].cpu().numpy(), ipo)
    assert np.array_equal(Ad[0].cpu().numpy().T, lu, equal_nan=True)


@pytest.mark.parametrize("n", [40, 130, 300, 700, 1100])
@pytest.mark.parametrize("kind", ["all_nan", "nan_column", "nan_diagonal", "scattered"])
def test_lu_nan_entries_keep_a_valid_pivot(ds, oracle, n, kind):
    """NaN entries (a NaN Jacobian from a host callback, a NaN start point): the ordered pivot search of the CPU loop
    keeps a NaN diagonal entry and never selects a NaN below it.  Every pivot index must stay inside the matrix (the
    interchange and solve kernels index with it) and equal the CPU loop's."""
    import ctypes as C
    rng = np.random.default_rng(n)
    Ah = np.asfortranarray(rng.standard_normal((n, n)))
    if kind == "all_nan":
        Ah[:] = np.nan
    elif kind == "nan_column":
        Ah[:, n // 3] = np.nan
    elif kind == "nan_diagonal":
        Ah[n // 2, n // 2] = np.nan
    else:
        Ah[rng.random((n, n)) < 0.02] = np.nan
    lu = Ah.copy(order="F")
    ipo = np.zeros(n, dtype=np.int32)
    L = oracle.lib()
    L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    Ad = torch.tensor(np.ascontiguousarray(Ah.T), device="cuda").unsqueeze(0)
    bd = torch.tensor(rng.standard_normal(n), device="cuda").unsqueeze(0)
    ipvt, info = ds.lu_factor(Ad)
    ds.lu_solve(Ad, ipvt, bd)
    torch.cuda.synchronize()
    ip = ipvt[0].cpu().numpy()
    assert ip.min() >= 0 and ip.max() < n
    assert np.array_equal(ip, ipo)                # (round 4: scattered NaNs too -- every search form maps a NaN below the diagonal to "never")


_LU_LOOKAHEAD = '''
import ctypes as C, numpy as np, torch
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
ds = DeviceSolver(0)
L = O.lib()
for n in (256, 300, 513, 700, 1024):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    if n == 300: a = rng.integers(-3, 4, size=(n, n)).astype(float)          # ties in every pivot search
    if n == 513: a[:, 7] = 0.0                                                # a zero column
    lu = np.array(a, order="F"); ipo = np.zeros(n, dtype=np.int32)
    L.nlo_lu_factor(n, lu.ctypes.data_as(C.POINTER(C.c_double)), n, ipo.ctypes.data_as(C.POINTER(C.c_int32)))
    Ad = torch.tensor(np.ascontiguousarray(a.T), device="cuda").reshape(1, n, n)
    ipvt, info = ds.lu_factor(Ad)
    assert np.array_equal(Ad[0].cpu().numpy().T, lu, equal_nan=True), n
    assert np.array_equal(ipvt[0].cpu().numpy(), ipo), n
print("ok")
'''


@pytest.mark.gpu
def test_lu_look_ahead_form_is_bitwise():
    """NLH_LU_LOOKAHEAD=1 (off by default: measured slower): every panel applies the previous panel to its own columns inside
    its kernel and the rest of the update runs on a side stream under it -- factors and interchanges bit for bit the
    oracle's, across the panel-width switches, with ties and a zero column."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NLH_LU_LOOKAHEAD="1")
    out = subprocess.run([sys.executable, "-c", _LU_LOOKAHEAD], capture_output=True, text=True, timeout=600, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
