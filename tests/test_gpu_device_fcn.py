"""The open device-residual path: a USER'S vecfcn / jacobianfcn handed in as launchers (include/nonlin_hip.h:
nlh_device_vecfcn, nlh_device_jacfcn; reference plugin layer src/nonlin_multi_eqn_mult_var.f90:14-38, 126-153, 198-277).

Residual families written outside the library (tests/device_model/user_models.hip -> libuser_models.so) are solved on the
GPU through nlh_lm_solve_batch_device / nlh_newton_solve_batch_device / nlh_quasi_newton_solve_batch_device and must be
bit-identical -- x, fvec, every count and flag -- to the CPU oracle driving the SAME arithmetic as a host callback; and
the built-in dense-quadratic family expressed through the same launchers must reproduce the bits of its own entry points.
"""
import ctypes as C

import numpy as np
import pytest

import user_models as UM

dp = C.POINTER(C.c_double)
KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")


# ---------------------------------------------------------------------------------------------------- CPU side
def test_user_families_host_twins():
    """(CPU) the host twins of the user's kernels compute what they say, with IEEE operations in the stated order."""
    so = UM.lib()
    t, y, xt, x0 = UM.lorentz_problems(2, 97, 3, seed=5)
    f = np.zeros(97)
    ctx = UM.LorentzHost(97, t[1].ctypes.data_as(dp), y[1].ctypes.data_as(dp), 0)
    so.lorentz_host_fcn(C.byref(ctx), 9, x0[1].ctypes.data_as(dp), 97, f.ctypes.data_as(dp))
    assert ctx.ncalls == 1
    assert np.array_equal(f, UM.lorentz_row_numpy(x0[1], t[1], y[1]))
    n = 11
    x = np.linspace(-1.2, 0.7, n)
    fb, jb = np.zeros(n), np.zeros((n, n))
    bc = UM.BtriHost(1.25, 0, 0)
    so.btri_host_fcn(C.byref(bc), n, x.ctypes.data_as(dp), n, fb.ctypes.data_as(dp))
    so.btri_host_jac(C.byref(bc), n, x.ctypes.data_as(dp), n, jb.ctypes.data_as(dp))
    xm = np.concatenate(([0.0], x[:-1])); xp = np.concatenate((x[1:], [0.0]))
    assert np.array_equal(fb, (((3.0 - 2.0 * x) * x - xm) - 2.0 * xp) + 1.25)
    J = jb.T                                                  # column-major
    assert np.array_equal(np.diag(J), 3.0 - 4.0 * x)
    assert np.all(np.diag(J, -1) == -1.0) and np.all(np.diag(J, 1) == -2.0)
    assert np.count_nonzero(J) == 3 * n - 2


def test_device_fcn_symbols_and_errors_without_gpu():
    """(CPU) the launcher entry points exist, and refuse a missing function the way the reference does (:188)."""
    from nonlin_amd import _lib
    lib = _lib.load()
    for name in ("nlh_lm_solve_batch_device", "nlh_newton_solve_batch_device", "nlh_quasi_newton_solve_batch_device",
                 "nlh_fd_jacobian_device", "nlh_lm_solve_batch_device_h", "nlh_newton_solve_batch_device_h",
                 "nlh_quasi_newton_solve_batch_device_h", "nlh_dq_device_fcn", "nlh_dq_device_jac"):
        assert hasattr(lib, name)
    o = _lib.default_options()
    null = C.cast(None, _lib.DEVFCN)
    assert lib.nlh_lm_solve_batch_device(None, C.byref(o), 1, 4, 2, null, null, None, None, None, None, None) == -3   # bad handle


# ---------------------------------------------------------------------------------------------------- helpers
def _oracle_lm(oracle, host_fcn, hctx, m, n, x0, **okw):
    oo = oracle.default_options(**okw)
    xo, fo, ibo = x0.copy(), np.zeros(m), oracle.IterationBehavior()
    rc = oracle.lib().nlo_lm_solve(C.byref(oo), C.cast(host_fcn, oracle.VECFCN), C.cast(None, oracle.JACFCN), C.byref(hctx), m, n,
                                   xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    return rc, xo, fo, ibo.as_dict()


def _same(a, b):
    return all(a[k] == b[k] for k in KEYS)


def _check_lorentz(ds, oracle, nprob, m, K, sample=None, seed=2024, hard_every=0, opt=None, **gen):
    import torch
    opt = dict(max_evals=500, **(opt or {}))
    t, y, xt, x0 = UM.lorentz_problems(nprob, m, K, seed=seed, hard_every=hard_every, **gen)
    batch = UM.LorentzBatch(t, y)
    n = 3 * K
    x = torch.tensor(x0, device=ds.device)
    fvec, ibs, status = ds.lm_solve_batch_device(batch.launch, batch.ctx, m, x, opts=ds.options(**opt))
    xg, fg = x.cpu().numpy(), fvec.cpu().numpy()
    iters = set()
    for p in (range(nprob) if sample is None else sample):
        hc = batch.host_ctx(p)
        rc, xo, fo, ibo = _oracle_lm(oracle, batch.host_fcn, hc, m, n, x0[p], **opt)
        assert status[p] == rc, (p, status[p], rc)
        assert _same(ibs[p], ibo), (p, ibs[p], ibo)
        assert np.array_equal(xg[p], xo), (p, np.abs(xg[p] - xo).max())
        assert np.array_equal(fg[p], fo), p
        iters.add(ibo["jacobian_count"])
    batch.close()
    return iters


# ---------------------------------------------------------------------------------------------------- GPU: LM
@pytest.mark.gpu
@pytest.mark.parametrize("m,K", [(512, 4), (301, 2), (1000, 7)])
def test_user_family_single_problem_bitwise(ds, oracle, m, K):
    """One problem of the user's family: least_squares_solver%solve through the launcher == the oracle driving the host twin."""
    _check_lorentz(ds, oracle, 1, m, K)


@pytest.mark.gpu
def test_user_family_batch_bitwise(ds, oracle):
    """A batch of 300 spectra, every fourth started four times further away: problems finish in different rounds (the
    launcher is asked for the ones still iterating only), sub-batches on private streams; every problem bitwise."""
    iters = _check_lorentz(ds, oracle, 300, 512, 4, hard_every=4)
    assert len(iters) >= 2, iters                              # a heterogeneous batch indeed


@pytest.mark.gpu
def test_user_family_trust_region_binding(ds, oracle):
    """factor = 0.1 forces lmpar's iteration and rejected trial points: the inner-loop repeats go
    through the launcher with only the rejected problems active."""
    _check_lorentz(ds, oracle, 24, 400, 3, hard_every=3, opt=dict(factor=0.1), spread=0.1)


@pytest.mark.gpu
def test_user_family_sub_batches_and_one_batch_agree(ds, oracle):
    import torch
    t, y, xt, x0 = UM.lorentz_problems(260, 256, 2, seed=11, hard_every=5)
    batch = UM.LorentzBatch(t, y)
    out = []
    for sb in (1, 2, 3):
        x = torch.tensor(x0, device=ds.device)
        fvec, ibs, status = ds.lm_solve_batch_device(batch.launch, batch.ctx, 256, x, opts=ds.options(max_evals=500, sub_batches=sb))
        out.append((x.cpu().numpy(), fvec.cpu().numpy(), ibs, status))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and o[2] == out[0][2] and o[3] == out[0][3]
    batch.close()


@pytest.mark.gpu
def test_user_family_host_array_entry_point(ds, oracle):
    """nlh_lm_solve_batch_device_h (what the Fortran shim calls): host arrays in and out, same bits."""
    from nonlin_amd import _lib
    nprob, m, K = 5, 200, 2
    n = 3 * K
    t, y, xt, x0 = UM.lorentz_problems(nprob, m, K, seed=3)
    batch = UM.LorentzBatch(t, y)
    x, f = x0.copy(), np.zeros((nprob, m))
    ib = (_lib.IterationBehavior * nprob)()
    st = (C.c_int32 * nprob)()
    o = ds.options(max_evals=500)
    rc = ds.lib.nlh_lm_solve_batch_device_h(ds.h.ptr, C.byref(o), nprob, m, n, C.cast(batch.launch, _lib.DEVFCN),
                                            C.cast(None, _lib.DEVFCN), batch.ctx, x.ctypes.data_as(dp), f.ctypes.data_as(dp), ib, st)
    assert rc == 0
    for p in range(nprob):
        hc = batch.host_ctx(p)
        rco, xo, fo, ibo = _oracle_lm(oracle, batch.host_fcn, hc, m, n, x0[p], max_evals=500)
        assert st[p] == rco and _same(ib[p].as_dict(), ibo)
        assert np.array_equal(x[p], xo) and np.array_equal(f[p], fo)
    batch.close()


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [2, 0, 1])
@pytest.mark.parametrize("nprob,m,n,gen,opt", [
    (6, 512, 64, {}, {}),
    (3, 301, 37, dict(gamma=2.0, sigma=0.1, spread=5.0), dict(factor=0.1)),     # odd m, lmpar's loop, rejected trials
    (40, 256, 32, dict(gamma=10.0, sigma=1.0, spread=50.0), dict(factor=0.1)),  # 21/20-iteration family, stragglers
    (2, 2048, 128, {}, {}),
])
def test_dense_quadratic_family_through_the_launcher_reproduces_its_bits(ds, nprob, m, n, gen, opt, policy):
    """(ii) of the open-path contract: nlh_dq_device_fcn + nlh_lm_solve_batch_device == nlh_dq_lm_solve_batch, bit for bit,
    under every factor policy (the Jacobian the forward differences leave is the same matrix)."""
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345, **gen)
    g = gen.get("gamma", 0.5)
    o = ds.options(max_evals=500, factor_policy=policy, **opt)
    x1 = x0.clone()
    f1, ib1, st1 = ds.lm_solve_batch(A, b, g, x1, o)
    fcn, jac, ctx = ds.dq_launchers(A, b, g)
    x2 = x0.clone()
    f2, ib2, st2 = ds.lm_solve_batch_device(fcn, ctx, m, x2, opts=o)
    assert st1 == st2 and ib1 == ib2
    assert np.array_equal(x1.cpu().numpy(), x2.cpu().numpy())
    assert np.array_equal(f1.cpu().numpy(), f2.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, 1])
@pytest.mark.parametrize("nprob,m,n,hard", [(20, 488, 5, False), (3, 132, 102, True), (20, 138, 101, True), (7, 386, 36, True)])
def test_launcher_sum_of_squares_is_the_built_in_familys(ds, nprob, m, n, hard, policy):
    """The normal-equations policies read ||f||^2 as per-block partial sums; a launcher's residual gets them from
    k_sumsq_part, which must add in the order of the built-in family's fused sums (two rows per thread when m is even) or
    trials near the ratio thresholds drift by an ulp (tests/soak_device_fcn.py found these shapes)."""
    gen = dict(gamma=2.0, sigma=0.1, spread=5.0) if hard else dict(gamma=0.5, sigma=1e-3, spread=0.3)
    fcn = jac = ctx = None
    for seed in (3, 1001, 77777):
        A, b, xt, x0 = ds.generate(nprob, m, n, seed0=seed, **gen)
        o = ds.options(max_evals=500, factor_policy=policy, fuse_fd=0)
        x1, x2 = x0.clone(), x0.clone()
        f1, ib1, st1 = ds.lm_solve_batch(A, b, gen["gamma"], x1, o)
        fcn, jac, ctx = ds.dq_launchers(A, b, gen["gamma"])
        f2, ib2, st2 = ds.lm_solve_batch_device(fcn, ctx, m, x2, opts=o)
        assert st1 == st2 and ib1 == ib2
        assert np.array_equal(x1.cpu().numpy(), x2.cpu().numpy())
        assert np.array_equal(f1.cpu().numpy(), f2.cpu().numpy())


@pytest.mark.gpu
def test_dense_quadratic_launcher_with_user_jacobian(ds, oracle):
    """A user's jacobianfcn launcher replaces the forward differences (:241-243): LM with the analytic Jacobian of the
    dense-quadratic family == the oracle given the same Jacobian callback."""
    nprob, m, n = 3, 300, 24
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=99)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    x = x0.clone()
    f, ibs, st = ds.lm_solve_batch_device(fcn, ctx, m, x, jac=jac, opts=ds.options(max_evals=500))
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        bh = b[p].cpu().numpy()
        rc, xo, fo, ibo = oracle.lm_solve(lambda xx, ff: ff.__setitem__(slice(None), oracle.dq_residual(Ah, bh, 0.5, xx)), m, n,
                                          x0[p].cpu().numpy(),
                                          jac=lambda xx, jj: jj.__setitem__(slice(None), oracle.dq_jacobian(Ah, bh, 0.5, xx)),
                                          opts=oracle.default_options(max_evals=500))
        assert st[p] == rc and _same(ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(f[p].cpu().numpy(), fo)


# ---------------------------------------------------------------------------------------------------- GPU: FD Jacobian
@pytest.mark.gpu
@pytest.mark.parametrize("nprob,m,n", [(3, 512, 64), (2, 301, 37), (1, 130, 33), (5, 1024, 96), (1, 7, 3)])
def test_fd_jacobian_device_bitwise(ds, oracle, nprob, m, n):
    """vecfcn_helper%jacobian on a device function == the oracle's vfh_jac_fcn column by column (true division, :274);
    fv given and fv = NULL (then F(x) is evaluated first, :257-259)."""
    import torch
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=4242)
    x0[0, 0] = 0.0                                             # h = sqrt(eps) when x_j == 0 (:269-270)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    f0 = ds.residual(A, b, 0.5, x0)
    J1 = ds.fd_jacobian_device(fcn, ctx, m, x0, fv=f0)
    J2 = ds.fd_jacobian_device(fcn, ctx, m, x0)
    assert torch.equal(J1, J2)
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        Jo = oracle.dq_fd_jacobian(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy())
        assert np.array_equal(J1[p].cpu().numpy().T, Jo), p


@pytest.mark.gpu
def test_fd_jacobian_device_more_problems_than_a_grid_dimension(ds, oracle):
    """70,000 problems of 2 x 2: the problem index rides in gridDim.y / .z of the FD kernels (at most 65535), so the call is
    served in slices of NLH_MAX_LOCKSTEP like every other *_device entry point (include/nonlin_hip.h: the number of problems
    of a batch is not limited).  First, last and the problems either side of the slice boundary against the oracle."""
    import torch
    nprob, m, n = 70000, 2, 2
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=99, square_shift=True)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    J = ds.fd_jacobian_device(fcn, ctx, m, x0)
    torch.cuda.synchronize()
    for p in (0, 1, 65534, 65535, 65536, nprob - 1):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        Jo = oracle.dq_fd_jacobian(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy())
        assert np.array_equal(J[p].cpu().numpy().T, Jo), p


@pytest.mark.gpu
@pytest.mark.parametrize("m,n", [(4096, 256), (2048, 128), (1000, 40), (333, 65), (129, 31), (64, 64)])
def test_fd_jacobian_into_the_working_matrix_is_the_column_major_one(ds, m, n):
    """k_fd_jacobian_qrx (panel -> the exact factorisation's row-blocked matrix, turned through LDS) against k_fd_jacobian
    (panel -> column-major J): the exact factorisation of both must be the same bits, with partial tiles in both directions."""
    import torch
    nprob = 3
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=31)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    o = ds.options(max_evals=1, factor_policy=2)               # one outer iteration: J, lmfactor, lmpar, one trial
    x1, x2 = x0.clone(), x0.clone()
    f1, ib1, st1 = ds.lm_solve_batch(A, b, 0.5, x1, ds.options(max_evals=1, factor_policy=2, fuse_fd=0))   # column-major J, re-laid out
    f2, ib2, st2 = ds.lm_solve_batch_device(fcn, ctx, m, x2, opts=o)
    assert ib1 == ib2 and torch.equal(x1, x2) and torch.equal(f1, f2)


# ---------------------------------------------------------------------------------------------------- GPU: Newton / Broyden
def _oracle_square(oracle, batch, p, n, x0, broyden, analytic, **okw):
    oo = oracle.default_options(**okw)
    hc = batch.host_ctx(p)
    xo, fo, ibo = x0.copy(), np.zeros(n), oracle.IterationBehavior()
    jp = C.cast(batch.host_jac if analytic else None, oracle.JACFCN)
    if broyden:
        rc = oracle.lib().nlo_quasi_newton_solve(C.byref(oo), 5, C.cast(batch.host_fcn, oracle.VECFCN), jp, C.byref(hc), n,
                                                 xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    else:
        rc = oracle.lib().nlo_newton_solve(C.byref(oo), C.cast(batch.host_fcn, oracle.VECFCN), jp, C.byref(hc), n,
                                           xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
    return rc, xo, fo, ibo.as_dict()


@pytest.mark.gpu
@pytest.mark.parametrize("broyden", [False, True])
@pytest.mark.parametrize("analytic", [True, False])
@pytest.mark.parametrize("nprob,n", [(1, 10), (37, 50), (4, 300)])
def test_user_square_family_newton_and_broyden_bitwise(ds, oracle, nprob, n, analytic, broyden):
    """Broyden's tridiagonal system (a second user family, with an analytic jacobianfcn launcher): newton_solver and
    quasi_newton_solver through the launchers == the oracle driving the host twins, line search on."""
    import torch
    c, x0 = UM.btri_problems(nprob, n)
    batch = UM.BtriBatch(c)
    x = torch.tensor(x0, device=ds.device)
    fvec, ibs, status = ds.square_solve_batch_device(batch.launch, batch.ctx, x, jac=batch.launch_jac if analytic else None,
                                                     opts=ds.options(max_evals=500), broyden=broyden)
    xg, fg = x.cpu().numpy(), fvec.cpu().numpy()
    for p in range(nprob):
        rc, xo, fo, ibo = _oracle_square(oracle, batch, p, n, x0[p], broyden, analytic, max_evals=500)
        assert status[p] == rc, (p, status[p], rc)
        assert _same(ibs[p], ibo), (p, ibs[p], ibo)
        assert np.array_equal(xg[p], xo) and np.array_equal(fg[p], fo), p
    batch.close()


@pytest.mark.gpu
@pytest.mark.parametrize("analytic", [True, False])
def test_dense_quadratic_newton_through_the_launcher_reproduces_its_bits(ds, analytic):
    nprob, n = 5, 96
    A, b, xt, x0 = ds.generate(nprob, n, n, seed0=12345, square_shift=True)
    x1 = x0.clone()
    f1, ib1, st1 = ds.newton_solve_batch(A, b, 0.5, x1, analytic=analytic)
    fcn, jac, ctx = ds.dq_launchers(A, b, 0.5)
    x2 = x0.clone()
    f2, ib2, st2 = ds.square_solve_batch_device(fcn, ctx, x2, jac=jac if analytic else None)
    assert st1 == st2 and ib1 == ib2
    # (problem 0 of the forward-difference run diverges to NaN in the reference's arithmetic: the same NaNs on both paths)
    assert np.array_equal(x1.cpu().numpy(), x2.cpu().numpy(), equal_nan=True)
    assert np.array_equal(f1.cpu().numpy(), f2.cpu().numpy(), equal_nan=True)
    assert any(s == 0 for s in st1)


@pytest.mark.gpu
def test_launcher_failure_is_reported(ds):
    """A launcher that returns non-zero aborts the solve with a library error (no silent garbage)."""
    import torch
    from nonlin_amd import _lib
    t, y, xt, x0 = UM.lorentz_problems(2, 64, 1)
    batch = UM.LorentzBatch(t, y)
    x = torch.tensor(x0, device=ds.device)
    with pytest.raises(RuntimeError):
        ds.lm_solve_batch_device(batch.launch, batch.ctx, 65, torch.zeros((2, 3), dtype=torch.float64, device=ds.device))   # m mismatch -> 1
    ib = (_lib.IterationBehavior * 2)()
    rc = ds.lib.nlh_lm_solve_batch_device(ds.h.ptr, C.byref(ds.options()), 2, 64, 3, C.cast(None, _lib.DEVFCN), C.cast(None, _lib.DEVFCN),
                                          None, x.data_ptr(), x.data_ptr(), ib, None)
    assert rc == 211                                          # NL_UNDEFINED_FUNCTION_ERROR (:188)
    batch.close()


# ---------------------------------------------------------------------------------------------------- GPU: Fortran
@pytest.mark.gpu
def test_user_device_fcn_through_the_fortran_shim(ds, oracle, tmp_path):
    """vecfcn_helper%set_device_fcn + solver%solve (the reference's own call) and device_model_batch%create_from_device_fcn
    + solve_batch, from a Fortran program linked with the shim, libnonlin_hip.so and the user's own library
    (tests/fortran/device_fcn_suite.f90): least squares on the Lorentzian family, Newton and quasi-Newton on Broyden's
    tridiagonal family (analytic jacobianfcn launcher, and forward differences), bfgs on a chained-Rosenbrock objective (a model
    of one function; gradient launcher, and forward differences) -- every problem bitwise the oracle's."""
    import os
    import shutil
    import struct
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "fortran", "build", "device_fcn_suite")
    if not os.path.exists(exe):
        if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
            pytest.skip("no Fortran compiler and no prebuilt tests/fortran/build/device_fcn_suite")
        root = os.path.dirname(here)
        subprocess.check_call(["make", "-C", os.path.join(root, "nonlin_amd", "fortran"), "-s"])
        subprocess.check_call(["make", "-C", os.path.join(here, "device_model"), "-s"])
        subprocess.check_call(["make", "-C", os.path.join(here, "fortran"), "-s"])
    nprob, m, K, nq = 7, 256, 2, 40
    n = 3 * K
    t, y, xt, x0 = UM.lorentz_problems(nprob, m, K, seed=77, hard_every=3)
    c, xs = UM.btri_problems(nprob, nq, seed=5)
    path = str(tmp_path / "df.bin")
    with open(path, "wb") as fh:
        fh.write(struct.pack("<iii", nprob, m, n))
        fh.write(t.tobytes()); fh.write(y.tobytes()); fh.write(x0.tobytes())
        fh.write(struct.pack("<i", nq))
        fh.write(c.tobytes()); fh.write(xs.tobytes())
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    unhex = lambda h: struct.unpack(">d", bytes.fromhex(h))[0]
    res = {}
    for line in out.stdout.splitlines():
        tk = line.split()
        res.setdefault(tk[0], []).append({"status": int(tk[1]), "counts": (int(tk[2]), int(tk[3]), int(tk[4])), "flags": tuple(tk[5:8]),
                                          "x": np.array([unhex(h) for h in tk[8:]])})

    def cmp(r, rc, xo, ibo):
        assert r["status"] == rc == 0
        assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]), (r, ibo)
        assert r["flags"] == tuple("T" if ibo[k] else "F" for k in ("converge_on_fcn", "converge_on_chng", "converge_on_zero_diff"))
        assert np.array_equal(r["x"], xo)

    so = UM.lib()
    sols = []
    for p in range(nprob):
        hc = UM.LorentzHost(m, t[p].ctypes.data_as(dp), y[p].ctypes.data_as(dp), 0)
        sols.append(_oracle_lm(oracle, so.lorentz_host_fcn, hc, m, n, x0[p], max_evals=500))
    rc, xo, fo, ibo = sols[0]
    cmp(res["df_lm_single"][0], rc, xo, ibo)
    assert np.array_equal(res["df_lm_single_fvec"][0]["x"], np.array([fo[0], fo[-1]]))
    f0 = UM.lorentz_row_numpy(x0[0], t[0], y[0])
    assert np.array_equal(res["df_eval"][0]["x"], np.array([f0[0], f0[-1]]))
    assert len(res["df_lm_batch"]) == nprob
    for p in range(nprob):
        cmp(res["df_lm_batch"][p], *sols[p][:1], *sols[p][1:2], sols[p][3])

    class B:                                                   # the host twins of the square family, per problem
        host_fcn, host_jac = so.btri_host_fcn, so.btri_host_jac

        @staticmethod
        def host_ctx(p):
            return UM.BtriHost(float(c[p]), 0, 0)
    for key, broyden, analytic in (("df_newton_batch", False, True), ("df_broyden_batch", True, True), ("df_newton_fd_batch", False, False)):
        assert len(res[key]) == nprob
        for p in range(nprob):
            rc, xo, fo, ibo = _oracle_square(oracle, B, p, nq, xs[p], broyden, analytic, max_evals=500)
            cmp(res[key][p], rc, xo, ibo)
    rc, xo, fo, ibo = _oracle_square(oracle, B, 2, nq, xs[2], True, True, max_evals=500)       # quasi_newton_solver%solve itself
    cmp(res["df_broyden_single"][0], rc, xo, ibo)
    # constrained_least_squares_solver: solve_batch and solve on the user's device function, the box of the Fortran program
    lo, hi = np.tile([0.4, -1.0, 0.02], K), np.tile([1.2, 2.0, 0.2], K)
    L = oracle.lib()

    def cls_oracle(p):
        hc = UM.LorentzHost(m, t[p].ctypes.data_as(dp), y[p].ctypes.data_as(dp), 0)
        oo = oracle.default_options(max_evals=500)
        xo, fo, ibo = x0[p].copy(), np.zeros(m), oracle.IterationBehavior()
        rc = L.nlo_cls_solve(C.byref(oo), C.c_double(1.0), C.c_double(1.0), lo.ctypes.data_as(dp), hi.ctypes.data_as(dp),
                             C.cast(so.lorentz_host_fcn, oracle.VECFCN), C.cast(None, oracle.JACFCN), C.byref(hc), m, n,
                             xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
        return rc, xo, ibo.as_dict()
    assert len(res["df_cls_batch"]) == nprob
    for p in range(nprob):
        rc, xo, ibo = cls_oracle(p)
        cmp(res["df_cls_batch"][p], rc, xo, ibo)
    rc, xo, ibo = cls_oracle(1)
    cmp(res["df_cls_single"][0], rc, xo, ibo)
    # bfgs%solve_batch on a model of ONE function (the user's fcnnvar launcher; with its gradient launcher, and with forward differences)
    for key, analytic in (("df_bfgs_batch", True), ("df_bfgs_fd_batch", False)):
        assert len(res[key]) == nprob
        for p in range(nprob):
            cp = float(c[p])
            f_host = lambda xx: so.crosen_host_f(cp, nq, np.ascontiguousarray(xx).ctypes.data_as(dp))          # noqa: E731
            g_host = (lambda xx, g: so.crosen_host_grad(cp, nq, np.ascontiguousarray(xx).ctypes.data_as(dp), g.ctypes.data_as(dp))) if analytic else None
            rc, xo, fo, ibo = oracle.bfgs_solve(f_host, nq, xs[p], grad=g_host, opts=oracle.default_options(max_evals=500))
            r = res[key][p]
            assert r["status"] == rc, (key, p, r["status"], rc)
            assert r["counts"] == (ibo["iter_count"], ibo["fcn_count"], ibo["gradient_count"]), (key, p, r, ibo)
            assert r["flags"] == ("F", "T" if ibo["converge_on_chng"] else "F", "T" if ibo["converge_on_zero_diff"] else "F")
            assert r["x"][0] == fo and np.array_equal(r["x"][1:], xo)


# ---------------------------------------------------------------------------------------------------- GPU: bounded least squares
@pytest.mark.gpu
@pytest.mark.parametrize("nprob,m,K", [(1, 300, 2), (40, 256, 3)])
def test_user_family_bounded_least_squares_bitwise(ds, oracle, nprob, m, K):
    """constrained_least_squares_solver%solve (bounded dog-leg, src/nonlin_least_squares.f90:938-1176) on the user's family
    through the launcher, a box that binds some unknowns: == the oracle's cls_solve driving the host twin."""
    import torch
    n = 3 * K
    t, y, xt, x0 = UM.lorentz_problems(nprob, m, K, seed=31, hard_every=3)
    lo = np.tile([0.4, -1.0, 0.02], K)                        # amplitudes >= 0.4, widths >= 0.02
    hi = np.tile([1.2, 2.0, 0.2], K)                          # amplitudes <= 1.2: binds where a_k in (1.2, 1.5)
    batch = UM.LorentzBatch(t, y)
    x = torch.tensor(x0, device=ds.device)
    fvec, ibs, status = ds.cls_solve_batch_device(batch.launch, batch.ctx, m, x, opts=ds.options(max_evals=500), lower=lo, upper=hi)
    xg, fg = x.cpu().numpy(), fvec.cpu().numpy()
    L = oracle.lib()
    bound = 0
    for p in range(nprob):
        hc = batch.host_ctx(p)
        oo = oracle.default_options(max_evals=500)
        xo, fo, ibo = x0[p].copy(), np.zeros(m), oracle.IterationBehavior()
        rc = L.nlo_cls_solve(C.byref(oo), C.c_double(1.0), C.c_double(1.0), lo.ctypes.data_as(dp), hi.ctypes.data_as(dp),
                             C.cast(batch.host_fcn, oracle.VECFCN), C.cast(None, oracle.JACFCN), C.byref(hc), m, n,
                             xo.ctypes.data_as(dp), fo.ctypes.data_as(dp), C.byref(ibo))
        assert status[p] == rc, (p, status[p], rc)
        assert _same(ibs[p], ibo.as_dict()), (p, ibs[p], ibo.as_dict())
        assert np.array_equal(xg[p], xo) and np.array_equal(fg[p], fo), p
        bound += int(np.any(xo == lo) or np.any(xo == hi))
    assert nprob == 1 or bound > 0                            # the box did bind somewhere
    batch.close()


# ---------------------------------------------------------------------------------------------------- bfgs on a user's fcnnvar
BKEYS = ("iter_count", "fcn_count", "gradient_count", "converge_on_chng", "converge_on_zero_diff")


def test_user_scalar_family_host_twin():
    """(CPU) the chained-Rosenbrock objective's host twin and its gradient, term by term in the stated order."""
    so = UM.lib()
    n, c = 9, 1.2
    x = np.linspace(-0.7, 0.9, n)
    f = so.crosen_host_f(c, n, x.ctypes.data_as(dp))
    s = 0.0
    for i in range(n - 1):
        d = x[i + 1] - x[i] * x[i]
        e = c - x[i]
        s = s + (10.0 * (d * d) + e * e)
    assert f == s
    g = np.zeros(n)
    so.crosen_host_grad(c, n, x.ctypes.data_as(dp), g.ctypes.data_as(dp))
    h = 1e-7
    for i in (0, 4, n - 1):
        xp = x.copy(); xp[i] += h
        assert abs((so.crosen_host_f(c, n, xp.ctypes.data_as(dp)) - f) / h - g[i]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("analytic", [False, True])
@pytest.mark.parametrize("nprob,n,line_search", [(1, 6, True), (50, 12, True), (7, 40, True), (5, 10, False)])
def test_user_scalar_family_bfgs_bitwise(ds, oracle, nprob, n, analytic, line_search):
    """bfgs%solve on a user's device fcnnvar (launcher with m = 1; forward-difference gradient built on the device, or the
    user's gradient launcher): x, f, every count and flag of every problem equal to the CPU oracle driving the same
    arithmetic as a host callback (nlo_bfgs_solve, src/nonlin_optimize.f90:557-770)."""
    import torch
    so = UM.lib()
    c, x0 = UM.crosen_problems(nprob, n, seed=100 + n)
    batch = UM.BtriBatch(c)
    try:
        okw = dict(max_evals=500, use_line_search=1 if line_search else 0)
        x = torch.tensor(x0, dtype=torch.float64, device="cuda")
        fcn = ds._devfcn(batch.crosen_launch)
        grad = ds._devfcn(batch.crosen_launch_grad) if analytic else None
        fout, ibs, st = ds.bfgs_solve_batch_device(fcn, batch.ctx, x, grad=grad, opts=ds.options(**okw))
        xg = x.cpu().numpy()
        for p in range(nprob):
            cp = float(c[p])
            f_host = lambda xx: so.crosen_host_f(cp, n, np.ascontiguousarray(xx).ctypes.data_as(dp))          # noqa: E731
            g_host = (lambda xx, g: so.crosen_host_grad(cp, n, np.ascontiguousarray(xx).ctypes.data_as(dp), g.ctypes.data_as(dp))) if analytic else None
            rc, xo, fo, ibo = oracle.bfgs_solve(f_host, n, x0[p], grad=g_host, opts=oracle.default_options(**okw))
            assert st[p] == rc, (p, st[p], rc)
            for k in BKEYS:
                assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
            assert np.array_equal(xg[p], xo), (p, np.abs(xg[p] - xo).max())
            assert fout[p] == fo
    finally:
        batch.close()


@pytest.mark.gpu
def test_user_scalar_family_bfgs_errors(ds):
    """No function: NLH_UNDEFINED_FUNCTION_ERROR and zeroed counts (:611-615); a failing launcher is reported."""
    import torch
    from nonlin_amd import _lib
    null = C.cast(None, _lib.DEVFCN)
    x = torch.zeros((2, 4), dtype=torch.float64, device="cuda")
    ib = (_lib.IterationBehavior * 2)()
    ib[0].iter_count = 7
    o = ds.options()
    rc = ds.lib.nlh_bfgs_solve_batch_device(ds.h.ptr, C.byref(o), 2, 4, null, null, None, x.data_ptr(), None, ib, None)
    assert rc == _lib.load().nlh_lm_solve_batch_device(ds.h.ptr, C.byref(o), 1, 4, 2, null, null, None, x.data_ptr(), x.data_ptr(), None, None)
    assert ib[0].iter_count == 0
    c, x0 = UM.crosen_problems(2, 4)
    batch = UM.BtriBatch(c)
    try:
        # the btri residual launcher refuses m = 1 (it wants m == n): the library reports the user's failure
        with pytest.raises(RuntimeError):
            ds.bfgs_solve_batch_device(ds._devfcn(batch.launch), batch.ctx, torch.tensor(x0, dtype=torch.float64, device="cuda"))
    finally:
        batch.close()
