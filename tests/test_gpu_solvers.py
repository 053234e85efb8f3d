"""End-to-end GPU parity tests: least_squares_solver / newton_solver through the C ABI against
the CPU oracle on identical inputs.  Bar (BASELINE.json north_star): iteration / evaluation /
Jacobian counts and convergence flags exact, x within 1e-10 relative (written below)."""
import numpy as np
import pytest
import torch

import problems_ref as P

pytestmark = pytest.mark.gpu

RTOL_X = 1e-10     # north_star: "within 1e-10 relative for fp64"
# Forward-difference Jacobians amplify rounding: |dJ| ~ 2 sqrt(eps) |r| / |x|, a chaotic function of
# the last bits of x, so on problems with a nonzero residual at the solution two implementations
# that are not bit-identical (including the reference built by two compilers: see
# tests/test_oracle.py::test_reference_is_compiler_dependent_at_fd_noise_level) land
# ~sqrt(eps) * |r| apart.  The normal-equations policy is held to that bound; the exact policy
# (factor_policy=2, reference operation order) is held to bit-identity.
RTOL_X_FD_NOISE = 2e-6
COUNT_KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng",
              "converge_on_zero_diff")


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _counts_match(ib, ibo, strict):
    """strict: every count and flag identical.  Otherwise (normal-equations policy) the last trial step
    of a converged solve can be a coin toss -- actred = 1 - (fnorm1/fnorm)^2 is pure rounding noise there,
    in the reference as well -- so counts may differ by one and the terminating flag may differ."""
    if strict:
        return all(ib[k] == ibo[k] for k in COUNT_KEYS)
    return all(abs(ib[k] - ibo[k]) <= 1 for k in ("iter_count", "fcn_count", "jacobian_count"))


def _check_batch(ds, oracle, nprob, m, n, seed0, gen_kw, opt_kw, rtol=RTOL_X, bitwise=False, strict=True):
    gamma = gen_kw.get("gamma", 0.5)
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=seed0, **gen_kw)
    x = x0.clone()
    if "factor_policy" not in opt_kw:          # the library default is the exact policy; these cases are about the
        opt_kw = dict(opt_kw, factor_policy=0)  # normal-equations one (NLH_FACTOR_AUTO), an explicit opt-in
    fvec, ibs, status = ds.lm_solve_batch(A, b, gamma, x, ds.options(**opt_kw))
    worst = 0.0
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        okw = {k: v for k, v in opt_kw.items() if k in ("max_evals", "factor", "ftol", "xtol", "gtol")}
        rc, xo, fo, ibo, _, _ = oracle.dq_lm_solve(Ah, b[p].cpu().numpy(), gamma, x0[p].cpu().numpy(),
                                                   opts=oracle.default_options(**okw))
        assert status[p] == rc, (p, status[p], rc)
        assert _counts_match(ibs[p], ibo, strict), (p, ibs[p], ibo)
        r = _rel(x[p].cpu().numpy(), xo)
        worst = max(worst, r)
        assert r <= rtol, (p, r)
        if bitwise:
            assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(fvec[p].cpu().numpy(), fo)
        assert np.abs(fvec[p].cpu().numpy() - fo).max() <= max(rtol, 1e-12) * 10 * max(np.abs(fo).max(), 1.0)   # F at the solution
    return worst


@pytest.mark.parametrize("m,n,nprob", [(512, 64, 6), (2048, 128, 4), (300, 37, 5), (64, 16, 8)])
def test_lm_batch_default_regime(ds, oracle, m, n, nprob):
    """Default synthetic family (Gauss-Newton step accepted: Gram + pivoted-Cholesky path)."""
    _check_batch(ds, oracle, nprob, m, n, 12345, {}, dict(max_evals=500), rtol=RTOL_X_FD_NOISE, strict=False)


def test_lm_batch_c2_single_problem(ds, oracle):
    """BASELINE config 2: one 4096 x 256 problem, seed 12345 (reference: 5 / 5 / 4)."""
    A, b, xt, x0 = ds.generate(1, 4096, 256, seed0=12345)
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, factor_policy=0))
    rc, xo, fo, ibo, _, _ = oracle.dq_lm_solve(np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy(), 0.5,
                                               x0[0].cpu().numpy(), opts=oracle.default_options(max_evals=500))
    assert (ibo["iter_count"], ibo["fcn_count"], ibo["jacobian_count"]) == (5, 5, 4)     # recorded reference counts
    # the last step of this solve is ~3e-8 of |x| -- the size of the FD noise -- so its acceptance is a
    # rounding coin toss for any implementation that is not bit-identical (see the exact-policy test)
    assert _counts_match(ibs[0], ibo, strict=False), (ibs[0], ibo)
    assert _rel(x[0].cpu().numpy(), xo) <= RTOL_X_FD_NOISE


@pytest.mark.parametrize("m,n,gen,opt", [
    (512, 64, dict(gamma=2.0, sigma=0.1, spread=5.0), dict(factor=0.1)),
    (512, 64, dict(gamma=2.0, sigma=0.1, spread=5.0), dict()),
    (256, 32, dict(gamma=10.0, sigma=1.0, spread=50.0), dict(factor=0.1)),
    (256, 32, dict(gamma=10.0, sigma=1.0, spread=50.0), dict()),
])
def test_lm_batch_hard_regime(ds, oracle, m, n, gen, opt):
    """Trust region binding: the lmpar loop with both deviations from MINPACK is exercised
    (normal-equations path falls back to the faithful Householder QR + Givens lmsolve)."""
    _check_batch(ds, oracle, 3, m, n, 12345, gen, dict(max_evals=500, **opt), rtol=RTOL_X_FD_NOISE, strict=False)


@pytest.mark.parametrize("m,n", [(512, 64), (64, 16)])
def test_lm_batch_always_qr(ds, oracle, m, n):
    _check_batch(ds, oracle, 3, m, n, 2024, {}, dict(max_evals=500, factor_policy=1), rtol=RTOL_X_FD_NOISE,
                 strict=False)


@pytest.mark.parametrize("m,n,nprob,gen,opt", [
    (512, 64, 4, {}, {}),
    (300, 37, 3, {}, {}),
    (64, 16, 4, {}, {}),
    (512, 64, 2, dict(gamma=2.0, sigma=0.1, spread=5.0), dict(factor=0.1)),
    (512, 64, 2, dict(gamma=2.0, sigma=0.1, spread=5.0), {}),
    (256, 32, 2, dict(gamma=10.0, sigma=1.0, spread=50.0), dict(factor=0.1)),
    (256, 32, 2, dict(gamma=10.0, sigma=1.0, spread=50.0), {}),
    (2048, 128, 2, {}, {}),
])
def test_lm_batch_exact_policy_bitwise(ds, oracle, m, n, nprob, gen, opt):
    """factor_policy = NLH_FACTOR_EXACT: every reduction in the reference's operation order.
    x and fvec are bit-identical to the CPU path, counts exact -- including the trust-region-binding
    regime that exercises lmpar's two deviations from MINPACK."""
    _check_batch(ds, oracle, nprob, m, n, 12345, gen, dict(max_evals=500, factor_policy=2, **opt),
                 rtol=0.0, bitwise=True)


def test_lm_exact_policy_c2_bitwise(ds, oracle):
    """BASELINE config 2 (4096 x 256, seed 12345) under the exact policy: bit-identical x."""
    _check_batch(ds, oracle, 1, 4096, 256, 12345, {}, dict(max_evals=500, factor_policy=2), rtol=0.0, bitwise=True)


def test_lm_batch_zero_residual(ds, oracle):
    """Zero-residual problems: the FD noise does not reach the solution, so x agrees to 1e-10 under the
    normal-equations policy too (counts can still differ by one at the noise-level last step)."""
    _check_batch(ds, oracle, 4, 256, 32, 7, dict(sigma=0.0), dict(max_evals=500), rtol=RTOL_X, strict=False)
    _check_batch(ds, oracle, 4, 256, 32, 7, dict(sigma=0.0), dict(max_evals=500, factor_policy=2), rtol=0.0, bitwise=True)


def test_lm_batch_max_evals_error(ds, oracle):
    """Too few evaluations: status NL_CONVERGENCE_ERROR for every problem, counts still exact."""
    _check_batch(ds, oracle, 2, 256, 32, 12345, dict(gamma=10.0, sigma=1.0, spread=50.0), dict(max_evals=5),
                 rtol=RTOL_X_FD_NOISE, strict=False)
    _check_batch(ds, oracle, 2, 256, 32, 12345, dict(gamma=10.0, sigma=1.0, spread=50.0),
                 dict(max_evals=5, factor_policy=2), rtol=0.0, bitwise=True)


def test_lm_linearity_property_full_size(ds):
    """Size-independent property at BASELINE size 4096 x 256: with gamma = 0 the model is linear,
    so LM must land on the least-squares solution: J^T r = 0 to rounding."""
    A, b, xt, x0 = ds.generate(2, 4096, 256, seed0=99, gamma=0.0, sigma=1e-3)
    for policy in (2, 0):
        x = x0.clone()
        fvec, ibs, status = ds.lm_solve_batch(A, b, 0.0, x, ds.options(max_evals=100, factor_policy=policy))
        assert status == [0, 0]
        grad = torch.matmul(A, fvec.unsqueeze(-1)).squeeze(-1)      # A^T r per problem
        # the FD Jacobian carries ~2 sqrt(eps) relative noise, so J_fd^T r = 0 leaves A^T r at that level
        assert float(grad.abs().max()) <= 1e-5 * float(fvec.norm(dim=1).max())


# ---------------------------------------------------------------------------
# host-callback drop-in API on the reference's own test problems
# ---------------------------------------------------------------------------
def _solve_lm_host(fcn, m, n, x0, jac=None, policy=0, **opts):
    """policy 0 = NLH_FACTOR_AUTO (normal equations, QR fallback), 2 = NLH_FACTOR_EXACT."""
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(fcn, m, n)
    if jac is not None:
        obj.set_jacobian(jac)
    s = nl.least_squares_solver()
    s.factor_policy = policy
    if "max_evals" in opts:
        s.set_max_fcn_evals(opts["max_evals"])
    if "factor" in opts:
        s.set_step_scaling_factor(opts["factor"])
    x = np.array(x0, dtype=np.float64)
    f = np.zeros(m)
    ib = nl.iteration_behavior()
    s.solve(obj, x, f, ib)
    return x, f, ib


def test_host_lm_readme_example_2(oracle):
    """BASELINE config 1 / README Example 2 / test_least_squares_3 through the drop-in API."""
    x, f, ib = _solve_lm_host(P.lsfcn1, 21, 4, [1.0] * 4)
    rc, xo, fo, ibo = oracle.lm_solve(lambda xx, ff: P.lsfcn1(xx, ff, None), 21, 4, [1.0] * 4)
    assert (ib.iter_count, ib.fcn_count, ib.jacobian_count) == (2, 3, 2)
    assert ib.converge_on_fcn and not ib.converge_on_chng and not ib.converge_on_zero_diff
    assert _rel(x, xo) <= RTOL_X
    # README.md:165-171, ten printed digits
    np.testing.assert_allclose(x, [1.0647627571, -0.1223202909, 0.4466134462, 1.1866142244], rtol=0, atol=6e-11)
    assert abs(np.abs(f).max() - 0.50636) < 5e-6


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
@pytest.mark.parametrize("analytic", [True, False])
def test_host_lm_fcn1(oracle, ic, analytic):
    """test_least_squares_1 / test_least_squares_4 (tests/nonlin_test_solve.f90:537-584, 657-726)."""
    jac = P.jac1 if analytic else None
    x, f, ib = _solve_lm_host(P.fcn1, 2, 2, ic, jac=jac)
    rc, xo, fo, ibo = oracle.lm_solve(lambda xx, ff: P.fcn1(xx, ff, None), 2, 2, ic,
                                      jac=(lambda xx, JJ: P.jac1(xx, JJ, None)) if analytic else None)
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6      # is_ans_1
    assert _counts_match(ib.as_dict(), ibo, strict=False), (ib.as_dict(), ibo)
    assert _rel(x, xo) <= RTOL_X


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_host_lm_fcn2_badly_scaled(oracle, ic):
    """test_least_squares_2 (:587-634): ill-conditioned Gram matrix => Householder QR path."""
    x, f, ib = _solve_lm_host(P.fcn2, 2, 2, ic, max_evals=1000)
    rc, xo, fo, ibo = oracle.lm_solve(lambda xx, ff: P.fcn2(xx, ff, None), 2, 2, ic,
                                      opts=oracle.default_options(max_evals=1000))
    assert abs(abs(x[0]) - 5.0e3) <= 1e-6 and abs(abs(x[1]) - 10.0) <= 1e-6    # is_ans_2
    assert _counts_match(ib.as_dict(), ibo, strict=False), (ib.as_dict(), ibo)
    assert _rel(x, xo) <= RTOL_X


@pytest.mark.parametrize("name,fcn,jac,m,n,x0,kw", [
    ("readme", P.lsfcn1, None, 21, 4, [1.0] * 4, {}),
    ("fcn1_fd_a", P.fcn1, None, 2, 2, (0.5, 0.5), {}),
    ("fcn1_fd_b", P.fcn1, None, 2, 2, (1.0, 1.0), {}),
    ("fcn1_an_a", P.fcn1, P.jac1, 2, 2, (0.5, 0.5), {}),
    ("fcn1_an_b", P.fcn1, P.jac1, 2, 2, (1.0, 1.0), {}),
    ("fcn2_a", P.fcn2, None, 2, 2, (0.5, 0.5), dict(max_evals=1000)),
    ("fcn2_b", P.fcn2, None, 2, 2, (1.0, 1.0), dict(max_evals=1000)),
    ("readme_binding", P.lsfcn1, None, 21, 4, [1.0] * 4, dict(factor=0.1)),
])
def test_host_lm_exact_policy_bitwise(oracle, name, fcn, jac, m, n, x0, kw):
    """The reference's own LM test problems through the drop-in API with the exact policy:
    x, fvec, every count and flag identical to the CPU path."""
    x, f, ib = _solve_lm_host(fcn, m, n, x0, jac=jac, policy=2, **kw)
    okw = dict(kw)
    rc, xo, fo, ibo = oracle.lm_solve(lambda xx, ff: fcn(xx, ff, None), m, n, x0,
                                      jac=(lambda xx, JJ: jac(xx, JJ, None)) if jac else None,
                                      opts=oracle.default_options(**okw))
    assert rc == 0
    for k in COUNT_KEYS:
        assert getattr(ib, k) == ibo[k], (k, ib.as_dict(), ibo)
    assert np.array_equal(x, xo), (x - xo)
    assert np.array_equal(f, fo)
    if name == "readme":
        # bit pattern of the reference's own answer (amdflang build)
        import struct
        hexes = " ".join("%016X" % struct.unpack(">Q", struct.pack(">d", v))[0] for v in x)
        assert hexes == "3FF10944AC39F2FE BFBF5061F1058060 3FDC9550905F97F8 3FF2FC5F326EDD8B"


def test_host_lm_args_passthrough():
    """class(*) args reaches the callback untouched (tests/nonlin_test_solve.f90:54-57)."""
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(P.fcn1a, 2, 2)
    s = nl.least_squares_solver()
    x = np.array([1.0, 1.0])
    f = np.zeros(2)
    s.solve(obj, x, f, args=2.0)
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6


def test_host_errors():
    import nonlin_amd as nl
    s = nl.least_squares_solver()
    obj = nl.vecfcn_helper()
    with pytest.raises(nl.NonlinError) as e:
        s.solve(obj, np.zeros(2), np.zeros(2))
    assert e.value.code == nl.NL_UNDEFINED_FUNCTION_ERROR            # :188
    obj.set_fcn(P.fcn1, 2, 3)
    with pytest.raises(nl.NonlinError) as e:
        s.solve(obj, np.zeros(3), np.zeros(2))
    assert e.value.code == nl.NL_UNDERDEFINED_PROBLEM_ERROR           # :189
    obj.set_fcn(P.fcn1, 2, 2)
    with pytest.raises(nl.NonlinError) as e:
        s.solve(obj, np.zeros(3), np.zeros(2))
    assert e.value.code == 3                                          # :191-192
    ns = nl.newton_solver()
    obj.set_fcn(P.lsfcn1, 21, 4)
    with pytest.raises(nl.NonlinError) as e:
        ns.solve(obj, np.zeros(4), np.zeros(21))
    assert e.value.code == nl.NL_INVALID_INPUT_ERROR                  # src/nonlin_solve.f90:519


@pytest.mark.parametrize("x", [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (0.5, -0.5)])
def test_host_fd_jacobian_polar(oracle, x):
    """test_jacobian_1 (tests/nonlin_test_jacobian.f90:89-177): FD vs exact within 1e-4, and
    bit-identical to the oracle's vfh_jac_fcn."""
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(P.polar, 2, 2)
    xx = np.array(x, dtype=np.float64)
    J = np.zeros((2, 2), order="F")
    obj.jacobian(xx, J)
    exact = np.zeros((2, 2), order="F")
    P.polar_jac(xx, exact, None)
    assert np.abs(J - exact).max() <= 1e-4
    Jo = oracle.fd_jacobian(lambda a, f: P.polar(a, f, None), 2, 2, x)
    assert np.array_equal(J, Jo)
    assert np.array_equal(xx, np.array(x))          # x restored (:273)


def test_host_fd_jacobian_args(oracle):
    """test_jacobian_2 (:180-269): scalar args multiplies the model."""
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(P.polar_scaled, 2, 2)
    xx = np.array([0.5, -0.5])
    J = np.zeros((2, 2), order="F")
    obj.jacobian(xx, J, args=0.37)
    exact = np.zeros((2, 2), order="F")
    P.polar_scaled_jac(xx, exact, 0.37)
    assert np.abs(J - exact).max() <= 1e-4


def _solve_newton_host(fcn, n, x0, jac=None, use_ls=True, **opts):
    import nonlin_amd as nl
    obj = nl.vecfcn_helper()
    obj.set_fcn(fcn, n, n)
    if jac is not None:
        obj.set_jacobian(jac)
    s = nl.newton_solver()
    s.set_use_line_search(use_ls)
    x = np.array(x0, dtype=np.float64)
    f = np.zeros(n)
    ib = nl.iteration_behavior()
    s.solve(obj, x, f, ib, args=opts.get("args"))
    return x, f, ib


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_host_newton_1(oracle, ic):
    """test_newton_1 (:362-409); (1,1) reproduces the recorded reference counts 6 / 9 / 6."""
    x, f, ib = _solve_newton_host(P.fcn1, 2, ic, jac=P.jac1)
    rc, xo, fo, ibo = oracle.newton_solve(lambda a, b: P.fcn1(a, b, None), 2, ic, jac=lambda a, b: P.jac1(a, b, None))
    assert abs(abs(x[0]) - 5.0) <= 1e-6 and abs(abs(x[1]) - 3.0) <= 1e-6
    for k in COUNT_KEYS:
        assert getattr(ib, k) == ibo[k], (k, ib.as_dict(), ibo)
    assert np.array_equal(x, xo)          # LU, line search and model are bit-identical to the CPU path
    if ic == (1.0, 1.0):
        assert (ib.iter_count, ib.fcn_count, ib.jacobian_count) == (6, 9, 6)


@pytest.mark.parametrize("ic", [(0.5, 0.5), (1.0, 1.0)])
def test_host_newton_2_fd_no_linesearch(oracle, ic):
    """test_newton_2 (:412-462): badly scaled, FD Jacobian, line search off."""
    x, f, ib = _solve_newton_host(P.fcn2, 2, ic, use_ls=False)
    rc, xo, fo, ibo = oracle.newton_solve(lambda a, b: P.fcn2(a, b, None), 2, ic,
                                          opts=oracle.default_options(use_line_search=0))
    assert abs(abs(x[0]) - 5.0e3) <= 1e-6 and abs(abs(x[1]) - 10.0) <= 1e-6
    for k in COUNT_KEYS:
        assert getattr(ib, k) == ibo[k]
    assert np.array_equal(x, xo)


def test_host_newton_4_powell(oracle):
    """test_newton_4 (:806-848): Powell badly scaled, analytic Jacobian, tol 1e-5."""
    x, f, ib = _solve_newton_host(P.powell, 2, [0.0, 1.0], jac=P.powell_jac)
    assert abs(x[0] - 1.098159e-5) <= 1e-5 and abs(x[1] - 9.106146) <= 1e-5
    rc, xo, fo, ibo = oracle.newton_solve(lambda a, b: P.powell(a, b, None), 2, [0.0, 1.0],
                                          jac=lambda a, b: P.powell_jac(a, b, None))
    for k in COUNT_KEYS:
        assert getattr(ib, k) == ibo[k]
    assert np.array_equal(x, xo)


def test_host_newton_fsolve_example(oracle):
    """examples/nonlin_newton_solve_jacobian.f90: 2x1 - x2 = exp(-x1), -x1 + 2x2 = exp(-x2)."""
    x, f, ib = _solve_newton_host(P.misc01, 2, [1.0, 1.0], jac=P.misc01_jac)
    assert np.abs(x - 0.5671432904097838).max() <= 1e-7
    rc, xo, fo, ibo = oracle.newton_solve(lambda a, b: P.misc01(a, b, None), 2, [1.0, 1.0],
                                          jac=lambda a, b: P.misc01_jac(a, b, None))
    assert np.array_equal(x, xo)


@pytest.mark.parametrize("n,analytic", [(64, True), (64, False), (200, True)])
def test_dq_newton_batch(ds, oracle, n, analytic):
    """BASELINE config 3 family at test size: square dense-quadratic system, A <- 2I + A."""
    A, b, xt, x0 = ds.generate(2, n, n, seed0=12345, sigma=0.0, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.newton_solve_batch(A, b, 0.5, x, analytic=analytic, opts=ds.options(max_evals=500))
    for p in range(2):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_newton_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=analytic,
                                                    opts=oracle.default_options(max_evals=500))
        assert status[p] == rc == 0
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (k, ibs[p], ibo)
        assert _rel(x[p].cpu().numpy(), xo) <= RTOL_X


def test_dq_newton_c3_full_size(ds, oracle):
    """BASELINE config 3 at full size: newton_solver on an n = 1024 square dense-quadratic system with
    the analytic Jacobian, line search on.  The blocked LU keeps the reference's operation order, so x,
    fvec and every count are bit-identical to the CPU path."""
    n = 1024
    A, b, xt, x0 = ds.generate(1, n, n, seed0=12345, sigma=0.0, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.newton_solve_batch(A, b, 0.5, x, analytic=True, opts=ds.options(max_evals=500))
    Ah = np.asfortranarray(A[0].cpu().numpy().T)
    rc, xo, fo, ibo, _ = oracle.dq_newton_solve(Ah, b[0].cpu().numpy(), 0.5, x0[0].cpu().numpy(), analytic=True,
                                                opts=oracle.default_options(max_evals=500))
    assert status[0] == rc == 0
    for k in COUNT_KEYS:
        assert ibs[0][k] == ibo[k], (k, ibs[0], ibo)
    assert np.array_equal(x[0].cpu().numpy(), xo)
    assert np.array_equal(fvec[0].cpu().numpy(), fo)
    assert np.abs(fo).max() < 1e-8


@pytest.mark.parametrize("n,seed,analytic,spread", [(64, 2506, True, 0.3), (200, 73602, True, 0.3), (200, 26774, False, 0.3),
                                                    (200, 19210, False, 0.1), (33, 22757, True, 0.3), (129, 63083, False, 0.3)])
def test_newton_backtracking_steps_bitwise(ds, oracle, n, seed, analytic, spread):
    """Starts far enough from the root that the line search backtracks (fcn_count > iter_count + 1): the step length
    then comes out of min_backtrack_search, which takes slope = dot(grad, dir) as an input, so grad = J^T f has to be
    summed in the reference's order too.  (Found by a randomised sweep: with a tree-reduced gradient these six cases
    differed from the CPU path in the last bits of x.)"""
    A, b, xt, x0 = ds.generate(2, n, n, seed0=seed, sigma=0.0, spread=spread, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.newton_solve_batch(A, b, 0.5, x, analytic=analytic, opts=ds.options(max_evals=300))
    backtracked = False
    for p in range(2):
        rc, xo, fo, ibo = oracle.dq_newton_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5,
                                                  x0[p].cpu().numpy(), analytic=analytic,
                                                  opts=oracle.default_options(max_evals=300))[:4]
        assert status[p] == rc
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(x[p].cpu().numpy(), xo) and np.array_equal(fvec[p].cpu().numpy(), fo)
        backtracked = backtracked or ibo["fcn_count"] > ibo["iter_count"] + 1
    assert backtracked


def test_lm_c4_batch_property_and_spot_parity(ds, oracle):
    """BASELINE config 4 shape (2048 x 128 problems, 256 of them here): every problem converges, counts
    are in the recorded range, and a spot check of three problems against the oracle holds the FD-noise bound."""
    nprob, m, n = 256, 2048, 128
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=12345)
    x = x0.clone()
    fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, factor_policy=0))
    assert all(s == 0 for s in status)
    assert all(3 <= ib["jacobian_count"] <= 8 for ib in ibs)
    # residual norm at the solution ~ sigma * sqrt(m/3): the noise floor of the generator
    fn = fvec.norm(dim=1).cpu().numpy()
    assert np.all(fn < 3 * 1e-3 * np.sqrt(m / 3.0)) and np.all(fn > 0.3 * 1e-3 * np.sqrt(m / 3.0))
    for p in (0, 101, 255):
        rc, xo, fo, ibo, _, _ = oracle.dq_lm_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5,
                                                   x0[p].cpu().numpy(), opts=oracle.default_options(max_evals=500))
        assert rc == 0 and _counts_match(ibs[p], ibo, strict=False)
        assert _rel(x[p].cpu().numpy(), xo) <= RTOL_X_FD_NOISE


def test_lm_batch_larger_than_the_chip(ds, oracle):
    """More problems than CUs (320 > 256): the Cholesky and lmpar stages then run as 512-thread workgroups, two to a
    CU, and 4096 x 256 problems take the whole-triangle Gram kernel.  Every problem converges; spot parity with the
    oracle at the FD-noise bound, for a small shape and for the headline shape."""
    for (nprob, m, n, spots) in ((320, 512, 128, (0, 160, 319)), (288, 4096, 256, (287,))):
        A, b, xt, x0 = ds.generate(nprob, m, n, seed0=4242)
        x = x0.clone()
        fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, factor_policy=0))
        assert all(s == 0 for s in status)
        for p in spots:
            rc, xo, fo, ibo, _, _ = oracle.dq_lm_solve(np.asfortranarray(A[p].cpu().numpy().T), b[p].cpu().numpy(), 0.5,
                                                       x0[p].cpu().numpy(), opts=oracle.default_options(max_evals=500))
            assert rc == 0 and _counts_match(ibs[p], ibo, strict=False)
            assert _rel(x[p].cpu().numpy(), xo) <= RTOL_X_FD_NOISE


@pytest.mark.parametrize("m,n", [(1024, 300), (700, 513)])
def test_lm_wide_problems_both_policies(ds, oracle, m, n):
    """n > 256 (several Gram blocks per row, multi-panel blocked Cholesky, n + 1 > 256 column threads):
    exact policy bit-identical, normal-equations policy within the FD-noise bound."""
    _check_batch(ds, oracle, 2, m, n, 4321, {}, dict(max_evals=500, factor_policy=2), rtol=0.0, bitwise=True)
    _check_batch(ds, oracle, 2, m, n, 4321, {}, dict(max_evals=500), rtol=RTOL_X_FD_NOISE, strict=False)


def test_host_lm_callback_medium_problem(oracle):
    """Host-callback mode at a size where the panel upload, the 16-byte FD path and the MFMA Gram kernel all
    matter: the dense-quadratic residual evaluated by a numpy callback in the oracle's operation order."""
    import nonlin_amd as nl
    m, n, gamma = 512, 64, 0.5
    A, b, xt, x0 = oracle.dq_generate(2468, m, n)

    def fcn(x, f, args=None):
        u = np.zeros(m)
        for j in range(n):                 # column sweep: each row accumulates j ascending, mul then add
            u = u + A[:, j] * x[j]
        f[:] = (u + (gamma * u) * u) - b

    for policy, bitwise in ((2, True), (0, False)):
        obj = nl.vecfcn_helper()
        obj.set_fcn(fcn, m, n)
        s = nl.least_squares_solver()
        s.factor_policy = policy
        s.set_max_fcn_evals(500)
        x = x0.copy()
        f = np.zeros(m)
        ib = nl.iteration_behavior()
        s.solve(obj, x, f, ib)
        rc, xo, fo, ibo, _, _ = oracle.dq_lm_solve(A, b, gamma, x0, opts=oracle.default_options(max_evals=500))
        assert rc == 0
        assert _counts_match(ib.as_dict(), ibo, strict=bitwise), (ib.as_dict(), ibo)
        if bitwise:
            assert np.array_equal(x, xo) and np.array_equal(f, fo)
        else:
            assert _rel(x, xo) <= RTOL_X_FD_NOISE


@pytest.mark.parametrize("m,n,policy", [(512, 64, 2), (4096, 256, 0), (300, 37, 2)])
def test_fused_fd_epilogue_is_bitwise_the_two_kernel_path(ds, m, n, policy):
    """opts.fuse_fd: the panel kernel forms (f_j - f0)/h_j itself; same operations per element, so x, fvec and
    all counts equal the panel + k_fd_jacobian path bit for bit under every factor policy."""
    A, b, xt, x0 = ds.generate(3, m, n, seed0=4711)
    outs = []
    for fuse in (0, 1):
        x = x0.clone()
        fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, factor_policy=policy, fuse_fd=fuse))
        outs.append((x.cpu().numpy(), fvec.cpu().numpy(), ibs, status))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]


@pytest.mark.parametrize("n,nprob,analytic,spread", [(64, 48, True, 0.3), (200, 24, False, 0.3), (129, 40, True, 2.0),
                                                     (256, 64, True, 0.3)])
def test_dq_newton_lockstep_batch_bitwise(ds, oracle, n, nprob, analytic, spread):
    """The lock-step Newton state machine (nlh_kernels_newton.h) on a batch whose problems do NOT march together: some
    converge in a few iterations, some backtrack, a large spread makes some stop in the line search.  Every problem must
    carry the bits, counts, flags and status the CPU path gives it alone (a problem's arithmetic never depends on which
    other problems share its round)."""
    A, b, xt, x0 = ds.generate(nprob, n, n, seed0=4711, sigma=0.0, spread=spread, square_shift=True)
    x = x0.clone()
    fvec, ibs, status = ds.newton_solve_batch(A, b, 0.5, x, analytic=analytic, opts=ds.options(max_evals=60))
    xs, fs = x.cpu().numpy(), fvec.cpu().numpy()
    iters = set()
    for p in range(nprob):
        Ah = np.asfortranarray(A[p].cpu().numpy().T)
        rc, xo, fo, ibo, _ = oracle.dq_newton_solve(Ah, b[p].cpu().numpy(), 0.5, x0[p].cpu().numpy(), analytic=analytic,
                                                    opts=oracle.default_options(max_evals=60))
        assert status[p] == rc, (p, status[p], rc)
        for k in COUNT_KEYS:
            assert ibs[p][k] == ibo[k], (p, k, ibs[p], ibo)
        assert np.array_equal(xs[p], xo), p
        assert np.array_equal(fs[p], fo), p
        iters.add((ibo["iter_count"], ibo["fcn_count"]))
    assert len(iters) > 1                    # the batch really was heterogeneous


def test_print_status_of_a_device_model_solve_matches_the_host_loop(ds, oracle, capfd):
    """set_print_status(.true.) on a single device-model solve: the lock-step drivers print the reference's status block
    (src/nonlin_helper.f90:17-33) after every outer iteration that goes on, as the host-callback loop does -- the same
    text, since both run the same arithmetic.  (Batches stay silent: the block belongs to one solve.)"""
    import ctypes
    import nonlin_amd as nl
    libc = ctypes.CDLL(None)

    def captured():
        libc.fflush(None)                 # the library prints through C stdio
        return capfd.readouterr().out
    m, n = 96, 12
    A, b, xt, x0 = ds.generate(1, m, n, seed0=99, spread=2.0)
    Ah, bh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy()
    o = ds.options(max_evals=200)
    o.print_status = 1
    captured()
    x = x0.clone()
    ds.lm_solve_batch(A, b, 0.5, x, o)
    dev_out = captured()
    obj = nl.vecfcn_helper()
    obj.set_fcn(lambda xx, ff, args=None: ff.__setitem__(slice(None), oracle.dq_residual(Ah, bh, 0.5, np.array(xx))), m, n)
    s = nl.least_squares_solver()
    s.factor_policy = 2
    s.set_max_fcn_evals(200)
    s.set_print_status(True)
    xh, fh, ib = x0[0].cpu().numpy().copy(), np.zeros(m), nl.iteration_behavior()
    s.solve(obj, xh, fh, ib)
    host_out = captured()
    assert "Iteration:" in host_out and dev_out == host_out
    assert np.array_equal(x[0].cpu().numpy(), xh)
    # Newton: same comparison against the host loop with an analytic Jacobian callback
    n = 24
    A, b, xt, x0 = ds.generate(1, n, n, seed0=3, sigma=0.0, spread=1.0, square_shift=True)
    Ah, bh = np.asfortranarray(A[0].cpu().numpy().T), b[0].cpu().numpy()
    x = x0.clone()
    ds.newton_solve_batch(A, b, 0.5, x, analytic=True, opts=o)
    dev_out = captured()
    obj = nl.vecfcn_helper()
    obj.set_fcn(lambda xx, ff, args=None: ff.__setitem__(slice(None), oracle.dq_residual(Ah, bh, 0.5, np.array(xx))), n, n)
    obj.set_jacobian(lambda xx, JJ, args=None: JJ.__setitem__((slice(None), slice(None)), oracle.dq_jacobian(Ah, bh, 0.5, np.array(xx))))
    s = nl.newton_solver()
    s.set_max_fcn_evals(200)
    s.set_print_status(True)
    xh, fh = x0[0].cpu().numpy().copy(), np.zeros(n)
    s.solve(obj, xh, fh, nl.iteration_behavior())
    host_out = captured()
    assert "Iteration:" in host_out and dev_out == host_out
    assert np.array_equal(x[0].cpu().numpy(), xh)
