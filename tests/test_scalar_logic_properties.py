"""Independent checks of the scalar logic that product and oracle restate as twins (min_backtrack_search, alpha_box,
Coleman-Li scaling, limit_search_vector): the bitwise GPU-vs-oracle tests cannot see a shared misreading of the reference,
these properties can -- each is what the reference's own text says the routine computes
(src/nonlin_linesearch.f90:495-551, 554-572; src/nonlin_least_squares.f90:1181-1260), checked against brute force that
shares no code with either implementation.  CPU only."""
import ctypes as C

import numpy as np
import pytest


def _lib(oracle):
    L = oracle.lib()
    L.nlo_min_backtrack_search.restype = C.c_double
    L.nlo_min_backtrack_search.argtypes = [C.c_int32] + [C.c_double] * 6
    L.nlo_alpha_box.restype = C.c_double
    dp = C.POINTER(C.c_double)
    L.nlo_alpha_box.argtypes = [C.c_int32, dp, dp, dp, dp]
    L.nlo_coleman_li_scaling.restype = None
    L.nlo_coleman_li_scaling.argtypes = [C.c_int32, dp, dp, dp, dp]
    L.nlo_limit_search_vector.restype = None
    L.nlo_limit_search_vector.argtypes = [C.c_int32, dp, C.c_double]
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_quadratic_backtrack_minimises_the_interpolating_parabola(oracle):
    """mode 1 (:529-531): lam is the stationary point of the parabola q with q(0) = f0, q'(0) = slope, q(1) = f -- the
    first backtrack always comes from the full step alam = 1 (ls_search_mimo :294: mode == 1 only on the first pass)."""
    L = _lib(oracle)
    rng = np.random.default_rng(1)
    for _ in range(200):
        f0 = rng.uniform(0.1, 10)
        slope = -rng.uniform(0.01, 5)                         # a descent direction
        f = f0 + rng.uniform(0.0, 5)                          # the full step did not decrease enough
        lam = L.nlo_min_backtrack_search(1, f0, f, 0.0, 1.0, 0.0, slope)
        c2 = f - f0 - slope                                   # q(t) = f0 + slope t + c2 t^2
        ts = np.linspace(0, 1, 200001)
        tb = ts[np.argmin(f0 + slope * ts + c2 * ts * ts)]
        assert c2 > 0
        assert abs(lam - min(max(-slope / (2 * c2), 0), 1)) < 1e-12 and abs(lam - tb) < 1e-5


def test_cubic_backtrack_minimises_the_interpolating_cubic_and_is_capped(oracle):
    """mode /= 1 (:532-549): lam minimises the cubic c with c(0) = f0, c'(0) = slope, c(alam) = f, c(alam1) = f1 over
    (0, alam/2] -- the positive local minimiser where the discriminant allows, alam/2 otherwise or when larger (:548)."""
    L = _lib(oracle)
    rng = np.random.default_rng(2)
    checked = 0
    for _ in range(400):
        f0 = rng.uniform(0.5, 5)
        slope = -rng.uniform(0.05, 3)
        alam1 = rng.uniform(0.2, 1.0)
        alam = alam1 * rng.uniform(0.1, 0.5)
        f1 = f0 + rng.uniform(-0.2, 2.0)
        f = f0 + rng.uniform(-0.05, 1.0)
        lam = L.nlo_min_backtrack_search(2, f0, f, f1, alam, alam1, slope)
        assert lam <= 0.5 * alam * (1 + 1e-15)
        # the cubic through the four conditions, by a linear solve that shares nothing with the closed form
        M = np.array([[alam ** 3, alam ** 2], [alam1 ** 3, alam1 ** 2]])
        rhs = np.array([f - f0 - slope * alam, f1 - f0 - slope * alam1])
        a, b = np.linalg.solve(M, rhs)
        disc = b * b - 3 * a * slope
        if disc < 0 or abs(a) < 1e-9:
            if disc < 0:
                assert lam == 0.5 * alam
            continue
        roots = [t for t in ((-b + np.sqrt(disc)) / (3 * a), (-b - np.sqrt(disc)) / (3 * a)) if t > 0 and 6 * a * t + 2 * b > 0]
        if not roots:
            continue
        tmin = min(roots)                                     # the local minimiser of the cubic on t > 0
        want = min(tmin, 0.5 * alam)
        assert abs(lam - want) <= 1e-9 * max(1.0, abs(want)), (lam, want, a, b, disc)
        checked += 1
    assert checked > 100


def test_alpha_box_is_the_largest_feasible_step(oracle):
    """alpha_box (:1181-1219): the largest alpha >= 0 with xl <= x + alpha p <= xu; 0 when x is already outside a bound the
    step moves further across; huge when nothing binds."""
    L = _lib(oracle)
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(1, 9))
        xl = rng.uniform(-2, 0, n)
        xu = xl + rng.uniform(0.1, 3, n)
        x = xl + rng.uniform(0, 1, n) * (xu - xl)
        p = rng.uniform(-1, 1, n)
        p[rng.uniform(size=n) < 0.2] = 0.0
        a = L.nlo_alpha_box(n, _p(x), _p(p), _p(xl), _p(xu))
        if not p.any():
            assert a == np.finfo(float).max
            continue
        # brute force: feasibility is monotone in alpha on a box (a segment leaves a convex set once)
        feas = lambda t: bool(np.all(x + t * p >= xl - 1e-12) and np.all(x + t * p <= xu + 1e-12))
        assert feas(a * (1 - 1e-9)) and not feas(a * (1 + 1e-6) + 1e-9)
    x = np.array([0.5, 2.0]); p = np.array([0.0, 1.0]); xl = np.array([0.0, 0.0]); xu = np.array([1.0, 1.0])
    assert L.nlo_alpha_box(2, _p(x), _p(p), _p(xl), _p(xu)) == 0.0      # above the upper bound and moving up (:1202-1205)


def test_coleman_li_scaling_is_the_clamped_reciprocal_distance(oracle):
    """coleman_li_scaling (:1222-1260): s_i = 1 / max(distance to the nearest FINITE bound, 1e-8), capped at 1e8;
    1 for an unbounded variable."""
    L = _lib(oracle)
    big = np.finfo(float).max
    x = np.array([0.25, 0.9, 0.5, 3.0, 1e-12, 0.0])
    xl = np.array([0.0, 0.0, -big, 1.0, 0.0, -big])
    xu = np.array([1.0, 1.0, 2.0, big, 1.0, big])
    s = np.zeros(6)
    L.nlo_coleman_li_scaling(6, _p(x), _p(xl), _p(xu), _p(s))
    want = np.array([1 / 0.25, 1 / (1.0 - 0.9), 1 / 1.5, 1 / 2.0, 1e8, 1.0])
    assert np.allclose(s, want, rtol=1e-15, atol=0)
    assert s[4] == 1e8


def test_limit_search_vector_scales_only_long_vectors(oracle):
    L = _lib(oracle)
    v = np.array([3.0, 4.0])
    L.nlo_limit_search_vector(2, _p(v), 10.0)
    assert np.array_equal(v, [3.0, 4.0])
    L.nlo_limit_search_vector(2, _p(v), 2.5)
    assert np.allclose(v, [1.5, 2.0], rtol=1e-15) and abs(np.hypot(*v) - 2.5) < 1e-15
    z = np.zeros(3)
    L.nlo_limit_search_vector(3, _p(z), 1.0)
    assert not z.any()
