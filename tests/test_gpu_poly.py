"""GPU parity tests for polynomial%fit / fit_thru_zero (src/nonlin_polynomials.f90:146-238): Vandermonde panel +
Householder QR + back substitution, bitwise against the CPU restatement, and README Example 3's printed output."""
import numpy as np
import pytest
import torch

import problems_ref as P

pytestmark = pytest.mark.gpu


def test_readme_example_3(oracle):
    """README.md:175-226: c0..c3 = 1.1866141861, 0.4466136311, -.1223204989, 1.0647628218; max residual 0.50636."""
    import nonlin_amd as nl
    p = nl.polynomial()
    yc = P.YP.copy()
    p.fit(P.XP, P.YP.copy(), 3)
    c = p.get_all()
    assert ["%.10f" % v for v in c] == ["1.1866141861", "0.4466136311", "-0.1223204989", "1.0647628218"]
    assert "%.5f" % np.abs(p.evaluate(P.XP) - yc).max() == "0.50636"
    rc, co = oracle.poly_fit(P.XP, P.YP, 3)
    assert rc == 0 and np.array_equal(c, co)
    assert np.array_equal(p.evaluate(P.XP), oracle.poly_eval(co, P.XP))
    assert p.order() == 3 and p.get(1) == c[0]


def test_fit_thru_zero_and_errors(oracle):
    import nonlin_amd as nl
    p = nl.polynomial()
    p.fit_thru_zero(P.XP, P.YP.copy(), 3)
    rc, co = oracle.poly_fit(P.XP, P.YP, 3, thru_zero=True)
    assert np.array_equal(p.get_all(), co) and co[0] == 0.0
    with pytest.raises(nl.NonlinError) as e:
        p.fit(P.XP[:3], P.YP[:3], 3)             # order >= n, :163-166
    assert e.value.code == 4
    with pytest.raises(nl.NonlinError) as e:
        p.fit(P.XP, P.YP[:5], 2)                 # size mismatch, :159-162
    assert e.value.code == 3


@pytest.mark.parametrize("npts,order,thru_zero", [(21, 3, False), (500, 7, False), (64, 5, True), (4096, 12, False),
                                                  (9000, 6, False), (17000, 5, False), (18001, 4, False), (70000, 5, True)])
def test_poly_fit_batch_bitwise(ds, oracle, npts, order, thru_zero):
    """Every form of the Householder steps: fused (npts <= 4096), workgroup-wide tiles (<= 8192), reflector in LDS (<= 18000)
    and -- what the reference has no limit for -- reflector in global memory beyond that."""
    nprob = 5 if npts <= 4096 else 2
    rng = np.random.default_rng(npts + order)
    xs = np.sort(rng.uniform(-1.0, 1.0, size=(nprob, npts)), axis=1)
    ys = np.cos(3.0 * xs) + 0.01 * rng.standard_normal((nprob, npts))
    coef = ds.poly_fit_batch(torch.from_numpy(xs).to(ds.device), torch.from_numpy(ys).to(ds.device), order, thru_zero)
    for p in range(nprob):
        rc, co = oracle.poly_fit(xs[p], ys[p], order, thru_zero=thru_zero)
        assert rc == 0
        assert np.array_equal(coef[p].cpu().numpy(), co)
    if not thru_zero and order <= 7:
        ref = np.polynomial.polynomial.polyfit(xs[0], ys[0], order)
        assert np.abs(ref - coef[0].cpu().numpy()).max() <= 1e-8 * max(1.0, np.abs(ref).max())
