"""The multi-CU blocked Cholesky of the opt-in normal-equations policy (k_chol_mc_begin / _step / _end: a launch per panel
step over many CUs, for a handful of problems -- BASELINE config 5) against the one-workgroup-per-problem kernel it
stands in for (k_chol_nopiv): the same operations in the same order, so an LM solve under NLH_FACTOR_AUTO must produce the
same bits either way (NLH_CHOL_MC = largest number of active problems that takes the multi-CU form; 0 = never)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,m,n", [(1, 3000, 512), (3, 2048, 256), (2, 1500, 200), (1, 4100, 500), (1, 2000, 700)])
def test_multi_cu_cholesky_same_bits_as_single_workgroup(ds, nb, m, n):
    A, b, xt, x0 = ds.generate(nb, m, n, seed0=4242)
    res = {}
    old = os.environ.get("NLH_CHOL_MC")
    try:
        for mc in ("0", "8"):
            os.environ["NLH_CHOL_MC"] = mc
            x = x0.clone()
            fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=500, factor_policy=0))
            torch.cuda.synchronize()
            res[mc] = (x.cpu().numpy().copy(), fvec.cpu().numpy().copy(), [tuple(sorted(i.items())) for i in ibs], list(status))
    finally:
        if old is None:
            os.environ.pop("NLH_CHOL_MC", None)
        else:
            os.environ["NLH_CHOL_MC"] = old
    assert res["0"][3] == res["8"][3] == [0] * nb
    assert res["0"][2] == res["8"][2]
    assert np.array_equal(res["0"][0], res["8"][0]) and np.array_equal(res["0"][1], res["8"][1])


def test_multi_cu_cholesky_hands_over_on_a_bad_pivot(ds):
    """Duplicated columns: J^T J is singular, the natural-order Cholesky meets a non-positive pivot and hands over to the
    pivoted kernel / QR (ST_NEED_PCHOL) -- same decision, same result in both forms."""
    A, b, xt, x0 = ds.generate(1, 1024, 256, seed0=99)
    A[:, 200:, :] = A[:, :56, :]
    res = {}
    old = os.environ.get("NLH_CHOL_MC")
    try:
        for mc in ("0", "8"):
            os.environ["NLH_CHOL_MC"] = mc
            x = x0.clone()
            fvec, ibs, status = ds.lm_solve_batch(A, b, 0.5, x, ds.options(max_evals=60, factor_policy=0))
            torch.cuda.synchronize()
            res[mc] = (x.cpu().numpy().copy(), [tuple(sorted(i.items())) for i in ibs], list(status))
    finally:
        if old is None:
            os.environ.pop("NLH_CHOL_MC", None)
        else:
            os.environ["NLH_CHOL_MC"] = old
    assert res["0"][1] == res["8"][1] and res["0"][2] == res["8"][2]
    assert np.array_equal(res["0"][0], res["8"][0])


@pytest.mark.parametrize("nb,m,n", [(1, 5000, 512), (2, 1000, 300), (1, 3100, 257), (2, 2048, 384), (1, 700, 400), (3, 1500, 511)])
def test_gram_512_same_bits_as_block_kernel(ds, nb, m, n):
    """k_gram_512 (256 < n <= 512: the two diagonal 256-column blocks and the two halves of the square between them as
    four workgroups per (problem, K-split)) against k_gram_mfma (64 x 64 blocks): the same accumulation order, hence
    the same bits of G (g to rounding); and both against a float64 reference."""
    g = torch.Generator(device="cpu").manual_seed(5)
    J = torch.randn((nb, n, m), dtype=torch.float64, generator=g).cuda()
    f = torch.randn((nb, m), dtype=torch.float64, generator=g).cuda()
    old = os.environ.get("NLH_GRAM512")
    try:
        os.environ["NLH_GRAM512"] = "0"
        G0, g0 = ds.gram(J, f)
        os.environ["NLH_GRAM512"] = "1"
        G1, g1 = ds.gram(J, f)
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("NLH_GRAM512", None)
        else:
            os.environ["NLH_GRAM512"] = old
    assert torch.equal(G0, G1)
    # g = J^T f: k_gram_512 adds sixteen partial sums per column and split (out of its loader's registers), the block kernel
    # four: the same vector to rounding, not to the bit
    assert float((g0 - g1).abs().max()) <= 1e-14 * float(g0.abs().max()) * np.sqrt(m)
    Gref = torch.matmul(J, J.transpose(1, 2))
    gref = torch.matmul(J, f.unsqueeze(-1)).squeeze(-1)
    assert float((G1 - Gref).abs().max()) <= 1e-13 * float(Gref.abs().max()) * np.sqrt(m)
    assert float((g1 - gref).abs().max()) <= 1e-13 * float(gref.abs().max()) * np.sqrt(m)
    assert torch.equal(G1, G1.transpose(1, 2))
