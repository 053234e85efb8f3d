"""Several GPUs behind the C boundary (nlh_device_set, nlh_dq_model_create_on): a model dealt block-cyclically over the
entries of a device set and solved by one host thread per entry must return, for every problem, the bits a single
handle returns.  A 1-GPU box exercises the dealing, the per-share threads and the gather with the degenerate set [0]
and with TWO shares on the same GPU ([0, 0]); on a multi-GPU node the same test spans the real devices."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("iter_count", "fcn_count", "jacobian_count", "converge_on_fcn", "converge_on_chng", "converge_on_zero_diff")


def _host_problem(ds, nprob, m, n, **kw):
    A, b, xt, x0 = ds.generate(nprob, m, n, seed0=777, **kw)
    return A.cpu().numpy(), b.cpu().numpy(), x0.cpu().numpy()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], None])
def test_lm_on_a_device_set_matches_one_handle_bitwise(ds, devices):
    from nonlin_amd.device import DeviceSet
    nprob, m, n = 11, 512, 48                              # 11 problems over 1 / 2 / 3 shares: ragged deal
    A, b, x0 = _host_problem(ds, nprob, m, n)
    ref = ds.model(A, b, 0.5)
    xr, fr, ibr, sr = ref.lm_solve(x0, ds.options(max_evals=200))
    dset = DeviceSet(devices)
    model = dset.model(A, b, 0.5)
    assert model.shares == len(dset)
    if devices is not None:
        assert len(dset) == len(devices)
    x, f, ib, st = model.lm_solve(x0, ds.options(max_evals=200))
    assert np.array_equal(x, xr) and np.array_equal(f, fr)
    assert st == sr
    for p in range(nprob):
        for k in KEYS:
            assert ib[p][k] == ibr[p][k], (p, k)
    # ... and the single-handle model agrees with the device-pointer entry point the rest of the suite tests
    At, bt, xt = (torch.tensor(v, device="cuda:0") for v in (A, b, x0))
    fv, ibs, status = ds.lm_solve_batch(At, bt, 0.5, xt, ds.options(max_evals=200))
    assert np.array_equal(xt.cpu().numpy(), xr)


def test_newton_on_a_device_set_matches_one_handle_bitwise(ds):
    from nonlin_amd.device import DeviceSet
    nprob, n = 7, 96
    A, b, x0 = _host_problem(ds, nprob, n, n, sigma=0.0, square_shift=True)
    xr, fr, ibr, sr = ds.model(A, b, 0.5).newton_solve(x0, True, ds.options(max_evals=200))
    x, f, ib, st = DeviceSet([0, 0]).model(A, b, 0.5).newton_solve(x0, True, ds.options(max_evals=200))
    assert np.array_equal(x, xr) and np.array_equal(f, fr) and st == sr
    assert [tuple(i[k] for k in KEYS) for i in ib] == [tuple(i[k] for k in KEYS) for i in ibr]


def test_more_shares_than_problems(ds):
    from nonlin_amd.device import DeviceSet
    A, b, x0 = _host_problem(ds, 2, 256, 16)
    xr, fr, ibr, sr = ds.model(A, b, 0.5).lm_solve(x0)
    x, f, ib, st = DeviceSet([0, 0, 0, 0]).model(A, b, 0.5).lm_solve(x0)      # two of the four shares are empty
    assert np.array_equal(x, xr) and np.array_equal(f, fr) and st == sr


def test_bad_device_id_is_refused():
    from nonlin_amd.device import DeviceSet
    from nonlin_amd import _lib
    with pytest.raises(_lib.NonlinHipUnavailable):
        DeviceSet([torch.cuda.device_count() + 3])


def test_model_level_quasi_newton_cls_bfgs_on_a_device_set(oracle):
    """nlh_dq_model_quasi_newton_solve / _cls_solve / _bfgs_solve on a model dealt over two shares (device 0 twice): the
    lock-step batches behind host arrays, every problem bit-identical to the CPU oracle."""
    import numpy as np
    from nonlin_amd.device import DeviceSet
    from nonlin_amd import _lib
    ds = DeviceSet([0, 0])
    o = _lib.default_options(); o.max_evals = 500
    gen = [oracle.dq_generate(900 + k, 40, 40, gamma=0.5, sigma=0.0, square_shift=True) for k in range(5)]
    A = np.stack([np.ascontiguousarray(g[0].T) for g in gen]); b = np.stack([g[1] for g in gen]); x0 = np.stack([g[3] for g in gen])
    md = ds.model(A, b, 0.5)
    x, f, ibs, st = md.quasi_newton_solve(x0, analytic=True, opts=o)
    for k, g in enumerate(gen):
        rc, xo, fo, ibo, _ = oracle.dq_quasi_newton_solve(g[0], g[1], 0.5, g[3], analytic=True, opts=oracle.default_options(max_evals=500))
        assert st[k] == rc and np.array_equal(x[k], xo) and np.array_equal(f[k], fo)
        assert ibs[k]["iter_count"] == ibo["iter_count"] and ibs[k]["fcn_count"] == ibo["fcn_count"]
    md.close()
    gen = [oracle.dq_generate(950 + k, 200, 24, gamma=0.5) for k in range(5)]
    A = np.stack([np.ascontiguousarray(g[0].T) for g in gen]); b = np.stack([g[1] for g in gen]); x0 = np.stack([g[3] for g in gen])
    md = ds.model(A, b, 0.5)
    lo, hi = np.full(24, -0.4), np.full(24, 0.3)
    x, f, ibs, st = md.cls_solve(x0, lower=lo, upper=hi, opts=o)
    for k, g in enumerate(gen):
        rc, xo, fo, ibo, _ = oracle.dq_cls_solve(g[0], g[1], 0.5, g[3], opts=oracle.default_options(max_evals=500), lower=lo, upper=hi)
        assert st[k] == rc and np.array_equal(x[k], xo) and np.array_equal(f[k], fo)
        assert ibs[k]["iter_count"] == ibo["iter_count"] and ibs[k]["jacobian_count"] == ibo["jacobian_count"]
    ob = dict(max_evals=300, gtol=1e-8, xtol=1e-12)
    o2 = _lib.default_options(); o2.max_evals = 300; o2.gtol = 1e-8; o2.xtol = 1e-12
    x, f, fo_, ibs, st = md.bfgs_solve(x0, opts=o2)
    for k, g in enumerate(gen):
        rc, xo, fo, ibo, _ = oracle.dq_bfgs_solve(g[0], g[1], 0.5, g[3], opts=oracle.default_options(**ob))
        assert st[k] == rc and np.array_equal(x[k], xo) and fo_[k] == fo
        assert np.array_equal(f[k], oracle.dq_residual(g[0], g[1], 0.5, xo))
        assert ibs[k]["gradient_count"] == ibo["gradient_count"]
    md.close()
    ds.close()


def test_long_columns_on_a_device_set(ds):
    """Columns of more than one 4096-row chunk on every share of a device set: the column sweep's two LDS product buffers
    (more than 64 KB of dynamic LDS) have to be allowed on each device a handle is created on."""
    from nonlin_amd.device import DeviceSet
    nprob, m, n = 3, 9000, 10
    A, b, x0 = _host_problem(ds, nprob, m, n)
    ref = ds.model(A, b, 0.5)
    xr, fr, ibr, sr = ref.lm_solve(x0, ds.options(max_evals=200))
    dset = DeviceSet(None)                                  # every visible device
    model = dset.model(A, b, 0.5)
    x, f, ib, st = model.lm_solve(x0, ds.options(max_evals=200))
    assert np.array_equal(x, xr) and np.array_equal(f, fr) and st == sr
    model.close(); ref.close(); dset.close()
