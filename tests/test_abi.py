"""CPU tests of the drop-in boundary: libnonlin_hip.so loads, exports every symbol that
include/nonlin_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "nonlin_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nlh_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from nonlin_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert set(_lib.SYMBOLS) == set(names)          # the ctypes table and the header agree


def test_struct_layouts_match_header():
    from nonlin_amd import _lib
    assert C.sizeof(_lib.IterationBehavior) == 28    # 4 x int32 + 3 x logical (src/nonlin_types.f90:8-29)
    o = _lib.default_options()
    assert (o.max_evals, o.ftol, o.xtol, o.gtol, o.print_status) == (100, 1e-8, 1e-12, 1e-12, 0)
    assert (o.factor, o.use_line_search, o.ls_max_evals, o.ls_alpha, o.ls_factor) == (100.0, 1, 100, 1e-4, 0.1)


def test_no_cpu_fallback():
    """Without a GPU the product path fails loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nonlin_amd as nl
    from nonlin_amd import _lib
    assert _lib.load().nlh_device_count() == 0
    obj = nl.vecfcn_helper()
    obj.set_fcn(lambda x, f, a: None, 2, 2)
    with pytest.raises(nl.NonlinHipUnavailable):
        nl.least_squares_solver().solve(obj, np.ones(2), np.zeros(2))
    with pytest.raises(nl.NonlinHipUnavailable):
        nl.newton_solver().solve(obj, np.ones(2), np.zeros(2))
    from nonlin_amd.device import DeviceSet
    with pytest.raises(nl.NonlinHipUnavailable):          # several GPUs behind the boundary: same rule
        DeviceSet()
    import ctypes as C
    set_ptr = C.c_void_p()
    assert _lib.load().nlh_device_set_create(C.byref(set_ptr), None, 0) == -1 and not set_ptr.value   # NLH_ERR_NO_DEVICE
    x, g = np.array([3.0, 0.0]), np.zeros(2)             # the one entry point with no device work: host FD gradient
    f = _lib.FCNNVAR(lambda ctx, n, xx: xx[0] * xx[0] + 2.0 * xx[1])
    assert _lib.load().nlh_fd_gradient(2, f, _lib.GRADFCN(), None, x.ctypes.data_as(_lib.c_double_p), None,
                                       g.ctypes.data_as(_lib.c_double_p)) == 0
    assert abs(g[0] - 6.0) < 1e-6 and abs(g[1] - 2.0) < 1e-6 and x[0] == 3.0 and x[1] == 0.0
    from nonlin_amd.device import DeviceSolver
    with pytest.raises(nl.NonlinHipUnavailable):
        DeviceSolver(0)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nonlin_amd")):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".f90")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, fn)
    hdr = open(os.path.join(ROOT, "include", "nonlin_hip.h")).read()
    assert "oracle" not in hdr.lower()


def test_host_api_mirrors_reference_defaults_and_clamps():
    import nonlin_amd as nl
    s = nl.least_squares_solver()
    assert (s.get_max_fcn_evals(), s.get_fcn_tolerance(), s.get_var_tolerance(), s.get_gradient_tolerance(),
            s.get_print_status(), s.get_step_scaling_factor()) == (100, 1e-8, 1e-12, 1e-12, False, 100.0)
    s.set_step_scaling_factor(1e-3); assert s.get_step_scaling_factor() == 0.1      # :108-114
    s.set_step_scaling_factor(1e3); assert s.get_step_scaling_factor() == 100.0
    ls = nl.line_search()
    assert (ls.get_max_fcn_evals(), ls.get_scaling_factor(), ls.get_distance_factor()) == (100, 1e-4, 0.1)
    ls.set_distance_factor(-1.0); assert ls.get_distance_factor() == 0.1             # linesearch.f90:142-148
    ls.set_distance_factor(2.0); assert ls.get_distance_factor() == 0.99
    ns = nl.newton_solver()
    assert ns.get_use_line_search() and not ns.is_line_search_defined()
    ns.set_default_line_search(); assert ns.is_line_search_defined()
    got = ns.get_line_search(); got.set_max_fcn_evals(7)                            # a copy (solve.f90:98-99)
    assert ns.get_line_search().get_max_fcn_evals() == 100
    h = nl.vecfcn_helper()
    assert not h.is_fcn_defined() and not h.is_jacobian_defined()
    h.set_fcn(lambda x, f, a: None, 21, 4)
    assert (h.get_equation_count(), h.get_variable_count(), h.is_fcn_defined()) == (21, 4, True)
    assert (nl.NL_CONVERGENCE_ERROR, nl.NL_UNDEFINED_FUNCTION_ERROR, nl.NL_UNDERDEFINED_PROBLEM_ERROR) == (106, 211, 212)


def test_print_status_text_matches_fortran_formats():
    """print_status (src/nonlin_helper.f90:17-33) prints A,I0 and A,E10.3; tests/golden/e10_3_flang.txt is what amdflang's
    runtime prints for those edit descriptors (generated by tests/golden/make_e10_3.f90, values listed there)."""
    from nonlin_amd import _lib
    lib = _lib.load()
    vals = [0.0, 1.23, -2.5, 1.23456e-4, 0.9996, 0.99949, 9.995, 1.0e100, 1.0e-100, 12345.678,
            4.44089209850063e-16, 1.28602518018146, 0.506363030790737, 1.0, 0.1, 99.95, -1.0e-7, 5.0e-324]
    golden = open(os.path.join(ROOT, "tests", "golden", "e10_3_flang.txt")).read().split("\n")
    buf = C.create_string_buffer(256)
    for k, v in enumerate(vals):
        n = lib.nlh_format_status(12, 34, 5, v, 1.0, buf, 256)
        lines = buf.value.decode().split("\n")
        assert n == len(buf.value)
        assert lines[0] == " " == golden[len(vals)]                       # print *, ""
        assert lines[1] == "Iteration: 12" == golden[len(vals) + 1]       # A, I0
        assert lines[2] == "Function Evaluations: 34" and lines[3] == "Jacobian Evaluations: 5"
        assert lines[4] == golden[k], (v, lines[4], golden[k])            # A, E10.3
        assert lines[5] == "Residual:  0.100E+01"
    lib.nlh_format_status(1, 2, 0, 1.0, 2.0, buf, 256)                    # no Jacobian line when the count is zero (:27)
    assert "Jacobian" not in buf.value.decode()
