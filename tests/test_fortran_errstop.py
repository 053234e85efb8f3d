"""CPU-side check of the Fortran shim's polynomial error stops (no GPU needed: the program stops before any device call).
src/nonlin_polynomials.f90:399: `get` on a polynomial that was never initialised stops with NL_INVALID_OPERATION_ERROR;
:402-405: an index out of range stops with NL_INDEX_OUT_OF_RANGE_ERROR; `set` on an uninitialised polynomial returns (:436)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(HERE, "fortran", "build", "dropin_suite")


def _exe():
    if not os.path.exists(EXE):
        if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
            pytest.skip("no Fortran compiler and no prebuilt tests/fortran/build/dropin_suite")
        if not os.path.exists(os.path.join(ROOT, "nonlin_amd", "libnonlin_hip.so")):
            pytest.skip("libnonlin_hip.so not built")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "nonlin_amd", "fortran"), "-s"])
        subprocess.check_call(["make", "-C", os.path.join(HERE, "fortran"), "-s"])
    return EXE


def test_get_on_uninitialised_polynomial_stops_with_invalid_operation():
    out = subprocess.run([_exe(), "errstop_poly_get"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 104, (out.returncode, out.stderr)      # NL_INVALID_OPERATION_ERROR (nonlin_error_handling.f90)


def test_index_out_of_range_stops_and_accessors_of_the_zero_polynomial():
    out = subprocess.run([_exe(), "errstop_poly_index"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 209, (out.returncode, out.stderr)      # NL_INDEX_OUT_OF_RANGE_ERROR
    assert out.stdout.split()[:3] == ["2", "0.", "3"]                # order, value of the zero polynomial, coefficient count
