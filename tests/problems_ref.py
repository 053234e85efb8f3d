"""The reference's own test / example problems for the hot path, transcribed as callbacks of
the form fcn(x, f, args) / jac(x, J, args) (J is an m x n Fortran-order view).

Sources (problem definitions and data only):
  fcn1, jac1, fcn1a, jac1a, fcn2, lsfcn1   tests/nonlin_test_solve.f90:41-159
  polar*, polar_scaled*                    tests/nonlin_test_jacobian.f90:20-85
  powell*                                  tests/powell_badly_scaled.f90:9-38
  misc01*                                  examples/example_problems.f90:71-91
"""
import math

import numpy as np


# Fortran evaluates x**2 as x*x; Python's ** goes through pow(), which is not guaranteed to be
# correctly rounded, so squares are written as products here.
def fcn1(x, f, args=None):
    f[0] = x[0] * x[0] + x[1] * x[1] - 34.0
    f[1] = x[0] * x[0] - 2.0 * (x[1] * x[1]) - 7.0


def jac1(x, J, args=None):
    J[0, 0] = 2.0 * x[0]
    J[1, 0] = 2.0 * x[0]
    J[0, 1] = 2.0 * x[1]
    J[1, 1] = 2.0 * (-2.0 * x[1])


def fcn1a(x, f, args=None):
    a = float(args)
    f[0] = x[0] * x[0] + x[1] * x[1] - 34.0
    f[1] = x[0] * x[0] - a * (x[1] * x[1]) - 7.0


def jac1a(x, J, args=None):
    a = float(args)
    J[0, 0] = 2.0 * x[0]
    J[1, 0] = 2.0 * x[0]
    J[0, 1] = 2.0 * x[1]
    J[1, 1] = 2.0 * (-a * x[1])


def fcn2(x, f, args=None):
    f[0] = x[1] - 10.0
    f[1] = x[0] * x[1] - 5e4      # the reference writes the single-precision literal 5e4 (exact)


XP = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.7,
               1.8, 1.9, 2.0])
YP = np.array([1.216737514, 1.250032542, 1.305579195, 1.040182335, 1.751867738, 1.109716707, 2.018141531,
               1.992418729, 1.807916923, 2.078806005, 2.698801324, 2.644662712, 3.412756702, 4.406137221,
               4.567156645, 4.999550779, 5.652854194, 6.784320119, 8.307936836, 8.395126494, 10.30252404])


def lsfcn1(x, f, args=None):
    """21 x 4 cubic fit = README Example 2 (README.md:139-159) = BASELINE config 1."""
    f[:] = x[0] * (XP * XP * XP) + x[1] * (XP * XP) + x[2] * XP + x[3] - YP


def polar(x, f, args=None):
    f[0] = x[0] * math.cos(x[1])
    f[1] = x[0] * math.sin(x[1])


def polar_jac(x, J, args=None):
    r, th = x[0], x[1]
    J[0, 0] = math.cos(th)
    J[1, 0] = math.sin(th)
    J[0, 1] = -r * math.sin(th)
    J[1, 1] = r * math.cos(th)


def polar_scaled(x, f, args=None):
    y = float(args)
    f[0] = y * x[0] * math.cos(x[1])
    f[1] = y * x[0] * math.sin(x[1])


def polar_scaled_jac(x, J, args=None):
    y = float(args)
    r, th = x[0], x[1]
    J[0, 0] = y * math.cos(th)
    J[1, 0] = y * math.sin(th)
    J[0, 1] = -y * r * math.sin(th)
    J[1, 1] = y * r * math.cos(th)


def powell(x, f, args=None):
    f[0] = 1.0e4 * x[0] * x[1] - 1.0
    f[1] = math.exp(-x[0]) + math.exp(-x[1]) - 1.0001


def powell_jac(x, J, args=None):
    J[0, 0] = 1.0e4 * x[1]
    J[1, 0] = -math.exp(-x[0])
    J[0, 1] = 1.0e4 * x[0]
    J[1, 1] = -math.exp(-x[1])


def misc01(x, f, args=None):
    f[0] = 2.0 * x[0] - x[1] - math.exp(-x[0])
    f[1] = -x[0] + 2.0 * x[1] - math.exp(-x[1])


def misc01_jac(x, J, args=None):
    J[0, 0] = math.exp(-x[0]) + 2.0
    J[1, 0] = -1.0
    J[0, 1] = -1.0
    J[1, 1] = math.exp(-x[1]) + 2.0
