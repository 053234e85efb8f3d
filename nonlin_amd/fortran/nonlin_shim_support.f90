! Marshalling helpers shared by every solve body of the shim (no counterpart in the reference, whose solvers
! compute in place): conversion of the C ABI's counters record into iteration_behavior, the size checks every
! `solve` performs before it crosses the boundary, and the two clamp rules the reference's setters use.
module nonlin_shim_support
    use, intrinsic :: iso_fortran_env, only : int32, real64
    use, intrinsic :: iso_c_binding
    use nonlin_types
    use nonlin_hip_c
    implicit none
    private
    public :: behavior_clear
    public :: behavior_import
    public :: require_vector_sizes
    public :: into_interval
    public :: positive_or

contains
    !> The state `solve` leaves in ib before anything has run.
    elemental subroutine behavior_clear(ib)
        type(iteration_behavior), intent(out) :: ib
        ib = iteration_behavior(0, 0, 0, 0, .false., .false., .false.)
    end subroutine

    !> C record (int32 flags) -> the reference's derived type (default logicals).
    elemental subroutine behavior_import(ib, c)
        type(iteration_behavior), intent(inout) :: ib
        type(nlh_iteration_behavior), intent(in) :: c
        ib%iter_count = c%iter_count
        ib%fcn_count = c%fcn_count
        ib%jacobian_count = c%jacobian_count
        ib%gradient_count = c%gradient_count
        ib%converge_on_fcn = (c%converge_on_fcn /= 0)
        ib%converge_on_chng = (c%converge_on_chng /= 0)
        ib%converge_on_zero_diff = (c%converge_on_zero_diff /= 0)
    end subroutine

    !> The reference stops with 3 when x has the wrong length and with 4 when fvec has
    !> (e.g. src/nonlin_least_squares.f90:191-196, src/nonlin_solve.f90:520-525).
    subroutine require_vector_sizes(nx, nf, nvar, neqn)
        integer(int32), intent(in) :: nx, nf, nvar, neqn
        if (nx /= nvar) error stop 3
        if (nf /= neqn) error stop 4
    end subroutine

    !> v limited to [lo, hi].
    pure elemental real(real64) function into_interval(v, lo, hi)
        real(real64), intent(in) :: v, lo, hi
        into_interval = min(max(v, lo), hi)
    end function

    !> v when it is positive, the fallback otherwise.
    pure elemental real(real64) function positive_or(v, fallback)
        real(real64), intent(in) :: v, fallback
        positive_or = merge(v, fallback, v > 0.0d0)
    end function
end module
