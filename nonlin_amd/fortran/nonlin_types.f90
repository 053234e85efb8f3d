! Same public names and component names as the reference module (src/nonlin_types.f90:8-37); nlh_iteration_behavior in
! include/nonlin_hip.h is the C-side record these counters and flags are filled from (nonlin_shim_support.f90).
module nonlin_types
    use iso_fortran_env, only : int32, real64
    implicit none
    private
    public :: iteration_behavior, value_pair

    type iteration_behavior
        integer(int32) :: iter_count, fcn_count, jacobian_count, gradient_count   ! counters reported by every solve
        logical :: converge_on_fcn, converge_on_chng, converge_on_zero_diff       ! which test stopped the iteration
    end type iteration_behavior

    type value_pair      ! a bracket [x1, x2]
        real(real64) :: x1, x2
    end type value_pair
end module nonlin_types
