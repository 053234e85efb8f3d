! Same public names as the reference module (src/nonlin_types.f90:8-37).
module nonlin_types
    use iso_fortran_env
    implicit none
    private
    public :: iteration_behavior
    public :: value_pair

    type iteration_behavior
        integer(int32) :: iter_count, fcn_count, jacobian_count, gradient_count   ! counters reported by every solve
        logical :: converge_on_fcn, converge_on_chng, converge_on_zero_diff       ! which test stopped the iteration
    end type

    type value_pair
        real(real64) :: x1, x2
    end type
end module
