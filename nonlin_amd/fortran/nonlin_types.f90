! Same public names as the reference module (src/nonlin_types.f90:8-37).
module nonlin_types
    use iso_fortran_env
    implicit none
    private
    public :: iteration_behavior
    public :: value_pair

    type iteration_behavior
        integer(int32) :: iter_count
        integer(int32) :: fcn_count
        integer(int32) :: jacobian_count
        integer(int32) :: gradient_count
        logical :: converge_on_fcn
        logical :: converge_on_chng
        logical :: converge_on_zero_diff
    end type

    type value_pair
        real(real64) :: x1
        real(real64) :: x2
    end type
end module
