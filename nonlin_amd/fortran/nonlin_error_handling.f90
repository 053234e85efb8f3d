! NL_* codes (src/nonlin_error_handling.f90:10-38): same public names, same values.  The reference aliases three of them to
! LA_* parameters of the un-vendored linalg_errors module; their values (104/105/106) are written out here because that
! module is not part of this build.  The C ABI reports the same numbers (include/nonlin_hip.h, NLH_*_ERROR).
module nonlin_error_handling
    use iso_fortran_env, only : int32
    implicit none
    public
    integer(int32), parameter :: NL_NO_ERROR = 0
    ! raised by argument checks before anything reaches the device
    integer(int32), parameter :: NL_INVALID_INPUT_ERROR = 201, NL_ARRAY_SIZE_ERROR = 202, &
        NL_INDEX_OUT_OF_RANGE_ERROR = 209, NL_UNDEFINED_FUNCTION_ERROR = 211, NL_UNDERDEFINED_PROBLEM_ERROR = 212
    ! shared with linalg (LA_INVALID_OPERATION_ERROR, LA_OUT_OF_MEMORY_ERROR, LA_CONVERGENCE_ERROR)
    integer(int32), parameter :: NL_INVALID_OPERATION_ERROR = 104, NL_OUT_OF_MEMORY_ERROR = 105, NL_CONVERGENCE_ERROR = 106
    ! outcomes of an iteration
    integer(int32), parameter :: NL_DIVERGENT_BEHAVIOR_ERROR = 206, NL_SPURIOUS_CONVERGENCE_ERROR = 207, &
        NL_TOLERANCE_TOO_SMALL_ERROR = 208, NL_DIVIDE_BY_ZERO_ERROR = 210
end module
