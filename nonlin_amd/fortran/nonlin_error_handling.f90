! NL_* codes (src/nonlin_error_handling.f90:10-38).  The reference aliases three of them to
! LA_* parameters of the un-vendored linalg_errors module; their values (105/104/106) are
! written out here because that module is not part of this build.
module nonlin_error_handling
    use iso_fortran_env
    implicit none
    integer(int32), parameter :: NL_NO_ERROR = 0
    integer(int32), parameter :: NL_INVALID_INPUT_ERROR = 201
    integer(int32), parameter :: NL_ARRAY_SIZE_ERROR = 202
    integer(int32), parameter :: NL_OUT_OF_MEMORY_ERROR = 105
    integer(int32), parameter :: NL_INVALID_OPERATION_ERROR = 104
    integer(int32), parameter :: NL_CONVERGENCE_ERROR = 106
    integer(int32), parameter :: NL_DIVERGENT_BEHAVIOR_ERROR = 206
    integer(int32), parameter :: NL_SPURIOUS_CONVERGENCE_ERROR = 207
    integer(int32), parameter :: NL_TOLERANCE_TOO_SMALL_ERROR = 208
    integer(int32), parameter :: NL_INDEX_OUT_OF_RANGE_ERROR = 209
    integer(int32), parameter :: NL_DIVIDE_BY_ZERO_ERROR = 210
    integer(int32), parameter :: NL_UNDEFINED_FUNCTION_ERROR = 211
    integer(int32), parameter :: NL_UNDERDEFINED_PROBLEM_ERROR = 212
end module
