! Boundary object for the line search: the reference's public type `line_search` and the free
! routine `limit_search_vector` (public names of src/nonlin_linesearch.f90:18-65, :554-572).
! Nothing is searched on the Fortran side: ls_search_mimo / ls_search_miso run inside the C layer
! (nlh_newton_solve, nlh_quasi_newton_solve, nlh_bfgs_solve); this type only carries the three
! knobs those entry points take in nlh_options (ls_max_evals, ls_alpha, ls_factor).
module nonlin_linesearch
    use, intrinsic :: iso_fortran_env, only : int32, real64
    implicit none
    private
    public :: line_search
    public :: limit_search_vector

    ! Knob indices of the record below.
    integer, parameter :: KNOB_ALPHA = 1      ! sufficient-decrease constant (reference default 1e-4)
    integer, parameter :: KNOB_SHRINK = 2     ! smallest allowed ratio of successive step lengths (0.1)

    type line_search
        integer(int32), private :: budget_ = 100
        real(real64), private :: knob_(2) = [1.0d-4, 0.1d0]
    contains
        procedure, public :: get_max_fcn_evals => search_budget
        procedure, public :: set_max_fcn_evals => search_put_budget
        procedure, public :: get_scaling_factor => search_alpha
        procedure, public :: set_scaling_factor => search_put_alpha
        procedure, public :: get_distance_factor => search_shrink
        procedure, public :: set_distance_factor => search_put_shrink
    end type

contains
    !> v if it lies strictly inside (lo, hi); otherwise the replacement the reference's setter stores
    !> for that side (the distance factor: src/nonlin_linesearch.f90:142-148).
    pure elemental function inside_or(v, lo, hi, below, above) result(r)
        real(real64), intent(in) :: v, lo, hi, below, above
        real(real64) :: r
        r = merge(below, merge(above, v, v >= hi), v <= lo)
    end function

    pure integer(int32) function search_budget(this)
        class(line_search), intent(in) :: this
        search_budget = this%budget_
    end function

    subroutine search_put_budget(this, x)
        class(line_search), intent(inout) :: this
        integer(int32), intent(in) :: x
        this%budget_ = x
    end subroutine

    pure real(real64) function search_alpha(this)
        class(line_search), intent(in) :: this
        search_alpha = this%knob_(KNOB_ALPHA)
    end function

    subroutine search_put_alpha(this, x)
        class(line_search), intent(inout) :: this
        real(real64), intent(in) :: x
        this%knob_(KNOB_ALPHA) = x
    end subroutine

    pure real(real64) function search_shrink(this)
        class(line_search), intent(in) :: this
        search_shrink = this%knob_(KNOB_SHRINK)
    end function

    subroutine search_put_shrink(this, x)
        class(line_search), intent(inout) :: this
        real(real64), intent(in) :: x
        this%knob_(KNOB_SHRINK) = inside_or(x, 0.0d0, 1.0d0, 0.1d0, 0.99d0)
    end subroutine

    !> Shortens x to Euclidean length lim when it is longer (a zero vector is left alone).
    subroutine limit_search_vector(x, lim)
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(in) :: lim
        real(real64) :: length
        length = norm2(x)
        if (length > lim .and. length /= 0.0d0) x = (lim / length) * x
    end subroutine
end module
