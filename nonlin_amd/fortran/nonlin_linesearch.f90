! line_search configuration object and limit_search_vector with the reference's names
! (src/nonlin_linesearch.f90:18-65, 71-149, 554-572).  The search itself (ls_search_mimo) runs
! inside nlh_newton_solve; this type carries its three parameters across the boundary.
module nonlin_linesearch
    use, intrinsic :: iso_fortran_env, only : int32, real64
    implicit none
    private
    public :: line_search
    public :: limit_search_vector

    type line_search
        integer(int32), private :: max_evals_ = 100
        real(real64), private :: m_alpha = 1.0d-4
        real(real64), private :: m_factor = 0.1d0
    contains
        procedure, public :: get_max_fcn_evals => ls_get_max_eval
        procedure, public :: set_max_fcn_evals => ls_set_max_eval
        procedure, public :: get_scaling_factor => ls_get_scale
        procedure, public :: set_scaling_factor => ls_set_scale
        procedure, public :: get_distance_factor => ls_get_dist
        procedure, public :: set_distance_factor => ls_set_dist
    end type

contains
    pure function ls_get_max_eval(this) result(n)
        class(line_search), intent(in) :: this
        integer(int32) :: n
        n = this%max_evals_
    end function

    subroutine ls_set_max_eval(this, x)
        class(line_search), intent(inout) :: this
        integer(int32), intent(in) :: x
        this%max_evals_ = x
    end subroutine

    pure function ls_get_scale(this) result(x)
        class(line_search), intent(in) :: this
        real(real64) :: x
        x = this%m_alpha
    end function

    subroutine ls_set_scale(this, x)
        class(line_search), intent(inout) :: this
        real(real64), intent(in) :: x
        this%m_alpha = x
    end subroutine

    pure function ls_get_dist(this) result(x)
        class(line_search), intent(in) :: this
        real(real64) :: x
        x = this%m_factor
    end function

    subroutine ls_set_dist(this, x)     ! clamp: src/nonlin_linesearch.f90:142-148
        class(line_search), intent(inout) :: this
        real(real64), intent(in) :: x
        if (x <= 0.0d0) then
            this%m_factor = 0.1d0
        else if (x >= 1.0d0) then
            this%m_factor = 0.99d0
        else
            this%m_factor = x
        end if
    end subroutine

    subroutine limit_search_vector(x, lim)
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(in) :: lim
        real(real64) :: mag
        mag = norm2(x)
        if (mag == 0.0d0) return
        if (mag > lim) x = (lim / mag) * x
    end subroutine
end module
