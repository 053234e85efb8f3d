! polynomial with the reference's public interface for the fitting front end
! (src/nonlin_polynomials.f90:39-62: initialize, order, fit, fit_thru_zero, evaluate (real), get, get_all, set);
! fit / fit_thru_zero marshal to nlh_poly_fit (Vandermonde panel + Householder QR + back substitution on the GPU,
! :146-238).  Roots, the companion matrix and polynomial arithmetic are outside the hot path and not provided.
module nonlin_polynomials
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_hip_c
    implicit none
    private
    public :: polynomial

    type polynomial
        real(real64), private, allocatable, dimension(:) :: m_coeffs
    contains
        generic, public :: initialize => init_poly, init_poly_coeffs
        procedure, public :: order => get_poly_order
        procedure, public :: fit => poly_fit
        procedure, public :: fit_thru_zero => poly_fit_thru_zero
        generic, public :: evaluate => evaluate_real
        procedure, public :: get => get_poly_coefficient
        procedure, public :: get_all => get_poly_coefficients
        procedure, public :: set => set_poly_coefficient
        procedure, private :: evaluate_real => poly_eval_double
        procedure, private :: init_poly
        procedure, private :: init_poly_coeffs
    end type

contains
    pure subroutine init_poly(this, order)                  ! :69-90
        class(polynomial), intent(inout) :: this
        integer(int32), intent(in) :: order
        if (order < 0) error stop 2
        if (allocated(this%m_coeffs)) deallocate(this%m_coeffs)
        allocate(this%m_coeffs(order + 1))
        this%m_coeffs = 0.0d0
    end subroutine

    pure subroutine init_poly_coeffs(this, c)               ! :93-109
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: c
        call init_poly(this, size(c) - 1)
        this%m_coeffs = c
    end subroutine

    pure function get_poly_order(this) result(n)            ! :112-143
        class(polynomial), intent(in) :: this
        integer(int32) :: n
        if (.not.allocated(this%m_coeffs)) then
            n = -1
        else
            n = size(this%m_coeffs) - 1
        end if
    end function

    subroutine poly_fit_impl(this, x, y, order, thru_zero)
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order, thru_zero
        integer(c_int) :: rc
        real(c_double), allocatable :: xc(:), yc(:), cc(:)
        if (size(y) /= size(x)) error stop 3                ! :159-162
        if (order >= size(x) .or. order < 1) error stop 4   ! :163-166
        if (this%order() /= order) call this%initialize(order)
        allocate(xc(size(x)), yc(size(x)), cc(order + 1))
        xc = x
        yc = y
        rc = nlh_poly_fit(nlh_default_handle(), int(size(x), c_int32_t), order, thru_zero, xc, yc, cc)
        if (rc /= 0) error stop rc
        this%m_coeffs = cc
    end subroutine

    subroutine poly_fit(this, x, y, order)                  ! :146-190
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order
        call poly_fit_impl(this, x, y, order, 0)
    end subroutine

    subroutine poly_fit_thru_zero(this, x, y, order)        ! :193-238
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order
        call poly_fit_impl(this, x, y, order, 1)
    end subroutine

    pure elemental function poly_eval_double(this, x) result(y)     ! :241-268
        class(polynomial), intent(in) :: this
        real(real64), intent(in) :: x
        real(real64) :: y
        integer(int32) :: j, order, n
        order = this%order()
        n = order + 1
        if (order == -1) then
            y = 0.0d0
            return
        else if (order == 0) then
            y = this%m_coeffs(1)
            return
        end if
        y = this%m_coeffs(n) * x + this%m_coeffs(order)
        do j = n - 2, 1, -1
            y = y * x + this%m_coeffs(j)
        end do
    end function

    pure function get_poly_coefficient(this, ind) result(c)
        class(polynomial), intent(in) :: this
        integer(int32), intent(in) :: ind
        real(real64) :: c
        c = 0.0d0
        if (.not.allocated(this%m_coeffs)) return
        if (ind <= 0 .or. ind > size(this%m_coeffs)) error stop 209     ! NL_INDEX_OUT_OF_RANGE_ERROR
        c = this%m_coeffs(ind)
    end function

    pure function get_poly_coefficients(this) result(c)
        class(polynomial), intent(in) :: this
        real(real64), allocatable, dimension(:) :: c
        if (allocated(this%m_coeffs)) then
            c = this%m_coeffs
        else
            allocate(c(0))
        end if
    end function

    pure subroutine set_poly_coefficient(this, ind, c)
        class(polynomial), intent(inout) :: this
        integer(int32), intent(in) :: ind
        real(real64), intent(in) :: c
        if (.not.allocated(this%m_coeffs)) return
        if (ind <= 0 .or. ind > size(this%m_coeffs)) error stop 209
        this%m_coeffs(ind) = c
    end subroutine
end module
