! `polynomial` as far as the fitting front end needs it (public surface of src/nonlin_polynomials.f90:39-62:
! initialize, order, fit, fit_thru_zero, evaluate (real argument), get, get_all, set).  The two fits marshal to
! nlh_poly_fit -- Vandermonde panel, Householder QR and back substitution on the GPU (:146-238).  Roots, the companion
! matrix and polynomial arithmetic are outside the hot path and not provided.
!
! Representation: `cf(0:deg)`, cf(k) multiplying x**k; an object that was never initialised has no `cf` and reports
! order -1 exactly as the reference does (:112-143).
module nonlin_polynomials
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_error_handling, only : NL_INVALID_OPERATION_ERROR, NL_INDEX_OUT_OF_RANGE_ERROR
    use nonlin_hip_c
    implicit none
    private
    public :: polynomial

    type polynomial
        real(real64), private, allocatable :: cf(:)          ! cf(0:deg)
    contains
        generic, public :: initialize => pl_alloc, pl_from_coefs
        procedure, public :: order => pl_degree
        procedure, public :: fit => pl_fit_free
        procedure, public :: fit_thru_zero => pl_fit_origin
        generic, public :: evaluate => pl_horner
        procedure, public :: get => pl_coef
        procedure, public :: get_all => pl_coefs
        procedure, public :: set => pl_put
        procedure, private :: pl_horner
        procedure, private :: pl_alloc
        procedure, private :: pl_from_coefs
    end type

contains
    ! zero polynomial of the given order (:69-90; a negative order is the reference's NL_INVALID_INPUT_ERROR there,
    ! reported here by the same small integer the shim uses for size errors)
    pure subroutine pl_alloc(this, order)
        class(polynomial), intent(inout) :: this
        integer(int32), intent(in) :: order
        if (order < 0) error stop 2
        if (allocated(this%cf)) deallocate(this%cf)
        allocate(this%cf(0:order), source = 0.0d0)
    end subroutine

    ! from a coefficient array, lowest power first (:93-109)
    pure subroutine pl_from_coefs(this, c)
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: c
        if (size(c) < 1) error stop 2
        if (allocated(this%cf)) deallocate(this%cf)
        allocate(this%cf(0:size(c) - 1))
        this%cf(0:) = c
    end subroutine

    pure integer(int32) function pl_degree(this) result(deg)
        class(polynomial), intent(in) :: this
        deg = -1
        if (allocated(this%cf)) deg = ubound(this%cf, 1)
    end function

    ! both fits: size checks of :159-166 / :206-213, then the device
    subroutine pl_fit_front(this, x, y, order, origin)
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order
        logical, intent(in) :: origin
        integer(c_int) :: rc
        integer(c_int32_t) :: npts
        real(c_double), allocatable :: xs(:), ys(:), sol(:)
        npts = int(size(x), c_int32_t)
        if (size(y) /= npts) error stop 3
        if (order < 1 .or. order >= npts) error stop 4
        allocate(xs(npts), source = x)
        allocate(ys(npts), source = y)
        allocate(sol(0:order))
        rc = nlh_poly_fit(nlh_default_handle(), npts, order, merge(1, 0, origin), xs, ys, sol)
        if (rc /= 0) error stop rc
        if (pl_degree(this) /= order) call pl_alloc(this, order)
        this%cf = sol
    end subroutine

    subroutine pl_fit_free(this, x, y, order)               ! :146-190
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order
        call pl_fit_front(this, x, y, order, .false.)
    end subroutine

    subroutine pl_fit_origin(this, x, y, order)             ! :193-238
        class(polynomial), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(inout), dimension(:) :: y
        integer(int32), intent(in) :: order
        call pl_fit_front(this, x, y, order, .true.)
    end subroutine

    ! Horner from the leading coefficient down: the operations of :241-268 in the same order (its first line is this
    ! loop's first trip), 0 for an uninitialised object
    pure elemental function pl_horner(this, x) result(y)
        class(polynomial), intent(in) :: this
        real(real64), intent(in) :: x
        real(real64) :: y
        integer(int32) :: k
        y = 0.0d0
        if (.not. allocated(this%cf)) return
        y = this%cf(ubound(this%cf, 1))
        do k = ubound(this%cf, 1) - 1, 0, -1
            y = y * x + this%cf(k)
        end do
    end function

    ! coefficient `ind` (1-based: c(1) + c(2) x + ...).  Asking an uninitialised polynomial for a coefficient is an
    ! invalid operation in the reference (:399) -- it stops; so does this.
    pure function pl_coef(this, ind) result(c)
        class(polynomial), intent(in) :: this
        integer(int32), intent(in) :: ind
        real(real64) :: c
        if (.not. allocated(this%cf)) error stop NL_INVALID_OPERATION_ERROR
        if (ind < 1 .or. ind > size(this%cf)) error stop NL_INDEX_OUT_OF_RANGE_ERROR
        c = this%cf(ind - 1)
    end function

    pure function pl_coefs(this) result(c)
        class(polynomial), intent(in) :: this
        real(real64), allocatable, dimension(:) :: c
        if (allocated(this%cf)) then
            allocate(c(size(this%cf)))
            c = this%cf
        else
            allocate(c(0))
        end if
    end function

    ! (:426-447: silently ignored on an uninitialised object, index checked otherwise)
    pure subroutine pl_put(this, ind, c)
        class(polynomial), intent(inout) :: this
        integer(int32), intent(in) :: ind
        real(real64), intent(in) :: c
        if (.not. allocated(this%cf)) return
        if (ind < 1 .or. ind > size(this%cf)) error stop NL_INDEX_OUT_OF_RANGE_ERROR
        this%cf(ind - 1) = c
    end subroutine
end module
