! line_search_optimizer and bfgs with the reference's public interface (src/nonlin_optimize.f90:44-72, 470-556);
! bfgs%solve marshals to nlh_bfgs_solve (bfgs_solve on the GPU, :557-770).  nelder_mead is outside the hot path.
module nonlin_optimize
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_linesearch
    use nonlin_error_handling
    use nonlin_multi_var
    use nonlin_types
    use nonlin_hip_c
    implicit none
    private
    public :: line_search_optimizer
    public :: bfgs

    type, abstract, extends(equation_optimizer) :: line_search_optimizer
        class(line_search), private, allocatable :: m_lineSearch
        logical, private :: m_useLineSearch = .true.
        real(real64), private :: xtol_ = 1.0d-12
    contains
        procedure, public :: get_line_search => lso_get_line_search
        procedure, public :: set_line_search => lso_set_line_search
        procedure, public :: set_default_line_search => lso_set_default
        procedure, public :: is_line_search_defined => lso_is_line_search_defined
        procedure, public :: get_use_line_search => lso_get_use_search
        procedure, public :: set_use_line_search => lso_set_use_search
        procedure, public :: get_var_tolerance => lso_get_var_tol
        procedure, public :: set_var_tolerance => lso_set_var_tol
    end type

    type, extends(line_search_optimizer) :: bfgs
    contains
        procedure, public :: solve => bfgs_solve
    end type

contains
    subroutine lso_get_line_search(this, ls)
        class(line_search_optimizer), intent(in) :: this
        class(line_search), intent(out), allocatable :: ls
        if (allocated(this%m_lineSearch)) allocate(ls, source = this%m_lineSearch)
    end subroutine

    subroutine lso_set_line_search(this, ls)
        class(line_search_optimizer), intent(inout) :: this
        class(line_search), intent(in) :: ls
        if (allocated(this%m_lineSearch)) deallocate(this%m_lineSearch)
        allocate(this%m_lineSearch, source = ls)
    end subroutine

    subroutine lso_set_default(this)
        class(line_search_optimizer), intent(inout) :: this
        type(line_search) :: ls
        call this%set_line_search(ls)
    end subroutine

    pure function lso_is_line_search_defined(this) result(x)
        class(line_search_optimizer), intent(in) :: this
        logical :: x
        x = allocated(this%m_lineSearch)
    end function

    pure function lso_get_use_search(this) result(x)
        class(line_search_optimizer), intent(in) :: this
        logical :: x
        x = this%m_useLineSearch
    end function

    subroutine lso_set_use_search(this, x)
        class(line_search_optimizer), intent(inout) :: this
        logical, intent(in) :: x
        this%m_useLineSearch = x
    end subroutine

    pure function lso_get_var_tol(this) result(x)
        class(line_search_optimizer), intent(in) :: this
        real(real64) :: x
        x = this%xtol_
    end function

    subroutine lso_set_var_tol(this, x)
        class(line_search_optimizer), intent(inout) :: this
        real(real64), intent(in) :: x
        this%xtol_ = x
    end subroutine

    subroutine bfgs_solve(this, fcn, x, fout, ib, args)     ! :557-770
        class(bfgs), intent(inout) :: this
        class(fcnnvar_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), optional :: fout
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        integer(int32) :: n
        integer(c_int) :: rc
        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: cib
        type(nlh_scalar_ctx), target :: ctx
        type(c_funptr) :: cgrad
        real(c_double), allocatable :: xc(:)
        real(c_double) :: fo
        class(line_search), allocatable :: ls

        n = fcn%get_variable_count()
        if (present(ib)) then           ! :594-602
            ib%iter_count = 0; ib%fcn_count = 0; ib%jacobian_count = 0; ib%gradient_count = 0
            ib%converge_on_fcn = .false.; ib%converge_on_chng = .false.; ib%converge_on_zero_diff = .false.
        end if
        call nlh_default_options(opts)
        if (this%get_use_line_search()) then        ! :603-608
            if (.not.this%is_line_search_defined()) call this%set_default_line_search()
            call this%get_line_search(ls)
            opts%ls_max_evals = ls%get_max_fcn_evals()
            opts%ls_alpha = ls%get_scaling_factor()
            opts%ls_factor = ls%get_distance_factor()
        end if
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! :614
        if (size(x) /= n) error stop NL_INVALID_INPUT_ERROR                      ! :615

        opts%max_evals = this%get_max_fcn_evals()
        opts%gtol = this%get_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%print_status = merge(1, 0, this%get_print_status())
        opts%use_line_search = merge(1, 0, this%get_use_line_search())

        ctx%helper => fcn
        if (present(args)) ctx%args => args
        cgrad = c_null_funptr
        if (fcn%is_gradient_defined()) cgrad = c_funloc(nlh_gradfcn_trampoline)
        allocate(xc(n))
        xc = x
        rc = nlh_bfgs_solve(nlh_default_handle(), opts, n, c_funloc(nlh_fcnnvar_trampoline), cgrad, c_loc(ctx), &
            xc, fo, cib)
        x = xc
        if (present(fout)) fout = fo    ! :762
        if (present(ib)) then           ! :751-759
            ib%iter_count = cib%iter_count
            ib%fcn_count = cib%fcn_count
            ib%jacobian_count = 0
            ib%gradient_count = cib%gradient_count
            ib%converge_on_fcn = .false.
            ib%converge_on_chng = cib%converge_on_chng /= 0
            ib%converge_on_zero_diff = cib%converge_on_zero_diff /= 0
        end if
        if (rc /= 0) error stop rc      ! :765-767
    end subroutine
end module
