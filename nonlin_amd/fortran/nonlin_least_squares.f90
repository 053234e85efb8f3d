! least_squares_solver with the reference's public interface (src/nonlin_least_squares.f90:20-31,
! 80-115); solve marshals to nlh_lm_solve (lss_solve on the GPU, :118-391) and performs the
! `error stop` the reference would.
module nonlin_least_squares
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_multi_eqn_mult_var
    use nonlin_error_handling
    use nonlin_types
    use nonlin_hip_c
    implicit none
    private
    public :: least_squares_solver

    type, extends(equation_solver) :: least_squares_solver
        real(real64), private :: m_factor = 100.0d0
    contains
        procedure, public :: get_step_scaling_factor => lss_get_factor
        procedure, public :: set_step_scaling_factor => lss_set_factor
        procedure, public :: solve => lss_solve
    end type

contains
    pure function lss_get_factor(this) result(x)
        class(least_squares_solver), intent(in) :: this
        real(real64) :: x
        x = this%m_factor
    end function

    subroutine lss_set_factor(this, x)      ! clamp: :108-114
        class(least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        if (x < 0.1d0) then
            this%m_factor = 0.1d0
        else if (x > 1.0d2) then
            this%m_factor = 1.0d2
        else
            this%m_factor = x
        end if
    end subroutine

    subroutine lss_solve(this, fcn, x, fvec, ib, args)
        class(least_squares_solver), intent(inout) :: this
        class(vecfcn_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: fvec
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        integer(int32) :: neqn, nvar, flag
        integer(c_int) :: rc
        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: cib
        type(nlh_callback_ctx), target :: ctx
        type(c_funptr) :: cjac
        real(c_double), allocatable :: xc(:), fc(:)

        neqn = fcn%get_equation_count()
        nvar = fcn%get_variable_count()
        if (present(ib)) then           ! :177-185
            ib%iter_count = 0; ib%fcn_count = 0; ib%jacobian_count = 0; ib%gradient_count = 0
            ib%converge_on_fcn = .false.; ib%converge_on_chng = .false.; ib%converge_on_zero_diff = .false.
        end if
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! :188
        if (nvar > neqn) error stop NL_UNDERDEFINED_PROBLEM_ERROR                ! :189
        flag = 0
        if (size(x) /= nvar) then
            flag = 3
        else if (size(fvec) /= neqn) then
            flag = 4
        end if
        if (flag /= 0) error stop flag

        call nlh_default_options(opts)
        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = merge(1, 0, this%get_print_status())
        opts%factor = this%m_factor
        opts%factor_policy = this%factor_policy

        ctx%helper => fcn
        if (present(args)) ctx%args => args
        cjac = c_null_funptr
        if (fcn%is_jacobian_defined()) cjac = c_funloc(nlh_jacfcn_trampoline)
        allocate(xc(nvar), fc(neqn))    ! contiguous copies: the dummies may be strided sections
        xc = x
        rc = nlh_lm_solve(nlh_default_handle(), opts, neqn, nvar, c_funloc(nlh_vecfcn_trampoline), cjac, &
            c_loc(ctx), xc, fc, cib)
        x = xc
        fvec = fc
        if (present(ib)) then           ! :378-385
            ib%iter_count = cib%iter_count
            ib%fcn_count = cib%fcn_count
            ib%jacobian_count = cib%jacobian_count
            ib%gradient_count = cib%gradient_count
            ib%converge_on_fcn = cib%converge_on_fcn /= 0
            ib%converge_on_chng = cib%converge_on_chng /= 0
            ib%converge_on_zero_diff = cib%converge_on_zero_diff /= 0
        end if
        if (rc /= 0) error stop rc      ! :388-390 (NL_CONVERGENCE_ERROR) or a library failure
    end subroutine
end module
