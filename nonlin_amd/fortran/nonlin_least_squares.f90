! least_squares_solver, constrained_equation_solver and constrained_least_squares_solver with the reference's
! public interface (src/nonlin_least_squares.f90:20-74, 80-115, 793-935); solve marshals to nlh_lm_solve
! (lss_solve on the GPU, :118-391) / nlh_cls_solve (cls_solve, :938-1176) and performs the `error stop`
! the reference would.
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
    public :: constrained_equation_solver
    public :: constrained_least_squares_solver

    type, extends(equation_solver) :: least_squares_solver
        real(real64), private :: m_factor = 100.0d0
    contains
        procedure, public :: get_step_scaling_factor => lss_get_factor
        procedure, public :: set_step_scaling_factor => lss_set_factor
        procedure, public :: solve => lss_solve
        procedure, public :: solve_batch => lss_solve_batch
    end type

    type, abstract, extends(least_squares_solver) :: constrained_equation_solver
        real(real64), private, allocatable, dimension(:) :: m_upper
        real(real64), private, allocatable, dimension(:) :: m_lower
    contains
        procedure, public :: get_upper_limits => ces_get_upper_bounds
        procedure, public :: set_upper_limits => ces_set_upper_bounds
        procedure, public :: get_lower_limits => ces_get_lower_bounds
        procedure, public :: set_lower_limits => ces_set_lower_bounds
        procedure, public :: apply_limits => ces_apply_limits
    end type

    type, extends(constrained_equation_solver) :: constrained_least_squares_solver
        real(real64), private :: m_delta = 1.0d0
        real(real64), private :: m_scaling = 1.0d0
    contains
        procedure, public :: get_trust_region_radius => cls_get_radius
        procedure, public :: set_trust_region_radius => cls_set_radius
        procedure, public :: get_step_scaling_factor => cls_get_factor
        procedure, public :: set_step_scaling_factor => cls_set_factor
        procedure, public :: solve => cls_solve
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
        type(nlh_iteration_behavior) :: cib, cibs(1)
        type(nlh_callback_ctx), target :: ctx
        type(c_funptr) :: cjac
        real(c_double), allocatable :: xc(:), fc(:)
        type(device_model_batch) :: dm
        integer(c_int32_t) :: st(1)

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
        if (fcn%is_device_model_defined()) then
            ! set_device_model: the whole iteration runs on the GPU (FD Jacobian, factorisation, lmpar, trial
            ! evaluations), no host callback; st is the code the reference would stop with
            dm = fcn%device_model()
            rc = nlh_dq_model_lm_solve(nlh_default_handle(), opts, dm%c_handle(), xc, fc, cibs, st)
            cib = cibs(1)
            if (rc == 0) rc = st(1)
        else
            rc = nlh_lm_solve(nlh_default_handle(), opts, neqn, nvar, c_funloc(nlh_vecfcn_trampoline), cjac, &
                c_loc(ctx), xc, fc, cib)
        end if
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

    !> Extension: least_squares_solver%solve (lss_solve, :118-391) for every problem of a device model batch in
    !> one call.  x(n, nprob) start points in / solutions out, fvec(m, nprob) residuals at the solutions,
    !> ib(nprob) the counters and flags of each solve, status(nprob) = 0 or the code the reference would
    !> `error stop` with for that problem (no process abort: the other problems are unaffected).
    subroutine lss_solve_batch(this, model, x, fvec, ib, status)
        class(least_squares_solver), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:,:) :: fvec
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        integer(int32) :: neqn, nvar, nprob, k
        integer(c_int) :: rc
        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: cib(:)
        integer(c_int32_t), allocatable :: st(:)
        real(c_double), allocatable :: xc(:,:), fc(:,:)

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        neqn = model%get_equation_count()
        nvar = model%get_variable_count()
        nprob = model%get_problem_count()
        if (nvar > neqn) error stop NL_UNDERDEFINED_PROBLEM_ERROR
        if (size(x, 1) /= nvar .or. size(x, 2) /= nprob) error stop 3
        if (size(fvec, 1) /= neqn .or. size(fvec, 2) /= nprob) error stop 4
        if (present(ib)) then
            if (size(ib) /= nprob) error stop 5
        end if
        if (present(status)) then
            if (size(status) /= nprob) error stop 6
        end if
        call nlh_default_options(opts)
        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = 0
        opts%factor = this%m_factor
        opts%factor_policy = this%factor_policy
        allocate(cib(nprob), st(nprob), fc(neqn, nprob))
        xc = x
        rc = nlh_dq_model_lm_solve(nlh_default_handle(), opts, model%c_handle(), xc, fc, cib, st)
        if (rc /= 0) error stop rc      ! a library failure, not a per-problem outcome
        x = xc
        fvec = fc
        if (present(status)) status = st
        if (present(ib)) then
            do k = 1, nprob
                ib(k)%iter_count = cib(k)%iter_count
                ib(k)%fcn_count = cib(k)%fcn_count
                ib(k)%jacobian_count = cib(k)%jacobian_count
                ib(k)%gradient_count = cib(k)%gradient_count
                ib(k)%converge_on_fcn = cib(k)%converge_on_fcn /= 0
                ib(k)%converge_on_chng = cib(k)%converge_on_chng /= 0
                ib(k)%converge_on_zero_diff = cib(k)%converge_on_zero_diff /= 0
            end do
        end if
    end subroutine

    pure function ces_get_upper_bounds(this) result(rst)    ! :796-808
        class(constrained_equation_solver), intent(in) :: this
        real(real64), allocatable, dimension(:) :: rst
        if (allocated(this%m_upper)) then
            rst = this%m_upper
        else
            allocate(rst(0))
        end if
    end function

    subroutine ces_set_upper_bounds(this, x)                ! :811-824
        class(constrained_equation_solver), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        if (allocated(this%m_upper)) deallocate(this%m_upper)
        this%m_upper = x
    end subroutine

    pure function ces_get_lower_bounds(this) result(rst)    ! :827-839
        class(constrained_equation_solver), intent(in) :: this
        real(real64), allocatable, dimension(:) :: rst
        if (allocated(this%m_lower)) then
            rst = this%m_lower
        else
            allocate(rst(0))
        end if
    end function

    subroutine ces_set_lower_bounds(this, x)                ! :842-855
        class(constrained_equation_solver), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        if (allocated(this%m_lower)) deallocate(this%m_lower)
        this%m_lower = x
    end subroutine

    subroutine ces_apply_limits(this, x)                    ! :858-883
        class(constrained_equation_solver), intent(in) :: this
        real(real64), intent(inout), dimension(:) :: x
        integer(int32) :: i, nu, nl, n
        real(real64), allocatable, dimension(:) :: maxX, minX
        maxX = this%get_upper_limits()
        minX = this%get_lower_limits()
        n = size(x)
        nu = min(n, size(maxX))
        nl = min(n, size(minX))
        do i = 1, nl
            if (x(i) < minX(i)) x(i) = minX(i)
        end do
        do i = 1, nu
            if (x(i) > maxX(i)) x(i) = maxX(i)
        end do
    end subroutine

    pure function cls_get_radius(this) result(rst)          ! :888-895
        class(constrained_least_squares_solver), intent(in) :: this
        real(real64) :: rst
        rst = this%m_delta
    end function

    subroutine cls_set_radius(this, x)                      ! :898-910
        class(constrained_least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        if (x <= 0.0d0) then
            this%m_delta = 1.0d0
        else
            this%m_delta = x
        end if
    end subroutine

    pure function cls_get_factor(this) result(rst)          ! :913-920
        class(constrained_least_squares_solver), intent(in) :: this
        real(real64) :: rst
        rst = this%m_scaling
    end function

    subroutine cls_set_factor(this, x)                      ! :923-935
        class(constrained_least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        if (x <= 0.0d0) then
            this%m_scaling = 1.0d0
        else
            this%m_scaling = x
        end if
    end subroutine

    subroutine cls_solve(this, fcn, x, fvec, ib, args)      ! :938-1176
        class(constrained_least_squares_solver), intent(inout) :: this
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
        real(c_double), allocatable :: xc(:), fc(:), xl(:), xu(:)

        neqn = fcn%get_equation_count()
        nvar = fcn%get_variable_count()
        xl = this%get_lower_limits()
        xu = this%get_upper_limits()
        if (present(ib)) then           ! :977-985
            ib%iter_count = 0; ib%fcn_count = 0; ib%jacobian_count = 0; ib%gradient_count = 0
            ib%converge_on_fcn = .false.; ib%converge_on_chng = .false.; ib%converge_on_zero_diff = .false.
        end if
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! :988
        if (nvar > neqn) error stop NL_UNDERDEFINED_PROBLEM_ERROR                ! :989
        flag = 0
        if (size(x) /= nvar) then
            flag = 3
        else if (size(fvec) /= neqn) then
            flag = 4
        end if
        if (flag /= 0) error stop flag
        if (size(xl) /= nvar) then      ! :999-1009: wrong-sized limits are replaced and stored
            deallocate(xl)
            allocate(xl(nvar), source = -huge(0.0d0))
            call this%set_lower_limits(xl)
        end if
        if (size(xu) /= nvar) then
            deallocate(xu)
            allocate(xu(nvar), source = huge(0.0d0))
            call this%set_upper_limits(xu)
        end if

        call nlh_default_options(opts)
        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = merge(1, 0, this%get_print_status())

        ctx%helper => fcn
        if (present(args)) ctx%args => args
        cjac = c_null_funptr
        if (fcn%is_jacobian_defined()) cjac = c_funloc(nlh_jacfcn_trampoline)
        allocate(xc(nvar), fc(neqn))
        xc = x
        rc = nlh_cls_solve(nlh_default_handle(), opts, this%m_delta, this%m_scaling, xl, xu, neqn, nvar, &
            c_funloc(nlh_vecfcn_trampoline), cjac, c_loc(ctx), xc, fc, cib)
        x = xc
        fvec = fc
        if (present(ib)) then           ! :1163-1170
            ib%iter_count = cib%iter_count
            ib%fcn_count = cib%fcn_count
            ib%jacobian_count = cib%jacobian_count
            ib%converge_on_fcn = cib%converge_on_fcn /= 0
            ib%converge_on_chng = cib%converge_on_chng /= 0
            ib%converge_on_zero_diff = cib%converge_on_zero_diff /= 0
        end if
        if (rc /= 0) error stop rc      ! :1173-1175
    end subroutine
end module
