! line_search_solver, newton_solver and quasi_newton_solver with the reference's public interface
! (src/nonlin_solve.f90:20-67, 92-151, 429-447); solve marshals to nlh_newton_solve (ns_solve, :452-638)
! and nlh_quasi_newton_solve (qns_solve, :156-427).
module nonlin_solve
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_error_handling
    use nonlin_multi_eqn_mult_var
    use nonlin_linesearch
    use nonlin_types
    use nonlin_hip_c
    implicit none
    private
    public :: line_search_solver
    public :: newton_solver
    public :: quasi_newton_solver

    type, abstract, extends(equation_solver) :: line_search_solver
        class(line_search), private, allocatable :: m_lineSearch
        logical, private :: m_useLineSearch = .true.
    contains
        procedure, public :: get_line_search => lss_get_line_search
        procedure, public :: set_line_search => lss_set_line_search
        procedure, public :: set_default_line_search => lss_set_default
        procedure, public :: is_line_search_defined => lss_is_line_search_defined
        procedure, public :: get_use_line_search => lss_get_use_search
        procedure, public :: set_use_line_search => lss_set_use_search
    end type

    type, extends(line_search_solver) :: newton_solver
    contains
        procedure, public :: solve => ns_solve
        procedure, public :: solve_batch => ns_solve_batch
    end type

    type, extends(line_search_solver) :: quasi_newton_solver
        integer(int32), private :: m_jDelta = 5         ! :51
    contains
        procedure, public :: solve => qns_solve
        procedure, public :: get_jacobian_interval => qns_get_jac_interval
        procedure, public :: set_jacobian_interval => qns_set_jac_interval
    end type

contains
    subroutine lss_get_line_search(this, ls)
        class(line_search_solver), intent(in) :: this
        class(line_search), intent(out), allocatable :: ls
        if (allocated(this%m_lineSearch)) allocate(ls, source = this%m_lineSearch)
    end subroutine

    subroutine lss_set_line_search(this, ls)
        class(line_search_solver), intent(inout) :: this
        class(line_search), intent(in) :: ls
        if (allocated(this%m_lineSearch)) deallocate(this%m_lineSearch)
        allocate(this%m_lineSearch, source = ls)
    end subroutine

    subroutine lss_set_default(this)
        class(line_search_solver), intent(inout) :: this
        type(line_search) :: ls
        call this%set_line_search(ls)
    end subroutine

    pure function lss_is_line_search_defined(this) result(x)
        class(line_search_solver), intent(in) :: this
        logical :: x
        x = allocated(this%m_lineSearch)
    end function

    pure function lss_get_use_search(this) result(x)
        class(line_search_solver), intent(in) :: this
        logical :: x
        x = this%m_useLineSearch
    end function

    subroutine lss_set_use_search(this, x)
        class(line_search_solver), intent(inout) :: this
        logical, intent(in) :: x
        this%m_useLineSearch = x
    end subroutine

    subroutine ns_solve(this, fcn, x, fvec, ib, args)
        class(newton_solver), intent(inout) :: this
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
        class(line_search), allocatable :: ls
        type(device_model_batch) :: dm
        integer(c_int32_t) :: st(1)

        neqn = fcn%get_equation_count()
        nvar = fcn%get_variable_count()
        if (present(ib)) then           ! :502-510
            ib%iter_count = 0; ib%fcn_count = 0; ib%jacobian_count = 0; ib%gradient_count = 0
            ib%converge_on_fcn = .false.; ib%converge_on_chng = .false.; ib%converge_on_zero_diff = .false.
        end if
        call nlh_default_options(opts)
        if (this%get_use_line_search()) then        ! :511-515
            if (.not.this%is_line_search_defined()) call this%set_default_line_search()
            call this%get_line_search(ls)
            opts%ls_max_evals = ls%get_max_fcn_evals()
            opts%ls_alpha = ls%get_scaling_factor()
            opts%ls_factor = ls%get_distance_factor()
        end if
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! :518
        if (nvar /= neqn) error stop NL_INVALID_INPUT_ERROR                      ! :519
        flag = 0
        if (size(x) /= nvar) then
            flag = 3
        else if (size(fvec) /= neqn) then
            flag = 4
        end if
        if (flag /= 0) error stop flag

        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = merge(1, 0, this%get_print_status())
        opts%use_line_search = merge(1, 0, this%get_use_line_search())

        ctx%helper => fcn
        if (present(args)) ctx%args => args
        cjac = c_null_funptr
        if (fcn%is_jacobian_defined()) cjac = c_funloc(nlh_jacfcn_trampoline)
        allocate(xc(nvar), fc(neqn))
        xc = x
        if (fcn%is_device_model_defined()) then
            ! set_device_model: residuals, Jacobian (the model's own when it was bound with analytic = .true.,
            ! forward differences otherwise) and the LU solve on the GPU, no host callback
            dm = fcn%device_model()
            rc = nlh_dq_model_newton_solve(nlh_default_handle(), opts, dm%c_handle(), &
                merge(1_c_int32_t, 0_c_int32_t, dm%uses_analytic_jacobian()), xc, fc, cibs, st)
            cib = cibs(1)
            if (rc == 0) rc = st(1)
        else
            rc = nlh_newton_solve(nlh_default_handle(), opts, nvar, c_funloc(nlh_vecfcn_trampoline), cjac, &
                c_loc(ctx), xc, fc, cib)
        end if
        x = xc
        fvec = fc
        if (present(ib)) then           ! :624-632
            ib%iter_count = cib%iter_count
            ib%fcn_count = cib%fcn_count
            ib%jacobian_count = cib%jacobian_count
            ib%gradient_count = 0
            ib%converge_on_fcn = cib%converge_on_fcn /= 0
            ib%converge_on_chng = cib%converge_on_chng /= 0
            ib%converge_on_zero_diff = cib%converge_on_zero_diff /= 0
        end if
        if (rc /= 0) error stop rc      ! :635-637, :604-608, line-search stops
    end subroutine

    !> Extension: newton_solver%solve (ns_solve, :452-638) for every (square) problem of a device model batch.
    !> Arguments as least_squares_solver%solve_batch.
    subroutine ns_solve_batch(this, model, x, fvec, ib, status)
        class(newton_solver), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:,:) :: fvec
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        integer(int32) :: n, nprob, k
        integer(c_int) :: rc
        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: cib(:)
        integer(c_int32_t), allocatable :: st(:)
        real(c_double), allocatable :: xc(:,:), fc(:,:)
        class(line_search), allocatable :: ls

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        n = model%get_variable_count()
        nprob = model%get_problem_count()
        if (model%get_equation_count() /= n) error stop NL_INVALID_INPUT_ERROR
        if (size(x, 1) /= n .or. size(x, 2) /= nprob) error stop 3
        if (size(fvec, 1) /= n .or. size(fvec, 2) /= nprob) error stop 4
        call nlh_default_options(opts)
        if (this%get_use_line_search()) then
            if (.not.this%is_line_search_defined()) call this%set_default_line_search()
            call this%get_line_search(ls)
            opts%ls_max_evals = ls%get_max_fcn_evals()
            opts%ls_alpha = ls%get_scaling_factor()
            opts%ls_factor = ls%get_distance_factor()
        end if
        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = 0
        opts%use_line_search = merge(1, 0, this%get_use_line_search())
        allocate(cib(nprob), st(nprob), fc(n, nprob))
        xc = x
        rc = nlh_dq_model_newton_solve(nlh_default_handle(), opts, model%c_handle(), &
            merge(1_c_int32_t, 0_c_int32_t, model%uses_analytic_jacobian()), xc, fc, cib, st)
        if (rc /= 0) error stop rc
        x = xc
        fvec = fc
        if (present(status)) status = st
        if (present(ib)) then
            do k = 1, nprob
                ib(k)%iter_count = cib(k)%iter_count
                ib(k)%fcn_count = cib(k)%fcn_count
                ib(k)%jacobian_count = cib(k)%jacobian_count
                ib(k)%gradient_count = 0
                ib(k)%converge_on_fcn = cib(k)%converge_on_fcn /= 0
                ib(k)%converge_on_chng = cib(k)%converge_on_chng /= 0
                ib(k)%converge_on_zero_diff = cib(k)%converge_on_zero_diff /= 0
            end do
        end if
    end subroutine

    subroutine qns_solve(this, fcn, x, fvec, ib, args)
        class(quasi_newton_solver), intent(inout) :: this
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
        class(line_search), allocatable :: ls

        neqn = fcn%get_equation_count()
        nvar = fcn%get_variable_count()
        if (present(ib)) then           ! :224-232
            ib%iter_count = 0; ib%fcn_count = 0; ib%jacobian_count = 0; ib%gradient_count = 0
            ib%converge_on_fcn = .false.; ib%converge_on_chng = .false.; ib%converge_on_zero_diff = .false.
        end if
        call nlh_default_options(opts)
        if (this%get_use_line_search()) then        ! :233-237
            if (.not.this%is_line_search_defined()) call this%set_default_line_search()
            call this%get_line_search(ls)
            opts%ls_max_evals = ls%get_max_fcn_evals()
            opts%ls_alpha = ls%get_scaling_factor()
            opts%ls_factor = ls%get_distance_factor()
        end if
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! :240
        if (nvar /= neqn) error stop NL_INVALID_INPUT_ERROR                      ! :241
        flag = 0
        if (size(x) /= nvar) then
            flag = 3
        else if (size(fvec) /= neqn) then
            flag = 4
        end if
        if (flag /= 0) error stop flag

        opts%max_evals = this%get_max_fcn_evals()
        opts%ftol = this%get_fcn_tolerance()
        opts%xtol = this%get_var_tolerance()
        opts%gtol = this%get_gradient_tolerance()
        opts%print_status = merge(1, 0, this%get_print_status())
        opts%use_line_search = merge(1, 0, this%get_use_line_search())

        ctx%helper => fcn
        if (present(args)) ctx%args => args
        cjac = c_null_funptr
        if (fcn%is_jacobian_defined()) cjac = c_funloc(nlh_jacfcn_trampoline)
        allocate(xc(nvar), fc(neqn))
        xc = x
        rc = nlh_quasi_newton_solve(nlh_default_handle(), opts, this%m_jDelta, nvar, &
            c_funloc(nlh_vecfcn_trampoline), cjac, c_loc(ctx), xc, fc, cib)
        x = xc
        fvec = fc
        if (present(ib)) then           ! :414-422
            ib%iter_count = cib%iter_count
            ib%fcn_count = cib%fcn_count
            ib%jacobian_count = cib%jacobian_count
            ib%gradient_count = 0
            ib%converge_on_fcn = cib%converge_on_fcn /= 0
            ib%converge_on_chng = cib%converge_on_chng /= 0
            ib%converge_on_zero_diff = cib%converge_on_zero_diff /= 0
        end if
        if (rc /= 0) error stop rc      ! :425-427, :376, line-search stops
    end subroutine

    pure function qns_get_jac_interval(this) result(n)      ! :429-436
        class(quasi_newton_solver), intent(in) :: this
        integer(int32) :: n
        n = this%m_jDelta
    end function

    subroutine qns_set_jac_interval(this, n)                ! :439-447
        class(quasi_newton_solver), intent(inout) :: this
        integer(int32), intent(in) :: n
        this%m_jDelta = n
    end subroutine
end module
