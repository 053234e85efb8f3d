! line_search_solver, newton_solver and quasi_newton_solver: the public types and bindings of
! src/nonlin_solve.f90:20-67.  `solve` marshals to nlh_newton_solve / nlh_dq_model_newton_solve (ns_solve on
! the GPU, :452-638) and nlh_quasi_newton_solve (qns_solve, :156-427); the line search the solver owns is only
! a parameter record here (nonlin_linesearch), the search runs behind the C ABI.
module nonlin_solve
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_error_handling
    use nonlin_multi_eqn_mult_var
    use nonlin_linesearch
    use nonlin_types
    use nonlin_hip_c
    use nonlin_shim_support
    implicit none
    private
    public :: line_search_solver
    public :: newton_solver
    public :: quasi_newton_solver

    type, abstract, extends(equation_solver) :: line_search_solver
        class(line_search), private, allocatable :: search_
        logical, private :: search_on_ = .true.
    contains
        procedure, public :: get_line_search => lsx_copy_search
        procedure, public :: set_line_search => lsx_put_search
        procedure, public :: set_default_line_search => lsx_default_search
        procedure, public :: is_line_search_defined => lsx_has_search
        procedure, public :: get_use_line_search => lsx_enabled
        procedure, public :: set_use_line_search => lsx_enable
        procedure, public :: export_search_options => lsx_export
    end type

    type, extends(line_search_solver) :: newton_solver
    contains
        procedure, public :: solve => newton_solve_one
        procedure, public :: solve_batch => newton_solve_many
    end type

    type, extends(line_search_solver) :: quasi_newton_solver
        integer(int32), private :: refresh_every_ = 5    ! iterations between fresh Jacobians (reference default, :51)
    contains
        procedure, public :: solve => broyden_solve_one
        procedure, public :: solve_batch => broyden_solve_many
        procedure, public :: get_jacobian_interval => broyden_refresh
        procedure, public :: set_jacobian_interval => broyden_put_refresh
    end type

contains
    ! ---- line_search_solver -------------------------------------------------------------------------------------
    pure logical function lsx_has_search(this)
        class(line_search_solver), intent(in) :: this
        lsx_has_search = allocated(this%search_)
    end function

    !> ls = a copy of the solver's search object; left unallocated while none has been set.
    subroutine lsx_copy_search(this, ls)
        class(line_search_solver), intent(in) :: this
        class(line_search), intent(out), allocatable :: ls
        if (this%is_line_search_defined()) allocate(ls, source = this%search_)
    end subroutine

    subroutine lsx_put_search(this, ls)
        class(line_search_solver), intent(inout) :: this
        class(line_search), intent(in) :: ls
        if (this%is_line_search_defined()) deallocate(this%search_)
        allocate(this%search_, source = ls)
    end subroutine

    subroutine lsx_default_search(this)
        class(line_search_solver), intent(inout) :: this
        call this%set_line_search(line_search())
    end subroutine

    pure logical function lsx_enabled(this)
        class(line_search_solver), intent(in) :: this
        lsx_enabled = this%search_on_
    end function

    subroutine lsx_enable(this, x)
        class(line_search_solver), intent(inout) :: this
        logical, intent(in) :: x
        this%search_on_ = x
    end subroutine

    !> Extension: export_options plus the search settings.  As in the reference (:511-515, :233-237) a solver that
    !> searches but has no search object yet gets the default one, and keeps it.
    subroutine lsx_export(this, opts, quiet)
        class(line_search_solver), intent(inout) :: this
        type(nlh_options), intent(out) :: opts
        logical, intent(in), optional :: quiet
        call this%export_options(opts, quiet)
        opts%use_line_search = merge(1, 0, this%search_on_)
        if (.not.this%search_on_) return
        if (.not.this%is_line_search_defined()) call this%set_default_line_search()
        opts%ls_max_evals = this%search_%get_max_fcn_evals()
        opts%ls_alpha = this%search_%get_scaling_factor()
        opts%ls_factor = this%search_%get_distance_factor()
    end subroutine

    !> Checks every square-system solve makes before it crosses the boundary (:518-525, :240-247).
    subroutine require_square_system(fcn, nx, nf)
        class(vecfcn_helper), intent(in) :: fcn
        integer(int32), intent(in) :: nx, nf
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        if (fcn%get_variable_count() /= fcn%get_equation_count()) error stop NL_INVALID_INPUT_ERROR
        call require_vector_sizes(nx, nf, fcn%get_variable_count(), fcn%get_equation_count())
    end subroutine

    ! ---- newton_solver -----------------------------------------------------------------------------------------
    subroutine newton_solve_one(this, fcn, x, fvec, ib, args)
        class(newton_solver), intent(inout) :: this
        class(vecfcn_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: fvec
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: counters(1)
        type(nlh_callback_ctx), target :: ctx
        type(device_model_batch) :: onchip
        type(c_funptr) :: jac_entry
        real(c_double), allocatable :: xwork(:), fwork(:)
        integer(c_int32_t) :: outcome(1)
        integer(c_int) :: rc
        integer(int32) :: n

        n = fcn%get_variable_count()
        if (present(ib)) call behavior_clear(ib)
        call this%export_search_options(opts)
        call require_square_system(fcn, size(x), size(fvec))
        allocate(xwork(n), source = x)
        allocate(fwork(n))
        if (fcn%is_device_model_defined()) then
            ! residuals, Jacobian (the model's own when it was bound with analytic = .true., forward differences
            ! otherwise) and the LU solve on the GPU, no host callback
            onchip = fcn%device_model()
            rc = nlh_dq_model_newton_solve(nlh_default_handle(), opts, onchip%c_handle(), &
                merge(1_c_int32_t, 0_c_int32_t, onchip%uses_analytic_jacobian()), xwork, fwork, counters, outcome)
            if (rc == 0) rc = outcome(1)
        else
            ctx%helper => fcn
            if (present(args)) ctx%args => args
            jac_entry = c_null_funptr
            if (fcn%is_jacobian_defined()) jac_entry = c_funloc(nlh_jacfcn_trampoline)
            rc = nlh_newton_solve(nlh_default_handle(), opts, n, c_funloc(nlh_vecfcn_trampoline), jac_entry, &
                c_loc(ctx), xwork, fwork, counters(1))
        end if
        x = xwork
        fvec = fwork
        if (present(ib)) call behavior_import(ib, counters(1))
        if (rc /= 0) error stop rc      ! the reference's stops at :604-608, :635-637 and inside the line search
    end subroutine

    !> Extension: newton_solver%solve (ns_solve, :452-638) for every (square) problem of a device model batch.
    !> Arguments as least_squares_solver%solve_batch.
    subroutine newton_solve_many(this, model, x, fvec, ib, status)
        class(newton_solver), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:,:) :: fvec
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: counters(:)
        integer(c_int32_t), allocatable :: outcome(:)
        real(c_double), allocatable :: xwork(:,:), fwork(:,:)
        integer(c_int) :: rc
        integer(int32) :: n, count

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        n = model%get_variable_count()
        count = model%get_problem_count()
        if (model%get_equation_count() /= n) error stop NL_INVALID_INPUT_ERROR
        if (any(shape(x) /= [n, count])) error stop 3
        if (any(shape(fvec) /= [n, count])) error stop 4
        call this%export_search_options(opts, quiet = .true.)
        allocate(counters(count), outcome(count), fwork(n, count))
        allocate(xwork(n, count), source = x)
        rc = nlh_dq_model_newton_solve(nlh_default_handle(), opts, model%c_handle(), &
            merge(1_c_int32_t, 0_c_int32_t, model%uses_analytic_jacobian()), xwork, fwork, counters, outcome)
        if (rc /= 0) error stop rc
        x = xwork
        fvec = fwork
        if (present(status)) status = outcome
        if (present(ib)) call behavior_import(ib, counters)
    end subroutine

    ! ---- quasi_newton_solver -----------------------------------------------------------------------------------
    !> Extension: quasi_newton_solver%solve (qns_solve, :156-427) for every (square) problem of a device model batch.
    !> Arguments as least_squares_solver%solve_batch.
    subroutine broyden_solve_many(this, model, x, fvec, ib, status)
        class(quasi_newton_solver), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:,:) :: fvec
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: counters(:)
        integer(c_int32_t), allocatable :: outcome(:)
        real(c_double), allocatable :: xwork(:,:), fwork(:,:)
        integer(c_int) :: rc
        integer(int32) :: n, count

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        n = model%get_variable_count()
        count = model%get_problem_count()
        if (model%get_equation_count() /= n) error stop NL_INVALID_INPUT_ERROR
        if (any(shape(x) /= [n, count])) error stop 3
        if (any(shape(fvec) /= [n, count])) error stop 4
        call this%export_search_options(opts, quiet = .true.)
        allocate(counters(count), outcome(count), fwork(n, count))
        allocate(xwork(n, count), source = x)
        rc = nlh_dq_model_quasi_newton_solve(nlh_default_handle(), opts, model%c_handle(), this%refresh_every_, &
            merge(1_c_int32_t, 0_c_int32_t, model%uses_analytic_jacobian()), xwork, fwork, counters, outcome)
        if (rc /= 0) error stop rc
        x = xwork
        fvec = fwork
        if (present(status)) status = outcome
        if (present(ib)) call behavior_import(ib, counters)
    end subroutine

    subroutine broyden_solve_one(this, fcn, x, fvec, ib, args)
        class(quasi_newton_solver), intent(inout) :: this
        class(vecfcn_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: fvec
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: counters
        type(nlh_callback_ctx), target :: ctx
        type(c_funptr) :: jac_entry
        real(c_double), allocatable :: xwork(:), fwork(:)
        integer(c_int) :: rc
        integer(int32) :: n

        n = fcn%get_variable_count()
        if (present(ib)) call behavior_clear(ib)
        call this%export_search_options(opts)
        call require_square_system(fcn, size(x), size(fvec))
        allocate(xwork(n), source = x)
        allocate(fwork(n))
        if (fcn%is_device_model_defined()) then
            ! a device residual (set_device_model / set_device_fcn): Broyden's iteration entirely on the GPU
            block
                type(device_model_batch) :: onchip
                type(nlh_iteration_behavior) :: c1(1)
                integer(c_int32_t) :: outcome(1)
                onchip = fcn%device_model()
                rc = nlh_dq_model_quasi_newton_solve(nlh_default_handle(), opts, onchip%c_handle(), this%refresh_every_, &
                    merge(1_c_int32_t, 0_c_int32_t, onchip%uses_analytic_jacobian()), xwork, fwork, c1, outcome)
                if (rc == 0) rc = outcome(1)
                counters = c1(1)
            end block
        else
            ctx%helper => fcn
            if (present(args)) ctx%args => args
            jac_entry = c_null_funptr
            if (fcn%is_jacobian_defined()) jac_entry = c_funloc(nlh_jacfcn_trampoline)
            rc = nlh_quasi_newton_solve(nlh_default_handle(), opts, this%refresh_every_, n, &
                c_funloc(nlh_vecfcn_trampoline), jac_entry, c_loc(ctx), xwork, fwork, counters)
        end if
        x = xwork
        fvec = fwork
        if (present(ib)) call behavior_import(ib, counters)
        if (rc /= 0) error stop rc      ! the reference's stops at :376, :425-427 and inside the line search
    end subroutine

    pure integer(int32) function broyden_refresh(this)
        class(quasi_newton_solver), intent(in) :: this
        broyden_refresh = this%refresh_every_
    end function

    subroutine broyden_put_refresh(this, n)
        class(quasi_newton_solver), intent(inout) :: this
        integer(int32), intent(in) :: n
        this%refresh_every_ = n
    end subroutine
end module
