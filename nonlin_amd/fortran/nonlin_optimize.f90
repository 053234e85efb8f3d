! line_search_optimizer and bfgs: the public types and bindings of src/nonlin_optimize.f90:44-72, 470-556.
! bfgs%solve marshals to nlh_bfgs_solve (bfgs_solve behind the C ABI, :557-770); the line search is a parameter
! record here.  nelder_mead is outside the hot path.
module nonlin_optimize
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_linesearch
    use nonlin_error_handling
    use nonlin_multi_var
    use nonlin_multi_eqn_mult_var, only : device_model_batch
    use nonlin_types
    use nonlin_hip_c
    use nonlin_shim_support
    implicit none
    private
    public :: line_search_optimizer
    public :: bfgs

    type, abstract, extends(equation_optimizer) :: line_search_optimizer
        class(line_search), private, allocatable :: search_
        logical, private :: search_on_ = .true.
        real(real64), private :: step_tol_ = 1.0d-12         ! convergence on the change in x
    contains
        procedure, public :: get_line_search => lsopt_copy_search
        procedure, public :: set_line_search => lsopt_put_search
        procedure, public :: set_default_line_search => lsopt_default_search
        procedure, public :: is_line_search_defined => lsopt_has_search
        procedure, public :: get_use_line_search => lsopt_enabled
        procedure, public :: set_use_line_search => lsopt_enable
        procedure, public :: get_var_tolerance => lsopt_step_tol
        procedure, public :: set_var_tolerance => lsopt_put_step_tol
        procedure, public :: export_options => lsopt_export
    end type

    type, extends(line_search_optimizer) :: bfgs
    contains
        procedure, public :: solve => bfgs_solve_one
        procedure, public :: solve_batch => bfgs_solve_many
    end type

contains
    pure logical function lsopt_has_search(this)
        class(line_search_optimizer), intent(in) :: this
        lsopt_has_search = allocated(this%search_)
    end function

    !> ls = a copy of the optimizer's search object; left unallocated while none has been set.
    subroutine lsopt_copy_search(this, ls)
        class(line_search_optimizer), intent(in) :: this
        class(line_search), intent(out), allocatable :: ls
        if (this%is_line_search_defined()) allocate(ls, source = this%search_)
    end subroutine

    subroutine lsopt_put_search(this, ls)
        class(line_search_optimizer), intent(inout) :: this
        class(line_search), intent(in) :: ls
        if (this%is_line_search_defined()) deallocate(this%search_)
        allocate(this%search_, source = ls)
    end subroutine

    subroutine lsopt_default_search(this)
        class(line_search_optimizer), intent(inout) :: this
        call this%set_line_search(line_search())
    end subroutine

    pure logical function lsopt_enabled(this)
        class(line_search_optimizer), intent(in) :: this
        lsopt_enabled = this%search_on_
    end function

    subroutine lsopt_enable(this, x)
        class(line_search_optimizer), intent(inout) :: this
        logical, intent(in) :: x
        this%search_on_ = x
    end subroutine

    pure real(real64) function lsopt_step_tol(this)
        class(line_search_optimizer), intent(in) :: this
        lsopt_step_tol = this%step_tol_
    end function

    subroutine lsopt_put_step_tol(this, x)
        class(line_search_optimizer), intent(inout) :: this
        real(real64), intent(in) :: x
        this%step_tol_ = x
    end subroutine

    !> Extension: this optimizer's settings as the C ABI's option record.  As in the reference (:603-608) an
    !> optimizer that searches but has no search object yet gets the default one, and keeps it.
    subroutine lsopt_export(this, opts)
        class(line_search_optimizer), intent(inout) :: this
        type(nlh_options), intent(out) :: opts
        call nlh_default_options(opts)
        opts%max_evals = this%get_max_fcn_evals()
        opts%gtol = this%get_tolerance()
        opts%xtol = this%step_tol_
        opts%print_status = merge(1, 0, this%get_print_status())
        opts%use_line_search = merge(1, 0, this%search_on_)
        if (.not.this%search_on_) return
        if (.not.this%is_line_search_defined()) call this%set_default_line_search()
        opts%ls_max_evals = this%search_%get_max_fcn_evals()
        opts%ls_alpha = this%search_%get_scaling_factor()
        opts%ls_factor = this%search_%get_distance_factor()
    end subroutine

    subroutine bfgs_solve_one(this, fcn, x, fout, ib, args)
        class(bfgs), intent(inout) :: this
        class(fcnnvar_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), optional :: fout
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: counters
        type(nlh_scalar_ctx), target :: ctx
        type(c_funptr) :: grad_entry
        real(c_double), allocatable :: xwork(:)
        real(c_double) :: fmin
        integer(c_int) :: rc
        integer(int32) :: n

        n = fcn%get_variable_count()
        if (present(ib)) call behavior_clear(ib)
        call this%export_options(opts)
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! reference :614
        if (size(x) /= n) error stop NL_INVALID_INPUT_ERROR                      ! reference :615
        ctx%helper => fcn
        if (present(args)) ctx%args => args
        grad_entry = c_null_funptr
        if (fcn%is_gradient_defined()) grad_entry = c_funloc(nlh_gradfcn_trampoline)
        allocate(xwork(n), source = x)
        rc = nlh_bfgs_solve(nlh_default_handle(), opts, n, c_funloc(nlh_fcnnvar_trampoline), grad_entry, c_loc(ctx), &
            xwork, fmin, counters)
        x = xwork
        if (present(fout)) fout = fmin
        if (present(ib)) then
            call behavior_import(ib, counters)      ! jacobian_count = 0, gradient_count filled by the C side
            ib%converge_on_fcn = .false.            ! bfgs has no such test (reference :751-759)
        end if
        if (rc /= 0) error stop rc      ! as at :765-767
    end subroutine

    !> Extension: bfgs%solve (bfgs_solve, :557-770) on the objective 0.5 ||F(x)||^2 of every problem of a device model
    !> batch (forward-difference gradient) -- or, for a batch created from the user's own device function with ONE
    !> function per problem (device_model_batch%create_from_device_fcn, nfcn = 1), on that function itself: the launcher is
    !> the user's fcnnvar, its optional second launcher the gradient (set_gradient_fcn).  x(n, count) in / out,
    !> fout(count): the objective values, status(count): the code each solve would have stopped with (0: converged).
    subroutine bfgs_solve_many(this, model, x, fout, ib, status)
        class(bfgs), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:), optional :: fout
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: counters(:)
        integer(c_int32_t), allocatable :: outcome(:)
        real(c_double), allocatable :: xwork(:,:), fwork(:,:), fmin(:)
        integer(c_int) :: rc
        integer(int32) :: m, n, count, k

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        m = model%get_equation_count()
        n = model%get_variable_count()
        count = model%get_problem_count()
        if (any(shape(x) /= [n, count])) error stop NL_INVALID_INPUT_ERROR
        call this%export_options(opts)
        opts%print_status = 0
        allocate(counters(count), outcome(count), fwork(m, count), fmin(count))
        allocate(xwork(n, count), source = x)
        rc = nlh_dq_model_bfgs_solve(nlh_default_handle(), opts, model%c_handle(), xwork, fwork, fmin, counters, outcome)
        if (rc /= 0) error stop rc
        x = xwork
        if (present(fout)) fout = fmin
        if (present(status)) status = outcome
        if (present(ib)) then
            call behavior_import(ib, counters)
            do k = 1, count
                ib(k)%converge_on_fcn = .false.
            end do
        end if
    end subroutine
end module
