! least_squares_solver, constrained_equation_solver and constrained_least_squares_solver: the public types and
! bindings of src/nonlin_least_squares.f90:20-74.  Every `solve` here is a marshalling body: settings go into an
! nlh_options record, the caller's arrays into contiguous copies, and the iteration itself runs behind the C ABI
! (nlh_lm_solve / nlh_dq_model_lm_solve = lss_solve on the GPU, :118-391; nlh_cls_solve = cls_solve, :938-1176).
! The `error stop` the reference would perform is performed here, after the outputs have been copied back.
module nonlin_least_squares
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_multi_eqn_mult_var
    use nonlin_error_handling
    use nonlin_types
    use nonlin_hip_c
    use nonlin_shim_support
    implicit none
    private
    public :: least_squares_solver
    public :: constrained_equation_solver
    public :: constrained_least_squares_solver

    type, extends(equation_solver) :: least_squares_solver
        real(real64), private :: step_bound_ = 100.0d0           ! lmdif's `factor`
    contains
        procedure, public :: get_step_scaling_factor => lm_step_bound
        procedure, public :: set_step_scaling_factor => lm_put_step_bound
        procedure, public :: solve => lm_solve_one
        procedure, public :: solve_batch => lm_solve_many
    end type

    type, abstract, extends(least_squares_solver) :: constrained_equation_solver
        real(real64), private, allocatable, dimension(:) :: box_hi_
        real(real64), private, allocatable, dimension(:) :: box_lo_
    contains
        procedure, public :: get_upper_limits => box_hi
        procedure, public :: set_upper_limits => box_put_hi
        procedure, public :: get_lower_limits => box_lo
        procedure, public :: set_lower_limits => box_put_lo
        procedure, public :: apply_limits => box_project
    end type

    type, extends(constrained_equation_solver) :: constrained_least_squares_solver
        real(real64), private :: radius0_ = 1.0d0                ! initial trust-region radius
        real(real64), private :: dogleg_scale_ = 1.0d0
    contains
        procedure, public :: get_trust_region_radius => tr_radius0
        procedure, public :: set_trust_region_radius => tr_put_radius0
        procedure, public :: get_step_scaling_factor => tr_dogleg_scale
        procedure, public :: set_step_scaling_factor => tr_put_dogleg_scale
        procedure, public :: solve => tr_solve_one
        procedure, public :: solve_batch => tr_solve_many
    end type

contains
    ! ---- least_squares_solver ----------------------------------------------------------------------------------
    pure real(real64) function lm_step_bound(this)
        class(least_squares_solver), intent(in) :: this
        lm_step_bound = this%step_bound_
    end function

    subroutine lm_put_step_bound(this, x)        ! stored inside [0.1, 100] (reference setter, :80-115)
        class(least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%step_bound_ = into_interval(x, 0.1d0, 1.0d2)
    end subroutine

    subroutine lm_solve_one(this, fcn, x, fvec, ib, args)
        class(least_squares_solver), intent(inout) :: this
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
        integer(int32) :: m, n

        m = fcn%get_equation_count()
        n = fcn%get_variable_count()
        if (present(ib)) call behavior_clear(ib)
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! reference :188
        if (n > m) error stop NL_UNDERDEFINED_PROBLEM_ERROR                      ! reference :189
        call require_vector_sizes(size(x), size(fvec), n, m)

        call this%export_options(opts)
        opts%factor = this%step_bound_
        allocate(xwork(n), source = x)       ! the dummies may be strided sections: the C side wants contiguous memory
        allocate(fwork(m))
        if (fcn%is_device_model_defined()) then
            ! residual family registered with set_device_model: FD Jacobian, factorisation, lmpar and the trial
            ! evaluations all run on the GPU; outcome(1) is the code the reference would stop with
            onchip = fcn%device_model()
            rc = nlh_dq_model_lm_solve(nlh_default_handle(), opts, onchip%c_handle(), xwork, fwork, counters, outcome)
            if (rc == 0) rc = outcome(1)
        else
            ctx%helper => fcn
            if (present(args)) ctx%args => args
            jac_entry = c_null_funptr
            if (fcn%is_jacobian_defined()) jac_entry = c_funloc(nlh_jacfcn_trampoline)
            rc = nlh_lm_solve(nlh_default_handle(), opts, m, n, c_funloc(nlh_vecfcn_trampoline), jac_entry, &
                c_loc(ctx), xwork, fwork, counters(1))
        end if
        x = xwork
        fvec = fwork
        if (present(ib)) call behavior_import(ib, counters(1))
        if (rc /= 0) error stop rc      ! NL_CONVERGENCE_ERROR as at :388-390, or a library failure
    end subroutine

    !> Extension: least_squares_solver%solve (lss_solve, :118-391) for every problem of a device model batch in
    !> one call.  x(n, nprob) start points in / solutions out, fvec(m, nprob) residuals at the solutions,
    !> ib(nprob) the counters and flags of each solve, status(nprob) = 0 or the code the reference would
    !> `error stop` with for that problem (no process abort: the other problems are unaffected).
    subroutine lm_solve_many(this, model, x, fvec, ib, status)
        class(least_squares_solver), intent(inout) :: this
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
        integer(int32) :: m, n, count

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        m = model%get_equation_count()
        n = model%get_variable_count()
        count = model%get_problem_count()
        if (n > m) error stop NL_UNDERDEFINED_PROBLEM_ERROR
        if (any(shape(x) /= [n, count])) error stop 3
        if (any(shape(fvec) /= [m, count])) error stop 4
        if (present(ib)) then
            if (size(ib) /= count) error stop 5
        end if
        if (present(status)) then
            if (size(status) /= count) error stop 6
        end if
        call this%export_options(opts, quiet = .true.)
        opts%factor = this%step_bound_
        allocate(counters(count), outcome(count), fwork(m, count))
        allocate(xwork(n, count), source = x)
        rc = nlh_dq_model_lm_solve(nlh_default_handle(), opts, model%c_handle(), xwork, fwork, counters, outcome)
        if (rc /= 0) error stop rc      ! a library failure, not a per-problem outcome
        x = xwork
        fvec = fwork
        if (present(status)) status = outcome
        if (present(ib)) call behavior_import(ib, counters)
    end subroutine

    ! ---- constrained_equation_solver: the box -------------------------------------------------------------------
    !> A copy of v, or a zero-length vector while v has not been set (what the reference's getters return).
    pure function copy_or_empty(v) result(r)
        real(real64), intent(in), allocatable, dimension(:) :: v
        real(real64), allocatable, dimension(:) :: r
        if (allocated(v)) then
            r = v
        else
            r = [real(real64) ::]
        end if
    end function

    pure function box_hi(this) result(r)
        class(constrained_equation_solver), intent(in) :: this
        real(real64), allocatable, dimension(:) :: r
        r = copy_or_empty(this%box_hi_)
    end function

    pure function box_lo(this) result(r)
        class(constrained_equation_solver), intent(in) :: this
        real(real64), allocatable, dimension(:) :: r
        r = copy_or_empty(this%box_lo_)
    end function

    subroutine box_put_hi(this, x)
        class(constrained_equation_solver), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        this%box_hi_ = x                 ! (re)allocates to the shape of x
    end subroutine

    subroutine box_put_lo(this, x)
        class(constrained_equation_solver), intent(inout) :: this
        real(real64), intent(in), dimension(:) :: x
        this%box_lo_ = x
    end subroutine

    !> Projection of x onto the box, lower limits first; limits shorter than x bound only its leading entries.
    subroutine box_project(this, x)
        class(constrained_equation_solver), intent(in) :: this
        real(real64), intent(inout), dimension(:) :: x
        integer(int32) :: k
        if (allocated(this%box_lo_)) then
            k = min(size(x), size(this%box_lo_))
            where (x(:k) < this%box_lo_(:k)) x(:k) = this%box_lo_(:k)
        end if
        if (allocated(this%box_hi_)) then
            k = min(size(x), size(this%box_hi_))
            where (x(:k) > this%box_hi_(:k)) x(:k) = this%box_hi_(:k)
        end if
    end subroutine

    ! ---- constrained_least_squares_solver -----------------------------------------------------------------------
    pure real(real64) function tr_radius0(this)
        class(constrained_least_squares_solver), intent(in) :: this
        tr_radius0 = this%radius0_
    end function

    subroutine tr_put_radius0(this, x)           ! non-positive input selects 1 (reference setter, :898-910)
        class(constrained_least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%radius0_ = positive_or(x, 1.0d0)
    end subroutine

    pure real(real64) function tr_dogleg_scale(this)
        class(constrained_least_squares_solver), intent(in) :: this
        tr_dogleg_scale = this%dogleg_scale_
    end function

    subroutine tr_put_dogleg_scale(this, x)      ! same rule (:923-935)
        class(constrained_least_squares_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%dogleg_scale_ = positive_or(x, 1.0d0)
    end subroutine

    subroutine tr_solve_one(this, fcn, x, fvec, ib, args)
        class(constrained_least_squares_solver), intent(inout) :: this
        class(vecfcn_helper), intent(in), target :: fcn
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: fvec
        type(iteration_behavior), optional :: ib
        class(*), intent(inout), optional, target :: args

        type(nlh_options) :: opts
        type(nlh_iteration_behavior) :: counters
        type(nlh_callback_ctx), target :: ctx
        type(c_funptr) :: jac_entry
        real(c_double), allocatable :: xwork(:), fwork(:), lo(:), hi(:)
        integer(c_int) :: rc
        integer(int32) :: m, n

        m = fcn%get_equation_count()
        n = fcn%get_variable_count()
        if (present(ib)) call behavior_clear(ib)
        if (.not.fcn%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR    ! reference :988
        if (n > m) error stop NL_UNDERDEFINED_PROBLEM_ERROR                      ! reference :989
        call require_vector_sizes(size(x), size(fvec), n, m)
        ! limits of the wrong length are replaced by an unbounded box, and the replacement is stored (:999-1009)
        lo = this%get_lower_limits()
        if (size(lo) /= n) then
            lo = spread(-huge(1.0d0), 1, n)
            call this%set_lower_limits(lo)
        end if
        hi = this%get_upper_limits()
        if (size(hi) /= n) then
            hi = spread(huge(1.0d0), 1, n)
            call this%set_upper_limits(hi)
        end if

        call this%export_options(opts)
        allocate(xwork(n), source = x)
        allocate(fwork(m))
        if (fcn%is_device_model_defined()) then
            ! a device residual (set_device_model / set_device_fcn): the bounded dog-leg iteration entirely on the GPU
            block
                type(device_model_batch) :: onchip
                type(nlh_iteration_behavior) :: c1(1)
                integer(c_int32_t) :: outcome(1)
                onchip = fcn%device_model()
                rc = nlh_dq_model_cls_solve(nlh_default_handle(), opts, onchip%c_handle(), this%radius0_, this%dogleg_scale_, lo, hi, &
                    xwork, fwork, c1, outcome)
                if (rc == 0) rc = outcome(1)
                counters = c1(1)
            end block
        else
            ctx%helper => fcn
            if (present(args)) ctx%args => args
            jac_entry = c_null_funptr
            if (fcn%is_jacobian_defined()) jac_entry = c_funloc(nlh_jacfcn_trampoline)
            rc = nlh_cls_solve(nlh_default_handle(), opts, this%radius0_, this%dogleg_scale_, lo, hi, m, n, &
                c_funloc(nlh_vecfcn_trampoline), jac_entry, c_loc(ctx), xwork, fwork, counters)
        end if
        x = xwork
        fvec = fwork
        if (present(ib)) call behavior_import(ib, counters)
        if (rc /= 0) error stop rc      ! as at :1173-1175
    end subroutine

    !> Extension: constrained_least_squares_solver%solve (cls_solve, :938-1176) for every problem of a device model batch,
    !> all of them inside this solver's box.  Arguments as least_squares_solver%solve_batch.
    subroutine tr_solve_many(this, model, x, fvec, ib, status)
        class(constrained_least_squares_solver), intent(inout) :: this
        class(device_model_batch), intent(in) :: model
        real(real64), intent(inout), dimension(:,:) :: x
        real(real64), intent(out), dimension(:,:) :: fvec
        type(iteration_behavior), intent(out), dimension(:), optional :: ib
        integer(int32), intent(out), dimension(:), optional :: status

        type(nlh_options) :: opts
        type(nlh_iteration_behavior), allocatable :: counters(:)
        integer(c_int32_t), allocatable :: outcome(:)
        real(c_double), allocatable :: xwork(:,:), fwork(:,:), lo(:), hi(:)
        integer(c_int) :: rc
        integer(int32) :: m, n, count

        if (.not.model%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        m = model%get_equation_count()
        n = model%get_variable_count()
        count = model%get_problem_count()
        if (n > m) error stop NL_UNDERDEFINED_PROBLEM_ERROR
        if (any(shape(x) /= [n, count])) error stop 3
        if (any(shape(fvec) /= [m, count])) error stop 4
        lo = this%get_lower_limits()
        if (size(lo) /= n) then
            lo = spread(-huge(1.0d0), 1, n)
            call this%set_lower_limits(lo)
        end if
        hi = this%get_upper_limits()
        if (size(hi) /= n) then
            hi = spread(huge(1.0d0), 1, n)
            call this%set_upper_limits(hi)
        end if
        call this%export_options(opts)
        opts%print_status = 0
        allocate(counters(count), outcome(count), fwork(m, count))
        allocate(xwork(n, count), source = x)
        rc = nlh_dq_model_cls_solve(nlh_default_handle(), opts, model%c_handle(), this%radius0_, this%dogleg_scale_, lo, hi, &
            xwork, fwork, counters, outcome)
        if (rc /= 0) error stop rc
        x = xwork
        fvec = fwork
        if (present(status)) status = outcome
        if (present(ib)) call behavior_import(ib, counters)
    end subroutine
end module
