! Problem-definition layer with the reference's public names and signatures
! (src/nonlin_multi_eqn_mult_var.f90): vecfcn, jacobianfcn, vecfcn_helper, equation_solver,
! nonlin_solver.  vecfcn_helper%jacobian marshals to nlh_fd_jacobian (GPU column write);
! the bind(C) trampolines at the bottom let the C layer call the user's Fortran procedures.
module nonlin_multi_eqn_mult_var
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_types
    use nonlin_error_handling
    use nonlin_hip_c
    implicit none
    private
    public :: vecfcn
    public :: jacobianfcn
    public :: vecfcn_helper
    public :: equation_solver
    public :: nonlin_solver
    public :: nlh_callback_ctx
    public :: device_model_batch
    public :: NLH_MODEL_DENSE_QUADRATIC
    public :: NLH_FACTOR_AUTO, NLH_FACTOR_QR, NLH_FACTOR_EXACT     ! values of equation_solver%factor_policy (from nonlin_hip_c)
    public :: nlh_use_devices
    public :: nlh_vecfcn_trampoline
    public :: nlh_jacfcn_trampoline

    interface
        subroutine vecfcn(x, f, args)
            use, intrinsic :: iso_fortran_env, only : real64
            real(real64), intent(in), dimension(:) :: x
            real(real64), intent(out), dimension(:) :: f
            class(*), intent(inout), optional :: args
        end subroutine

        subroutine jacobianfcn(x, jac, args)
            use, intrinsic :: iso_fortran_env, only : real64
            real(real64), intent(in), dimension(:) :: x
            real(real64), intent(out), dimension(:,:) :: jac
            class(*), intent(inout), optional :: args
        end subroutine
    end interface

    !> Registered device residual families (SURVEY.md 8(d)): 1 = dense quadratic,
    !> r_i = (u_i + gamma u_i u_i) - b_i with u = A x accumulated in ascending column order.
    integer(int32), parameter :: NLH_MODEL_DENSE_QUADRATIC = 1

    !> Extension (no counterpart in the reference): the data of nprob independent problems of a registered
    !> residual family, resident on the GPU.  What least_squares_solver%solve_batch / newton_solver%solve_batch
    !> solve in one call, and what vecfcn_helper%set_device_model wraps for a single problem.
    type device_model_batch
        type(c_ptr), private :: model_ = c_null_ptr
        integer(int32), private :: nprob_ = 0
        integer(int32), private :: neqn_ = 0
        integer(int32), private :: nvar_ = 0
        logical, private :: analytic_ = .false.
    contains
        procedure, public :: create => dmb_create
        procedure, public :: create_from_device_fcn => dmb_create_fcn
        procedure, public :: destroy => dmb_destroy
        procedure, public :: is_defined => dmb_defined
        procedure, public :: get_problem_count => dmb_nprob
        procedure, public :: get_equation_count => dmb_neqn
        procedure, public :: get_variable_count => dmb_nvar
        procedure, public :: uses_analytic_jacobian => dmb_analytic
        procedure, public :: evaluate => dmb_eval
        procedure, public :: c_handle => dmb_handle
    end type

    type vecfcn_helper
        procedure(vecfcn), private, pointer, nopass :: fcn_ptr_ => null()
        procedure(jacobianfcn), private, pointer, nopass :: jac_ptr_ => null()
        integer(int32), private :: neqn_ = 0
        integer(int32), private :: nvar_ = 0
        type(device_model_batch), private :: model_      ! set_device_model: one problem on the device
    contains
        procedure, public :: set_device_model => helper_bind_model
        procedure, public :: set_device_fcn => helper_bind_device_fcn
        procedure, public :: clear_device_model => helper_drop_model
        procedure, public :: is_device_model_defined => helper_has_model
        procedure, public :: device_model => helper_model
        procedure, public :: set_fcn => helper_bind_fcn
        procedure, public :: set_jacobian => helper_bind_jac
        procedure, public :: is_fcn_defined => helper_has_fcn
        procedure, public :: is_jacobian_defined => helper_has_jac
        procedure, public :: fcn => helper_eval
        procedure, public :: jacobian => helper_jacobian
        procedure, public :: get_equation_count => helper_neqn
        procedure, public :: get_variable_count => helper_nvar
        procedure, public :: call_jacobian => helper_user_jac
    end type

    !> What the C layer hands back to the trampolines through its void* ctx.
    type nlh_callback_ctx
        class(vecfcn_helper), pointer :: helper => null()
        class(*), pointer :: args => null()
    end type

    type, abstract :: equation_solver
        integer(int32), private :: max_evals_ = 100
        real(real64), private :: ftol_ = 1.0d-8
        real(real64), private :: xtol_ = 1.0d-12
        real(real64), private :: gtol_ = 1.0d-12
        logical, private :: verbose_ = .false.
        !> Extension: NLH_FACTOR_EXACT (default; reference operation order, bit-identical
        !> results), NLH_FACTOR_AUTO (J^T J + Cholesky) or NLH_FACTOR_QR.
        integer(int32), public :: factor_policy = NLH_FACTOR_EXACT
    contains
        procedure, public :: get_max_fcn_evals => cfg_max_evals
        procedure, public :: set_max_fcn_evals => cfg_put_max_evals
        procedure, public :: get_fcn_tolerance => cfg_ftol
        procedure, public :: set_fcn_tolerance => cfg_put_ftol
        procedure, public :: get_var_tolerance => cfg_xtol
        procedure, public :: set_var_tolerance => cfg_put_xtol
        procedure, public :: get_gradient_tolerance => cfg_gtol
        procedure, public :: set_gradient_tolerance => cfg_put_gtol
        procedure, public :: get_print_status => cfg_verbose
        procedure, public :: set_print_status => cfg_put_verbose
        procedure, public :: export_options => cfg_export
        procedure(nonlin_solver), deferred, public, pass :: solve
    end type

    interface
        subroutine nonlin_solver(this, fcn, x, fvec, ib, args)
            use, intrinsic :: iso_fortran_env, only : real64
            use nonlin_types, only : iteration_behavior
            import equation_solver
            import vecfcn_helper
            class(equation_solver), intent(inout) :: this
            class(vecfcn_helper), intent(in), target :: fcn
            real(real64), intent(inout), dimension(:) :: x
            real(real64), intent(out), dimension(:) :: fvec
            type(iteration_behavior), optional :: ib
            class(*), intent(inout), optional, target :: args
        end subroutine
    end interface

contains
    subroutine helper_bind_fcn(this, fcn, nfcn, nvar)
        class(vecfcn_helper), intent(inout) :: this
        procedure(vecfcn), intent(in), pointer :: fcn
        integer(int32), intent(in) :: nfcn
        integer(int32), intent(in) :: nvar
        this%fcn_ptr_ => fcn
        this%neqn_ = nfcn
        this%nvar_ = nvar
    end subroutine

    subroutine helper_bind_jac(this, jac)
        class(vecfcn_helper), intent(inout) :: this
        procedure(jacobianfcn), intent(in), pointer :: jac
        this%jac_ptr_ => jac
    end subroutine

    function helper_has_fcn(this) result(x)
        class(vecfcn_helper), intent(in) :: this
        logical :: x
        x = associated(this%fcn_ptr_) .or. this%model_%is_defined()
    end function

    !> Extension: instead of a host procedure, the residual is a registered device model (kind =
    !> NLH_MODEL_DENSE_QUADRATIC: a(m,n), b(m), gamma).  solver%solve then runs the whole iteration on the GPU
    !> (no host callbacks); analytic = .true. makes newton_solver use the model's own Jacobian, as set_jacobian would.
    subroutine helper_bind_model(this, kind, a, b, gamma, analytic)
        class(vecfcn_helper), intent(inout) :: this
        integer(int32), intent(in) :: kind
        real(real64), intent(in), dimension(:,:) :: a
        real(real64), intent(in), dimension(:) :: b
        real(real64), intent(in) :: gamma
        logical, intent(in), optional :: analytic
        real(real64), allocatable :: a3(:,:,:), b2(:,:)
        allocate(a3(size(a, 1), size(a, 2), 1), b2(size(b), 1))
        a3(:,:,1) = a
        b2(:,1) = b
        call this%model_%create(kind, a3, b2, gamma, analytic)
        this%neqn_ = size(a, 1)
        this%nvar_ = size(a, 2)
    end subroutine

    !> Extension: the residual is the USER'S OWN device function -- the device form of set_fcn (reference :126-140).  fcn is a
    !> launcher with the C signature nlh_device_vecfcn of include/nonlin_hip.h (a host procedure, bind(C) or written in
    !> C / HIP, that enqueues the user's kernel on the stream it is handed), ctx whatever that launcher needs (its device
    !> data); jac, optional, a launcher for the analytic Jacobian (the device form of set_jacobian, :143-153).
    !> solver%solve(obj, x, fvec, ib) stays the reference's call; the whole iteration then runs on the GPU.
    subroutine helper_bind_device_fcn(this, fcn, ctx, nfcn, nvar, jac)
        class(vecfcn_helper), intent(inout) :: this
        type(c_funptr), intent(in) :: fcn
        type(c_ptr), intent(in) :: ctx
        integer(int32), intent(in) :: nfcn
        integer(int32), intent(in) :: nvar
        type(c_funptr), intent(in), optional :: jac
        call this%model_%create_from_device_fcn(fcn, ctx, 1, nfcn, nvar, jac)
        this%neqn_ = nfcn
        this%nvar_ = nvar
    end subroutine

    subroutine helper_drop_model(this)
        class(vecfcn_helper), intent(inout) :: this
        call this%model_%destroy()
    end subroutine

    function helper_has_model(this) result(x)
        class(vecfcn_helper), intent(in) :: this
        logical :: x
        x = this%model_%is_defined() .and. .not.associated(this%fcn_ptr_)
    end function

    function helper_model(this) result(md)
        class(vecfcn_helper), intent(in) :: this
        type(device_model_batch) :: md
        md = this%model_
    end function

    ! ---- device_model_batch -------------------------------------------------------------------
    subroutine dmb_create(this, kind, a, b, gamma, analytic)
        class(device_model_batch), intent(inout) :: this
        integer(int32), intent(in) :: kind
        real(real64), intent(in), dimension(:,:,:) :: a      ! (m, n, nprob)
        real(real64), intent(in), dimension(:,:) :: b        ! (m, nprob)
        real(real64), intent(in) :: gamma
        logical, intent(in), optional :: analytic
        integer(c_int) :: rc
        real(c_double), allocatable :: ac(:,:,:), bc(:,:)
        if (kind /= NLH_MODEL_DENSE_QUADRATIC) error stop NL_INVALID_INPUT_ERROR
        if (size(b, 1) /= size(a, 1) .or. size(b, 2) /= size(a, 3)) error stop NL_ARRAY_SIZE_ERROR
        call this%destroy()
        ac = a                                               ! contiguous copies: the dummies may be sections
        bc = b
        if (c_associated(nlh_default_device_set())) then         ! nlh_use_devices / NLH_DEVICES: dealt over several GPUs
            rc = nlh_dq_model_create_on(nlh_default_device_set(), int(size(a, 3), c_int32_t), int(size(a, 1), c_int32_t), &
                int(size(a, 2), c_int32_t), ac, bc, gamma, this%model_)
        else
            rc = nlh_dq_model_create(nlh_default_handle(), int(size(a, 3), c_int32_t), int(size(a, 1), c_int32_t), &
                int(size(a, 2), c_int32_t), ac, bc, gamma, this%model_)
        end if
        if (rc /= 0) error stop rc
        this%neqn_ = size(a, 1)
        this%nvar_ = size(a, 2)
        this%nprob_ = size(a, 3)
        this%analytic_ = .false.
        if (present(analytic)) this%analytic_ = analytic
    end subroutine

    !> nprob problems of the user's own device residual family (launchers: see vecfcn_helper%set_device_fcn); the user's
    !> kernel tells the problems apart by the index the launcher is handed for every point (0-based, position in x(:, k)).
    subroutine dmb_create_fcn(this, fcn, ctx, nprob, nfcn, nvar, jac)
        class(device_model_batch), intent(inout) :: this
        type(c_funptr), intent(in) :: fcn
        type(c_ptr), intent(in) :: ctx
        integer(int32), intent(in) :: nprob, nfcn, nvar
        type(c_funptr), intent(in), optional :: jac
        type(c_funptr) :: jentry
        integer(c_int) :: rc
        if (.not.c_associated(fcn)) error stop NL_UNDEFINED_FUNCTION_ERROR
        call this%destroy()
        jentry = c_null_funptr
        if (present(jac)) jentry = jac
        rc = nlh_device_fcn_model_create(int(nprob, c_int32_t), int(nfcn, c_int32_t), int(nvar, c_int32_t), fcn, jentry, ctx, &
            this%model_)
        if (rc /= 0) error stop rc
        this%neqn_ = nfcn
        this%nvar_ = nvar
        this%nprob_ = nprob
        this%analytic_ = c_associated(jentry)
    end subroutine

    subroutine dmb_destroy(this)
        class(device_model_batch), intent(inout) :: this
        if (c_associated(this%model_)) call nlh_dq_model_destroy(this%model_)
        this%model_ = c_null_ptr
        this%nprob_ = 0; this%neqn_ = 0; this%nvar_ = 0
    end subroutine

    pure function dmb_defined(this) result(x)
        class(device_model_batch), intent(in) :: this
        logical :: x
        x = c_associated(this%model_)
    end function

    pure function dmb_nprob(this) result(n)
        class(device_model_batch), intent(in) :: this
        integer(int32) :: n
        n = this%nprob_
    end function

    pure function dmb_neqn(this) result(n)
        class(device_model_batch), intent(in) :: this
        integer(int32) :: n
        n = this%neqn_
    end function

    pure function dmb_nvar(this) result(n)
        class(device_model_batch), intent(in) :: this
        integer(int32) :: n
        n = this%nvar_
    end function

    pure function dmb_analytic(this) result(x)
        class(device_model_batch), intent(in) :: this
        logical :: x
        x = this%analytic_
    end function

    function dmb_handle(this) result(h)
        class(device_model_batch), intent(in) :: this
        type(c_ptr) :: h
        h = this%model_
    end function

    !> vecfcn of every problem: f(:,k) = F_k(x(:,k)).
    subroutine dmb_eval(this, x, f)
        class(device_model_batch), intent(in) :: this
        real(real64), intent(in), dimension(:,:) :: x        ! (n, nprob)
        real(real64), intent(out), dimension(:,:) :: f       ! (m, nprob)
        integer(c_int) :: rc
        real(c_double), allocatable :: xc(:,:), fc(:,:)
        if (.not.this%is_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        if (size(x, 1) /= this%nvar_ .or. size(x, 2) /= this%nprob_) error stop NL_ARRAY_SIZE_ERROR
        if (size(f, 1) /= this%neqn_ .or. size(f, 2) /= this%nprob_) error stop NL_ARRAY_SIZE_ERROR
        xc = x
        allocate(fc(this%neqn_, this%nprob_))
        rc = nlh_dq_model_eval(nlh_default_handle(), this%model_, xc, fc)
        if (rc /= 0) error stop rc
        f = fc
    end subroutine

    function helper_has_jac(this) result(x)
        class(vecfcn_helper), intent(in) :: this
        logical :: x
        x = associated(this%jac_ptr_)
    end function

    subroutine helper_eval(this, x, f, args)
        class(vecfcn_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        real(real64), allocatable :: x2(:,:), f2(:,:)
        if (associated(this%fcn_ptr_)) then
            call this%fcn_ptr_(x, f, args)
        else if (this%model_%is_defined()) then                 ! device model: one evaluation on the GPU
            allocate(x2(size(x), 1), f2(size(f), 1))
            x2(:,1) = x
            call this%model_%evaluate(x2, f2)
            f = f2(:,1)
        end if
    end subroutine

    !> Invokes the user's analytic Jacobian routine (used by the C-side trampoline).
    subroutine helper_user_jac(this, x, jac, args)
        class(vecfcn_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:,:) :: jac
        class(*), intent(inout), optional :: args
        if (associated(this%jac_ptr_)) call this%jac_ptr_(x, jac, args)
    end subroutine

    !> helper_jacobian (src/nonlin_multi_eqn_mult_var.f90:198-277): analytic dispatch, or n
    !> perturbed evaluations on the host + the (f1 - f0)/h column write on the GPU.
    subroutine helper_jacobian(this, x, jac, fv, args)
        class(vecfcn_helper), intent(in), target :: this
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:,:) :: jac
        real(real64), intent(in), dimension(:), optional, target :: fv
        class(*), intent(inout), optional, target :: args

        integer(int32) :: m, n, flag
        integer(c_int) :: rc
        type(nlh_callback_ctx), target :: ctx
        real(c_double), allocatable, target :: xc(:), jc(:,:), fvc(:)
        type(c_funptr) :: cjac
        type(c_ptr) :: fvp

        m = this%get_equation_count()
        n = this%get_variable_count()
        flag = 0
        if (size(x) /= n) then
            flag = 2
        else if (size(jac, 1) /= m .or. size(jac, 2) /= n) then
            flag = 3
        end if
        if (flag /= 0) error stop flag
        if (.not.this%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR

        ctx%helper => this
        if (present(args)) ctx%args => args
        allocate(xc(n), jc(m, n))
        xc = x
        cjac = c_null_funptr
        if (associated(this%jac_ptr_)) cjac = c_funloc(nlh_jacfcn_trampoline)
        fvp = c_null_ptr
        if (present(fv)) then
            allocate(fvc(m))
            fvc = fv(1:m)
            fvp = c_loc(fvc)
        end if
        rc = nlh_fd_jacobian(nlh_default_handle(), m, n, c_funloc(nlh_vecfcn_trampoline), cjac, &
            c_loc(ctx), xc, fvp, jc)
        if (rc /= 0) error stop rc
        x = xc
        jac = jc
    end subroutine

    function helper_neqn(this) result(n)
        class(vecfcn_helper), intent(in) :: this
        integer(int32) :: n
        n = this%neqn_
    end function

    function helper_nvar(this) result(n)
        class(vecfcn_helper), intent(in) :: this
        integer(int32) :: n
        n = this%nvar_
    end function

    pure function cfg_max_evals(this) result(n)
        class(equation_solver), intent(in) :: this
        integer(int32) :: n
        n = this%max_evals_
    end function

    subroutine cfg_put_max_evals(this, n)
        class(equation_solver), intent(inout) :: this
        integer(int32), intent(in) :: n
        this%max_evals_ = n
    end subroutine

    pure function cfg_ftol(this) result(x)
        class(equation_solver), intent(in) :: this
        real(real64) :: x
        x = this%ftol_
    end function

    subroutine cfg_put_ftol(this, x)
        class(equation_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%ftol_ = x
    end subroutine

    pure function cfg_xtol(this) result(x)
        class(equation_solver), intent(in) :: this
        real(real64) :: x
        x = this%xtol_
    end function

    subroutine cfg_put_xtol(this, x)
        class(equation_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%xtol_ = x
    end subroutine

    pure function cfg_gtol(this) result(x)
        class(equation_solver), intent(in) :: this
        real(real64) :: x
        x = this%gtol_
    end function

    subroutine cfg_put_gtol(this, x)
        class(equation_solver), intent(inout) :: this
        real(real64), intent(in) :: x
        this%gtol_ = x
    end subroutine

    pure function cfg_verbose(this) result(x)
        class(equation_solver), intent(in) :: this
        logical :: x
        x = this%verbose_
    end function

    subroutine cfg_put_verbose(this, x)
        class(equation_solver), intent(inout) :: this
        logical, intent(in) :: x
        this%verbose_ = x
    end subroutine

    !> Extension used by every solve body of the shim: the C ABI's option record with this solver's settings
    !> (library defaults for everything the base type does not hold); quiet = .true. suppresses print_status.
    subroutine cfg_export(this, opts, quiet)
        class(equation_solver), intent(in) :: this
        type(nlh_options), intent(out) :: opts
        logical, intent(in), optional :: quiet
        call nlh_default_options(opts)
        opts%max_evals = this%max_evals_
        opts%ftol = this%ftol_
        opts%xtol = this%xtol_
        opts%gtol = this%gtol_
        opts%print_status = merge(1, 0, this%verbose_)
        if (present(quiet)) then
            if (quiet) opts%print_status = 0
        end if
        opts%factor_policy = this%factor_policy
    end subroutine

    ! ---- trampolines: the C layer's nlh_vecfcn / nlh_jacfcn --------------------------------
    subroutine nlh_vecfcn_trampoline(ctx, n, x, m, f) bind(C)
        type(c_ptr), value :: ctx
        integer(c_int32_t), value :: n, m
        real(c_double), intent(in) :: x(n)
        real(c_double), intent(out) :: f(m)
        type(nlh_callback_ctx), pointer :: c
        call c_f_pointer(ctx, c)
        if (associated(c%args)) then
            call c%helper%fcn(x, f, c%args)
        else
            call c%helper%fcn(x, f)
        end if
    end subroutine

    subroutine nlh_jacfcn_trampoline(ctx, n, x, m, jac) bind(C)
        type(c_ptr), value :: ctx
        integer(c_int32_t), value :: n, m
        real(c_double), intent(in) :: x(n)
        real(c_double), intent(out) :: jac(m, n)
        type(nlh_callback_ctx), pointer :: c
        call c_f_pointer(ctx, c)
        if (associated(c%args)) then
            call c%helper%call_jacobian(x, jac, c%args)
        else
            call c%helper%call_jacobian(x, jac)
        end if
    end subroutine
end module
