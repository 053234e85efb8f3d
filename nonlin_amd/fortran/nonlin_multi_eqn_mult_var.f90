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

    type vecfcn_helper
        procedure(vecfcn), private, pointer, nopass :: fcn_ptr_ => null()
        procedure(jacobianfcn), private, pointer, nopass :: jac_ptr_ => null()
        integer(int32), private :: neqn_ = 0
        integer(int32), private :: nvar_ = 0
    contains
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
        x = associated(this%fcn_ptr_)
    end function

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
        if (this%is_fcn_defined()) then
            call this%fcn_ptr_(x, f, args)
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
