! fcnnvar, gradientfcn, fcnnvar_helper, equation_optimizer with the reference's public interface
! (src/nonlin_multi_var.f90); the bind(C) trampolines at the bottom let the C layer call the user's procedures.
module nonlin_multi_var
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_types
    use nonlin_error_handling
    use nonlin_hip_c, only : nlh_fd_gradient
    implicit none
    private
    public :: fcnnvar
    public :: gradientfcn
    public :: fcnnvar_helper
    public :: equation_optimizer
    public :: nonlin_optimize_fcn
    public :: nlh_scalar_ctx
    public :: nlh_fcnnvar_trampoline
    public :: nlh_gradfcn_trampoline

    interface
        function fcnnvar(x, args) result(f)
            use, intrinsic :: iso_fortran_env, only : real64
            real(real64), intent(in), dimension(:) :: x
            class(*), intent(inout), optional :: args
            real(real64) :: f
        end function

        subroutine gradientfcn(x, g, args)
            use, intrinsic :: iso_fortran_env, only : real64
            real(real64), intent(in), dimension(:) :: x
            real(real64), intent(out), dimension(:) :: g
            class(*), intent(inout), optional :: args
        end subroutine
    end interface

    type fcnnvar_helper
        private
        procedure(fcnnvar), private, pointer, nopass :: fcn_ptr_ => null()
        procedure(gradientfcn), private, pointer, nopass :: grad_ptr_ => null()
        integer(int32), private :: nvar_ = 0
    contains
        procedure, public :: fcn => obj_eval
        procedure, public :: is_fcn_defined => obj_has_fcn
        procedure, public :: set_fcn => obj_bind_fcn
        procedure, public :: get_variable_count => obj_nvar
        procedure, public :: set_gradient_fcn => obj_bind_grad
        procedure, public :: is_gradient_defined => obj_has_grad
        procedure, public :: gradient => obj_gradient
        procedure, public :: call_gradient => obj_user_grad
    end type

    !> What the C layer hands back to the trampolines through its void* ctx.
    type nlh_scalar_ctx
        class(fcnnvar_helper), pointer :: helper => null()
        class(*), pointer :: args => null()
    end type

    type, abstract :: equation_optimizer
        integer(int32), private :: max_evals_ = 500
        real(real64), private :: tol_ = 1.0d-12
        logical, private :: verbose_ = .false.
    contains
        procedure, public :: get_max_fcn_evals => opt_max_evals
        procedure, public :: set_max_fcn_evals => opt_put_max_evals
        procedure, public :: get_tolerance => opt_tol
        procedure, public :: set_tolerance => opt_put_tol
        procedure, public :: get_print_status => opt_verbose
        procedure, public :: set_print_status => opt_put_verbose
        procedure(nonlin_optimize_fcn), deferred, public, pass :: solve
    end type

    interface
        subroutine nonlin_optimize_fcn(this, fcn, x, fout, ib, args)
            use, intrinsic :: iso_fortran_env, only : real64
            use nonlin_types, only : iteration_behavior
            import equation_optimizer
            import fcnnvar_helper
            class(equation_optimizer), intent(inout) :: this
            class(fcnnvar_helper), intent(in), target :: fcn
            real(real64), intent(inout), dimension(:) :: x
            real(real64), intent(out), optional :: fout
            type(iteration_behavior), optional :: ib
            class(*), intent(inout), optional, target :: args
        end subroutine
    end interface
contains
    function obj_eval(this, x, args) result(f)               ! :81-89
        class(fcnnvar_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        class(*), intent(inout), optional :: args
        real(real64) :: f
        f = 0.0d0
        if (associated(this%fcn_ptr_)) f = this%fcn_ptr_(x, args)
    end function

    function obj_has_fcn(this) result(x)
        class(fcnnvar_helper), intent(in) :: this
        logical :: x
        x = associated(this%fcn_ptr_)
    end function

    subroutine obj_bind_fcn(this, fcn, nvar)                 ! :99-106
        class(fcnnvar_helper), intent(inout) :: this
        procedure(fcnnvar), intent(in), pointer :: fcn
        integer(int32), intent(in) :: nvar
        this%fcn_ptr_ => fcn
        this%nvar_ = nvar
    end subroutine

    function obj_nvar(this) result(n)
        class(fcnnvar_helper), intent(in) :: this
        integer(int32) :: n
        n = this%nvar_
    end function

    subroutine obj_bind_grad(this, fcn)
        class(fcnnvar_helper), intent(inout) :: this
        procedure(gradientfcn), pointer, intent(in) :: fcn
        this%grad_ptr_ => fcn
    end subroutine

    function obj_has_grad(this) result(x)
        class(fcnnvar_helper), intent(in) :: this
        logical :: x
        x = associated(this%grad_ptr_)
    end function

    subroutine obj_user_grad(this, x, g, args)
        class(fcnnvar_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: g
        class(*), intent(inout), optional :: args
        call this%grad_ptr_(x, g, args)
    end subroutine

    !> fcnnvar_helper%gradient (public behaviour of fnh_grad_fcn, src/nonlin_multi_var.f90:182-246): marshals to
    !> nlh_fd_gradient, which calls the user's gradient routine when one is bound and otherwise takes forward
    !> differences through the objective trampoline; x is perturbed and restored by the C side's working copy.
    subroutine obj_gradient(this, x, g, fv, args)
        class(fcnnvar_helper), intent(in), target :: this
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: g
        real(real64), intent(in), optional :: fv
        class(*), intent(inout), optional, target :: args

        type(nlh_scalar_ctx), target :: ctx
        real(c_double), allocatable :: xwork(:), gwork(:)
        real(c_double), target :: f0
        type(c_funptr) :: grad_entry
        type(c_ptr) :: f0_entry
        integer(c_int) :: rc
        integer(int32) :: n

        n = this%get_variable_count()
        if (size(x) /= n) error stop 2                          ! the reference's size checks stop with 2 and 3
        if (size(g) /= n) error stop 3
        if (.not.this%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        ctx%helper => this
        if (present(args)) ctx%args => args
        grad_entry = c_null_funptr
        if (this%is_gradient_defined()) grad_entry = c_funloc(nlh_gradfcn_trampoline)
        f0_entry = c_null_ptr
        if (present(fv)) then
            f0 = fv
            f0_entry = c_loc(f0)
        end if
        allocate(xwork(n), source = x)
        allocate(gwork(n))
        rc = nlh_fd_gradient(n, c_funloc(nlh_fcnnvar_trampoline), grad_entry, c_loc(ctx), xwork, f0_entry, gwork)
        if (rc /= 0) error stop rc
        g = gwork
    end subroutine

    pure function opt_max_evals(this) result(n)
        class(equation_optimizer), intent(in) :: this
        integer(int32) :: n
        n = this%max_evals_
    end function
    subroutine opt_put_max_evals(this, n)
        class(equation_optimizer), intent(inout) :: this
        integer(int32), intent(in) :: n
        this%max_evals_ = n
    end subroutine
    pure function opt_tol(this) result(x)
        class(equation_optimizer), intent(in) :: this
        real(real64) :: x
        x = this%tol_
    end function
    subroutine opt_put_tol(this, x)
        class(equation_optimizer), intent(inout) :: this
        real(real64), intent(in) :: x
        this%tol_ = x
    end subroutine
    pure function opt_verbose(this) result(x)
        class(equation_optimizer), intent(in) :: this
        logical :: x
        x = this%verbose_
    end function
    subroutine opt_put_verbose(this, x)
        class(equation_optimizer), intent(inout) :: this
        logical, intent(in) :: x
        this%verbose_ = x
    end subroutine

    function nlh_fcnnvar_trampoline(ctx, n, x) bind(C) result(f)
        type(c_ptr), value :: ctx
        integer(c_int32_t), value :: n
        real(c_double), intent(in) :: x(n)
        real(c_double) :: f
        type(nlh_scalar_ctx), pointer :: c
        call c_f_pointer(ctx, c)
        if (associated(c%args)) then
            f = c%helper%fcn(x, c%args)
        else
            f = c%helper%fcn(x)
        end if
    end function

    subroutine nlh_gradfcn_trampoline(ctx, n, x, g) bind(C)
        type(c_ptr), value :: ctx
        integer(c_int32_t), value :: n
        real(c_double), intent(in) :: x(n)
        real(c_double), intent(out) :: g(n)
        type(nlh_scalar_ctx), pointer :: c
        call c_f_pointer(ctx, c)
        if (associated(c%args)) then
            call c%helper%call_gradient(x, g, c%args)
        else
            call c%helper%call_gradient(x, g)
        end if
    end subroutine
end module
