! fcnnvar, gradientfcn, fcnnvar_helper, equation_optimizer with the reference's public interface
! (src/nonlin_multi_var.f90); the bind(C) trampolines at the bottom let the C layer call the user's procedures.
module nonlin_multi_var
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin_types
    use nonlin_error_handling
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
        procedure(fcnnvar), private, pointer, nopass :: m_fcn => null()
        procedure(gradientfcn), private, pointer, nopass :: m_grad => null()
        integer(int32), private :: m_nvar = 0
    contains
        procedure, public :: fcn => fnh_fcn
        procedure, public :: is_fcn_defined => fnh_is_fcn_defined
        procedure, public :: set_fcn => fnh_set_fcn
        procedure, public :: get_variable_count => fnh_get_nvar
        procedure, public :: set_gradient_fcn => fnh_set_grad
        procedure, public :: is_gradient_defined => fnh_is_grad_defined
        procedure, public :: gradient => fnh_grad_fcn
        procedure, public :: call_gradient => fnh_call_grad
    end type

    !> What the C layer hands back to the trampolines through its void* ctx.
    type nlh_scalar_ctx
        class(fcnnvar_helper), pointer :: helper => null()
        class(*), pointer :: args => null()
    end type

    type, abstract :: equation_optimizer
        integer(int32), private :: m_maxEval = 500
        real(real64), private :: m_tol = 1.0d-12
        logical, private :: m_printStatus = .false.
    contains
        procedure, public :: get_max_fcn_evals => oe_get_max_eval
        procedure, public :: set_max_fcn_evals => oe_set_max_eval
        procedure, public :: get_tolerance => oe_get_tol
        procedure, public :: set_tolerance => oe_set_tol
        procedure, public :: get_print_status => oe_get_print_status
        procedure, public :: set_print_status => oe_set_print_status
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
    function fnh_fcn(this, x, args) result(f)               ! :81-89
        class(fcnnvar_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        class(*), intent(inout), optional :: args
        real(real64) :: f
        f = 0.0d0
        if (associated(this%m_fcn)) f = this%m_fcn(x, args)
    end function

    function fnh_is_fcn_defined(this) result(x)
        class(fcnnvar_helper), intent(in) :: this
        logical :: x
        x = associated(this%m_fcn)
    end function

    subroutine fnh_set_fcn(this, fcn, nvar)                 ! :99-106
        class(fcnnvar_helper), intent(inout) :: this
        procedure(fcnnvar), intent(in), pointer :: fcn
        integer(int32), intent(in) :: nvar
        this%m_fcn => fcn
        this%m_nvar = nvar
    end subroutine

    function fnh_get_nvar(this) result(n)
        class(fcnnvar_helper), intent(in) :: this
        integer(int32) :: n
        n = this%m_nvar
    end function

    subroutine fnh_set_grad(this, fcn)
        class(fcnnvar_helper), intent(inout) :: this
        procedure(gradientfcn), pointer, intent(in) :: fcn
        this%m_grad => fcn
    end subroutine

    function fnh_is_grad_defined(this) result(x)
        class(fcnnvar_helper), intent(in) :: this
        logical :: x
        x = associated(this%m_grad)
    end function

    subroutine fnh_call_grad(this, x, g, args)
        class(fcnnvar_helper), intent(in) :: this
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: g
        class(*), intent(inout), optional :: args
        call this%m_grad(x, g, args)
    end subroutine

    ! fnh_grad_fcn, :182-246: n + 1 evaluations of a scalar; the work is the user's function, so it stays here.
    subroutine fnh_grad_fcn(this, x, g, fv, args)
        class(fcnnvar_helper), intent(in) :: this
        real(real64), intent(inout), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: g
        real(real64), intent(in), optional :: fv
        class(*), intent(inout), optional :: args
        integer(int32) :: j, n, flag
        real(real64) :: eps, h, temp, f, f1
        n = this%get_variable_count()
        flag = 0
        if (size(x) /= n) then
            flag = 2
        else if (size(g) /= n) then
            flag = 3
        end if
        if (flag /= 0) error stop flag
        if (.not.this%is_fcn_defined()) error stop NL_UNDEFINED_FUNCTION_ERROR
        if (this%is_gradient_defined()) then
            call this%m_grad(x, g, args)
        else
            if (present(fv)) then
                f = fv
            else
                f = this%fcn(x, args)
            end if
            eps = sqrt(epsilon(eps))
            do j = 1, n
                temp = x(j)
                h = eps * abs(temp)
                if (h == 0.0d0) h = eps
                x(j) = temp + h
                f1 = this%fcn(x, args)
                x(j) = temp
                g(j) = (f1 - f) / h
            end do
        end if
    end subroutine

    pure function oe_get_max_eval(this) result(n)
        class(equation_optimizer), intent(in) :: this
        integer(int32) :: n
        n = this%m_maxEval
    end function
    subroutine oe_set_max_eval(this, n)
        class(equation_optimizer), intent(inout) :: this
        integer(int32), intent(in) :: n
        this%m_maxEval = n
    end subroutine
    pure function oe_get_tol(this) result(x)
        class(equation_optimizer), intent(in) :: this
        real(real64) :: x
        x = this%m_tol
    end function
    subroutine oe_set_tol(this, x)
        class(equation_optimizer), intent(inout) :: this
        real(real64), intent(in) :: x
        this%m_tol = x
    end subroutine
    pure function oe_get_print_status(this) result(x)
        class(equation_optimizer), intent(in) :: this
        logical :: x
        x = this%m_printStatus
    end function
    subroutine oe_set_print_status(this, x)
        class(equation_optimizer), intent(inout) :: this
        logical, intent(in) :: x
        this%m_printStatus = x
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
