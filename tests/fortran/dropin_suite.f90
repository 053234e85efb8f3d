! Drop-in check of the Fortran shim: user code written against nonlin's API (use nonlin;
! vecfcn_helper / least_squares_solver / newton_solver / iteration_behavior) runs unchanged on
! the GPU library.  Problems are the reference's own test problems (tests/nonlin_test_solve.f90,
! tests/nonlin_test_jacobian.f90, README Example 2).  Every result is printed as
!   <case> <iter> <fcn> <jac> <cf> <cx> <cg> <hex x...>
! and compared with the CPU oracle by tests/test_gpu_fortran.py.
module dropin_problems
    use iso_fortran_env
    implicit none
contains
    subroutine fcn1(x, f, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        f(1) = x(1)**2 + x(2)**2 - 34.0d0
        f(2) = x(1)**2 - 2.0d0 * x(2)**2 - 7.0d0
    end subroutine

    subroutine jac1(x, j, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:,:) :: j
        class(*), intent(inout), optional :: args
        j(1,1) = 2.0d0 * x(1)
        j(2,1) = 2.0d0 * x(1)
        j(1,2) = 2.0d0 * x(2)
        j(2,2) = 2.0d0 * (-2.0d0 * x(2))
    end subroutine

    subroutine fcn1a(x, f, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        real(real64) :: a
        a = 0.0d0
        select type (args)
        type is (real(real64))
            a = args
        end select
        f(1) = x(1)**2 + x(2)**2 - 34.0d0
        f(2) = x(1)**2 - a * x(2)**2 - 7.0d0
    end subroutine

    subroutine fcn2(x, f, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        f(1) = x(2) - 10.0d0
        f(2) = x(1) * x(2) - 5.0d4
    end subroutine

    subroutine cubicfit(x, f, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        real(real64), dimension(21) :: xp, yp
        xp = [0.0d0, 0.1d0, 0.2d0, 0.3d0, 0.4d0, 0.5d0, 0.6d0, 0.7d0, 0.8d0, &
            0.9d0, 1.0d0, 1.1d0, 1.2d0, 1.3d0, 1.4d0, 1.5d0, 1.6d0, 1.7d0, &
            1.8d0, 1.9d0, 2.0d0]
        yp = [1.216737514d0, 1.250032542d0, 1.305579195d0, 1.040182335d0, &
            1.751867738d0, 1.109716707d0, 2.018141531d0, 1.992418729d0, &
            1.807916923d0, 2.078806005d0, 2.698801324d0, 2.644662712d0, &
            3.412756702d0, 4.406137221d0, 4.567156645d0, 4.999550779d0, &
            5.652854194d0, 6.784320119d0, 8.307936836d0, 8.395126494d0, &
            10.30252404d0]
        f = x(1) * xp**3 + x(2) * xp**2 + x(3) * xp + x(4) - yp
    end subroutine

    subroutine polar(x, f, args)
        real(real64), intent(in), dimension(:) :: x
        real(real64), intent(out), dimension(:) :: f
        class(*), intent(inout), optional :: args
        f(1) = x(1) * cos(x(2))
        f(2) = x(1) * sin(x(2))
    end subroutine

    function rosenbrock(x, args) result(f)
        real(real64), intent(in), dimension(:) :: x
        class(*), intent(inout), optional :: args
        real(real64) :: f, a, t
        a = 1.0d2
        if (present(args)) then
            select type (args)
            type is (real(real64))
                a = args
            end select
        end if
        t = x(2) - x(1) * x(1)
        f = a * (t * t) + (x(1) - 1.0d0) * (x(1) - 1.0d0)
    end function

    function beale(x, args) result(f)
        real(real64), intent(in), dimension(:) :: x
        class(*), intent(inout), optional :: args
        real(real64) :: f, a, b, c
        a = 1.5d0 - x(1) + x(1) * x(2)
        b = 2.25d0 - x(1) + x(1) * (x(2) * x(2))
        c = 2.625d0 - x(1) + x(1) * (x(2) * x(2) * x(2))
        f = a * a + b * b + c * c
    end function

end module

program dropin_suite
    use iso_fortran_env
    use nonlin
    use dropin_problems
    implicit none

    type(vecfcn_helper) :: obj
    procedure(vecfcn), pointer :: fcn
    procedure(jacobianfcn), pointer :: jac
    type(least_squares_solver) :: lm
    type(newton_solver) :: nt
    type(iteration_behavior) :: ib
    real(real64) :: x2(2), f2(2), x4(4), f21(21), a, numjac(2,2), ics(2)
    integer :: k
    character(len=64) :: mode

    ! `dropin_suite errstop_poly_get`: a coefficient of a polynomial that was never initialised -- the reference stops with
    ! NL_INVALID_OPERATION_ERROR (src/nonlin_polynomials.f90:399); tests/test_gpu_fortran.py checks the exit code.
    ! `... errstop_poly_index`: index out of range on an initialised one (NL_INDEX_OUT_OF_RANGE_ERROR, :402-405).
    if (command_argument_count() >= 1) then
        call get_command_argument(1, mode)
        block
            type(polynomial) :: pq
            if (trim(mode) == "errstop_poly_get") then
                print *, pq%get(1)
            else if (trim(mode) == "errstop_poly_index") then
                call pq%initialize(2)
                print *, pq%order(), pq%evaluate(1.5d0), size(pq%get_all())
                call pq%set(4, 1.0d0)
            end if
        end block
        stop 0
    end if

    ! README Example 2 (BASELINE config 1)
    fcn => cubicfit
    call obj%set_fcn(fcn, 21, 4)
    x4 = 1.0d0
    call lm%solve(obj, x4, f21, ib)
    call report("lm_readme", ib, x4)
    print '(A,F12.10)', "# c0: ", x4(4)
    print '(A,F7.5)', "# Max Residual: ", maxval(abs(f21))

    ! test_least_squares_1 / _4: fcn1, FD then analytic, two starts
    ics = [0.5d0, 1.0d0]
    fcn => fcn1
    do k = 1, 2
        call set_plain(obj, fcn)
        x2 = ics(k)
        call lm%solve(obj, x2, f2, ib)
        call report("lm_fcn1_fd", ib, x2)
    end do
    jac => jac1
    call obj%set_jacobian(jac)
    do k = 1, 2
        x2 = ics(k)
        call lm%solve(obj, x2, f2, ib)
        call report("lm_fcn1_an", ib, x2)
    end do

    ! test_least_squares_2: badly scaled, 1000 evaluations allowed
    block
        type(vecfcn_helper) :: o2
        type(least_squares_solver) :: lm2
        fcn => fcn2
        call o2%set_fcn(fcn, 2, 2)
        call lm2%set_max_fcn_evals(1000)
        do k = 1, 2
            x2 = ics(k)
            call lm2%solve(o2, x2, f2, ib)
            call report("lm_fcn2", ib, x2)
        end do
    end block

    ! args pass-through (class(*) -> real64), Newton with FD Jacobian (test_newton_3a)
    block
        type(vecfcn_helper) :: o3
        fcn => fcn1a
        call o3%set_fcn(fcn, 2, 2)
        a = 2.0d0
        x2 = 1.0d0
        call nt%solve(o3, x2, f2, ib, args = a)
        call report("newton_fcn1a_fd", ib, x2)
    end block

    ! test_newton_1: analytic Jacobian, line search on
    block
        type(vecfcn_helper) :: o4
        fcn => fcn1
        jac => jac1
        call o4%set_fcn(fcn, 2, 2)
        call o4%set_jacobian(jac)
        x2 = 1.0d0
        call nt%solve(o4, x2, f2, ib)
        call report("newton_fcn1_an", ib, x2)
    end block

    ! test_newton_2: line search off, FD Jacobian
    block
        type(vecfcn_helper) :: o5
        type(newton_solver) :: nt2
        fcn => fcn2
        call o5%set_fcn(fcn, 2, 2)
        call nt2%set_use_line_search(.false.)
        x2 = 0.5d0
        call nt2%solve(o5, x2, f2, ib)
        call report("newton_fcn2_nols", ib, x2)
    end block

    ! test_quasinewton_1 (analytic Jacobian), _2 (line search off, FD), _3a (args, FD)
    block
        type(vecfcn_helper) :: o7, o8, o9
        type(quasi_newton_solver) :: qn, qn2
        fcn => fcn1
        jac => jac1
        call o7%set_fcn(fcn, 2, 2)
        call o7%set_jacobian(jac)
        x2 = 1.0d0
        call qn%solve(o7, x2, f2, ib)
        call report("qn_fcn1_an", ib, x2)
        fcn => fcn2
        call o8%set_fcn(fcn, 2, 2)
        call qn2%set_use_line_search(.false.)
        x2 = 0.5d0
        call qn2%solve(o8, x2, f2, ib)
        call report("qn_fcn2_nols", ib, x2)
        fcn => fcn1a
        call o9%set_fcn(fcn, 2, 2)
        a = 2.0d0
        x2 = 0.5d0
        call qn%set_jacobian_interval(3)
        call qn%solve(o9, x2, f2, ib, args = a)
        call report("qn_fcn1a_fd_j3", ib, x2)
        if (qn%get_jacobian_interval() /= 3) error stop 99
    end block

    ! test_constrained_least_squares_1 (analytic Jacobian, infinite limits) and _bounds (FD Jacobian, box)
    block
        type(vecfcn_helper) :: o10, o11
        type(constrained_least_squares_solver) :: cs, cs2
        real(real64) :: big
        big = huge(big)
        fcn => fcn1
        jac => jac1
        call o10%set_fcn(fcn, 2, 2)
        call o10%set_jacobian(jac)
        call cs%set_upper_limits([big, big])
        call cs%set_lower_limits([-big, -big])
        x2 = 0.5d0
        call cs%solve(o10, x2, f2, ib)
        call report("cls_fcn1_an", ib, x2)
        call o11%set_fcn(fcn, 2, 2)
        call cs2%set_lower_limits([4.0d0, 2.0d0])
        call cs2%set_upper_limits([5.6d0, 3.6d0])
        x2 = 1.0d0
        call cs2%solve(o11, x2, f2, ib)
        call report("cls_fcn1_box", ib, x2)
    end block

    ! README Example 3: polynomial%fit on the Example 2 data
    block
        type(polynomial) :: pf
        real(real64) :: xp(21), yp(21), yc(21)
        integer :: i
        xp = [(0.1d0 * (i - 1), i = 1, 21)]
        xp = [0.0d0, 0.1d0, 0.2d0, 0.3d0, 0.4d0, 0.5d0, 0.6d0, 0.7d0, 0.8d0, 0.9d0, 1.0d0, 1.1d0, 1.2d0, 1.3d0, &
            1.4d0, 1.5d0, 1.6d0, 1.7d0, 1.8d0, 1.9d0, 2.0d0]
        yp = [1.216737514d0, 1.250032542d0, 1.305579195d0, 1.040182335d0, 1.751867738d0, 1.109716707d0, &
            2.018141531d0, 1.992418729d0, 1.807916923d0, 2.078806005d0, 2.698801324d0, 2.644662712d0, &
            3.412756702d0, 4.406137221d0, 4.567156645d0, 4.999550779d0, 5.652854194d0, 6.784320119d0, &
            8.307936836d0, 8.395126494d0, 10.30252404d0]
        yc = yp
        call pf%fit(xp, yp, 3)
        print '(A,I0,A,F12.10)', ("# poly c", i - 1, " = ", pf%get(i), i = 1, 4)
        print '(A,F7.5)', "# poly Max Residual: ", maxval(abs(pf%evaluate(xp) - yc))
        print '(A,4(1X,Z16.16))', "poly_readme 0 0 0 F F F", pf%get(1), pf%get(2), pf%get(3), pf%get(4)
    end block

    ! test_bfgs_1 (Rosenbrock from 0), test_bfgs_2 (Beale from 1), test_bfgs_3 (Rosenbrock, a passed through args)
    block
        type(bfgs) :: bs
        type(fcnnvar_helper) :: so
        procedure(fcnnvar), pointer :: sf
        real(real64) :: fo
        sf => rosenbrock
        call so%set_fcn(sf, 2)
        x2 = 0.0d0
        call bs%solve(so, x2, fo, ib)
        print '(A,3(1X,I0),3(1X,L1),2(1X,Z16.16))', "bfgs_rosen", ib%iter_count, ib%fcn_count, ib%gradient_count, &
            ib%converge_on_fcn, ib%converge_on_chng, ib%converge_on_zero_diff, x2(1), x2(2)
        sf => beale
        call so%set_fcn(sf, 2)
        x2 = 1.0d0
        call bs%solve(so, x2, fo, ib)
        print '(A,3(1X,I0),3(1X,L1),2(1X,Z16.16))', "bfgs_beale", ib%iter_count, ib%fcn_count, ib%gradient_count, &
            ib%converge_on_fcn, ib%converge_on_chng, ib%converge_on_zero_diff, x2(1), x2(2)
        sf => rosenbrock
        call so%set_fcn(sf, 2)
        a = 1.0d2
        x2 = 0.0d0
        call bs%solve(so, x2, fo, ib, args = a)
        print '(A,3(1X,I0),3(1X,L1),2(1X,Z16.16))', "bfgs_rosen_args", ib%iter_count, ib%fcn_count, ib%gradient_count, &
            ib%converge_on_fcn, ib%converge_on_chng, ib%converge_on_zero_diff, x2(1), x2(2)
    end block

    ! test_jacobian_1: vecfcn_helper%jacobian (no fv)
    block
        type(vecfcn_helper) :: o6
        fcn => polar
        call o6%set_fcn(fcn, 2, 2)
        x2 = [0.5d0, -0.5d0]
        call o6%jacobian(x2, numjac)
        print '(A,4(1X,Z16.16))', "jac_polar 0 0 0 0 0 0", numjac(1,1), numjac(2,1), numjac(1,2), numjac(2,2)
    end block

contains
    subroutine set_plain(o, f)
        type(vecfcn_helper), intent(out) :: o
        procedure(vecfcn), pointer, intent(in) :: f
        call o%set_fcn(f, 2, 2)
    end subroutine

    subroutine report(name, b, x)
        character(len=*), intent(in) :: name
        type(iteration_behavior), intent(in) :: b
        real(real64), intent(in) :: x(:)
        print '(A,3(1X,I0),3(1X,L1),*(1X,Z16.16))', name, b%iter_count, b%fcn_count, b%jacobian_count, &
            b%converge_on_fcn, b%converge_on_chng, b%converge_on_zero_diff, x
    end subroutine
end program
