! User-side program for the OPEN device-residual path of the drop-in layer: the user's own residual family lives in the
! user's own shared object (tests/device_model/user_models.hip: a HIP kernel + a launcher with the C signature
! nlh_device_vecfcn of include/nonlin_hip.h) and is handed to nonlin's types the way set_fcn hands in a host procedure
! (reference src/nonlin_multi_eqn_mult_var.f90:126-140):
!     call obj%set_device_fcn(c_funloc(lorentz_launch), ctx, m, n)     ! instead of obj%set_fcn(fcn, m, n)
!     call solver%solve(obj, x, fvec, ib)                              ! the reference's call, unchanged
! and, for many problems at once, device_model_batch%create_from_device_fcn + solver%solve_batch.
! Reads the spectra written by tests/test_gpu_fortran.py (stream binary: nprob, m, n (int32), t(m,nprob), y(m,nprob),
! x0(n,nprob)), then c(nprob), xs(nq,nprob) for the square family; prints status, counts, flags and the bit patterns of x.
program device_fcn_suite
    use iso_fortran_env
    use, intrinsic :: iso_c_binding
    use nonlin
    implicit none

    interface   ! the user's library (tests/device_model/user_models.hip)
        function lorentz_create(nprob, m, t, y) bind(C, name="lorentz_create") result(ctx)
            import :: c_ptr, c_int32_t, c_double
            integer(c_int32_t), value :: nprob, m
            real(c_double), intent(in) :: t(*), y(*)
            type(c_ptr) :: ctx
        end function
        subroutine lorentz_destroy(ctx) bind(C, name="lorentz_destroy")
            import :: c_ptr
            type(c_ptr), value :: ctx
        end subroutine
        function lorentz_launch(ctx, stream, npoints, dprob, n, dx, m, df) bind(C, name="lorentz_launch") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: ctx, stream, dprob, dx, df
            integer(c_int32_t), value :: npoints, n, m
            integer(c_int) :: rc
        end function
        function btri_create(nprob, c) bind(C, name="btri_create") result(ctx)
            import :: c_ptr, c_int32_t, c_double
            integer(c_int32_t), value :: nprob
            real(c_double), intent(in) :: c(*)
            type(c_ptr) :: ctx
        end function
        subroutine btri_destroy(ctx) bind(C, name="btri_destroy")
            import :: c_ptr
            type(c_ptr), value :: ctx
        end subroutine
        function btri_launch(ctx, stream, npoints, dprob, n, dx, m, df) bind(C, name="btri_launch") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: ctx, stream, dprob, dx, df
            integer(c_int32_t), value :: npoints, n, m
            integer(c_int) :: rc
        end function
        function btri_launch_jac(ctx, stream, npoints, dprob, n, dx, m, dj) bind(C, name="btri_launch_jac") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: ctx, stream, dprob, dx, dj
            integer(c_int32_t), value :: npoints, n, m
            integer(c_int) :: rc
        end function
        function crosen_launch(ctx, stream, npoints, dprob, n, dx, m, df) bind(C, name="crosen_launch") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: ctx, stream, dprob, dx, df
            integer(c_int32_t), value :: npoints, n, m
            integer(c_int) :: rc
        end function
        function crosen_launch_grad(ctx, stream, npoints, dprob, n, dx, m, dg) bind(C, name="crosen_launch_grad") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: ctx, stream, dprob, dx, dg
            integer(c_int32_t), value :: npoints, n, m
            integer(c_int) :: rc
        end function
    end interface

    character(len=512) :: path
    integer(int32) :: nprob, m, n, nq, k, u
    real(real64), allocatable :: t(:,:), y(:,:), x0(:,:), x(:,:), f(:,:), x1(:), f1(:), c(:), xs(:,:), fs(:,:)
    type(c_ptr) :: ctx, ctx1, bctx
    type(vecfcn_helper) :: obj
    type(device_model_batch) :: batch
    type(least_squares_solver) :: lm
    type(newton_solver) :: nt
    type(quasi_newton_solver) :: qn
    type(bfgs) :: bf
    real(real64), allocatable :: fmin(:)
    type(iteration_behavior) :: ib
    type(iteration_behavior), allocatable :: ibs(:)
    integer(int32), allocatable :: st(:)

    if (command_argument_count() < 1) error stop 2
    call get_command_argument(1, path)
    open(newunit=u, file=trim(path), access="stream", form="unformatted", status="old")
    read(u) nprob, m, n
    allocate(t(m, nprob), y(m, nprob), x0(n, nprob))
    read(u) t
    read(u) y
    read(u) x0
    read(u) nq
    allocate(c(nprob), xs(nq, nprob))
    read(u) c
    read(u) xs
    close(u)

    call lm%set_max_fcn_evals(500)
    ! ---- one problem, the reference's own call: the user's data for problem 1 alone, its launcher, solver%solve
    ctx1 = lorentz_create(1, m, t(:,1), y(:,1))
    if (.not.c_associated(ctx1)) error stop 3
    call obj%set_device_fcn(c_funloc(lorentz_launch), ctx1, m, n)
    if (.not.obj%is_fcn_defined() .or. .not.obj%is_device_model_defined()) error stop 4
    if (obj%get_equation_count() /= m .or. obj%get_variable_count() /= n) error stop 5
    allocate(x1(n), f1(m))
    x1 = x0(:,1)
    call obj%fcn(x1, f1)                                       ! vecfcn of the user's device function: one evaluation on the GPU
    print '(A,*(1X,Z16.16))', "df_eval 0 0 0 0 F F F", f1(1), f1(m)
    call lm%solve(obj, x1, f1, ib)
    call report("df_lm_single", ib, 0, x1)
    print '(A,*(1X,Z16.16))', "df_lm_single_fvec 0 0 0 0 F F F", f1(1), f1(m)
    call obj%clear_device_model()
    call lorentz_destroy(ctx1)

    ! ---- every problem in one call
    ctx = lorentz_create(nprob, m, t, y)
    call batch%create_from_device_fcn(c_funloc(lorentz_launch), ctx, nprob, m, n)
    allocate(x(n, nprob), f(m, nprob), ibs(nprob), st(nprob))
    x = x0
    call lm%solve_batch(batch, x, f, ibs, st)
    do k = 1, nprob
        call report("df_lm_batch", ibs(k), st(k), x(:,k))
    end do
    call batch%destroy()

    ! ---- bounded least squares (constrained_least_squares_solver) on the same family: every problem inside one box
    block
        type(constrained_least_squares_solver) :: tr
        real(real64), allocatable :: lo(:), hi(:)
        integer(int32) :: q
        allocate(lo(n), hi(n))
        do q = 1, n, 3
            lo(q) = 0.4d0;  hi(q) = 1.2d0                      ! amplitudes
            lo(q + 1) = -1.0d0; hi(q + 1) = 2.0d0              ! centres
            lo(q + 2) = 0.02d0; hi(q + 2) = 0.2d0              ! widths
        end do
        call tr%set_max_fcn_evals(500)
        call tr%set_lower_limits(lo)
        call tr%set_upper_limits(hi)
        call batch%create_from_device_fcn(c_funloc(lorentz_launch), ctx, nprob, m, n)
        x = x0
        call tr%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("df_cls_batch", ibs(k), st(k), x(:,k))
        end do
        call batch%destroy()
        ! one problem through the reference's own call
        ctx1 = lorentz_create(1, m, t(:,2), y(:,2))
        call obj%set_device_fcn(c_funloc(lorentz_launch), ctx1, m, n)
        x1 = x0(:,2)
        call tr%solve(obj, x1, f1, ib)
        call report("df_cls_single", ib, 0, x1)
        call obj%clear_device_model()
        call lorentz_destroy(ctx1)
    end block
    call lorentz_destroy(ctx)

    ! ---- a square family with an analytic jacobianfcn launcher: newton_solver and quasi_newton_solver
    bctx = btri_create(nprob, c)
    call batch%create_from_device_fcn(c_funloc(btri_launch), bctx, nprob, nq, nq, c_funloc(btri_launch_jac))
    allocate(fs(nq, nprob))
    call nt%set_max_fcn_evals(500)
    call qn%set_max_fcn_evals(500)
    deallocate(x)
    allocate(x(nq, nprob))
    x = xs
    call nt%solve_batch(batch, x, fs, ibs, st)
    do k = 1, nprob
        call report("df_newton_batch", ibs(k), st(k), x(:,k))
    end do
    x = xs
    call qn%solve_batch(batch, x, fs, ibs, st)
    do k = 1, nprob
        call report("df_broyden_batch", ibs(k), st(k), x(:,k))
    end do
    call batch%destroy()
    ! forward differences instead of the analytic Jacobian (no jac launcher)
    call batch%create_from_device_fcn(c_funloc(btri_launch), bctx, nprob, nq, nq)
    x = xs
    call nt%solve_batch(batch, x, fs, ibs, st)
    do k = 1, nprob
        call report("df_newton_fd_batch", ibs(k), st(k), x(:,k))
    end do
    call batch%destroy()
    call btri_destroy(bctx)
    ! one square problem through quasi_newton_solver%solve itself (analytic jacobianfcn launcher)
    bctx = btri_create(1, c(3:3))
    call obj%set_device_fcn(c_funloc(btri_launch), bctx, nq, nq, c_funloc(btri_launch_jac))
    deallocate(x1, f1)
    allocate(x1(nq), f1(nq))
    x1 = xs(:,3)
    call qn%solve(obj, x1, f1, ib)
    call report("df_broyden_single", ib, 0, x1)
    call obj%clear_device_model()
    call btri_destroy(bctx)

    ! ---- a scalar objective (the reference's fcnnvar) as a model of ONE function: bfgs%solve_batch; the gradient launcher
    ! plays set_gradient_fcn, without it the forward differences are built on the device
    bctx = btri_create(nprob, c)
    allocate(fmin(nprob))
    call bf%set_max_fcn_evals(500)
    call batch%create_from_device_fcn(c_funloc(crosen_launch), bctx, nprob, 1, nq, c_funloc(crosen_launch_grad))
    x = xs
    call bf%solve_batch(batch, x, fmin, ibs, st)
    do k = 1, nprob
        call report_bfgs("df_bfgs_batch", ibs(k), st(k), fmin(k), x(:,k))
    end do
    call batch%destroy()
    call batch%create_from_device_fcn(c_funloc(crosen_launch), bctx, nprob, 1, nq)
    x = xs
    call bf%solve_batch(batch, x, fmin, ibs, st)
    do k = 1, nprob
        call report_bfgs("df_bfgs_fd_batch", ibs(k), st(k), fmin(k), x(:,k))
    end do
    call batch%destroy()
    call btri_destroy(bctx)

contains
    subroutine report_bfgs(name, b, st, fv, x)
        character(len=*), intent(in) :: name
        type(iteration_behavior), intent(in) :: b
        integer(int32), intent(in) :: st
        real(real64), intent(in) :: fv, x(:)
        print '(A,1X,I0,3(1X,I0),3(1X,L1),*(1X,Z16.16))', name, st, b%iter_count, b%fcn_count, b%gradient_count, &
            b%converge_on_fcn, b%converge_on_chng, b%converge_on_zero_diff, fv, x
    end subroutine

    subroutine report(name, b, st, x)
        character(len=*), intent(in) :: name
        type(iteration_behavior), intent(in) :: b
        integer(int32), intent(in) :: st
        real(real64), intent(in) :: x(:)
        print '(A,1X,I0,3(1X,I0),3(1X,L1),*(1X,Z16.16))', name, st, b%iter_count, b%fcn_count, b%jacobian_count, &
            b%converge_on_fcn, b%converge_on_chng, b%converge_on_zero_diff, x
    end subroutine
end program
