! User-side program for the device-model extension of the drop-in layer (vecfcn_helper%set_device_model,
! device_model_batch, solve_batch of the least-squares, Newton, quasi-Newton, bounded least-squares and BFGS solvers):
! reads problem data written by
! tests/test_gpu_fortran.py (stream binary: nprob, m, n (int32), gamma (real64), A(m,n,nprob), b(m,nprob), x0(n,nprob)),
! solves on the GPU through `use nonlin`, prints counts, flags and the bit patterns of x.
program device_model_suite
    use iso_fortran_env
    use nonlin
    implicit none
    character(len=512) :: path
    integer :: nargs

    nargs = command_argument_count()
    if (nargs < 2) error stop 2
    call get_command_argument(1, path)
    call run_lm(trim(path))
    call get_command_argument(2, path)
    call run_newton(trim(path))
    call get_command_argument(1, path)
    call run_box_and_bfgs(trim(path))
    if (nargs >= 3) then                                      ! zero-residual problems under the MFMA / Cholesky policy (N1)
        call get_command_argument(3, path)
        call run_lm_auto_policy(trim(path))
    end if

contains
    subroutine load(path, nprob, m, n, gamma, a, b, x0)
        character(len=*), intent(in) :: path
        integer(int32), intent(out) :: nprob, m, n
        real(real64), intent(out) :: gamma
        real(real64), allocatable, intent(out) :: a(:,:,:), b(:,:), x0(:,:)
        integer :: u
        open(newunit=u, file=path, access="stream", form="unformatted", status="old")
        read(u) nprob, m, n, gamma
        allocate(a(m, n, nprob), b(m, nprob), x0(n, nprob))
        read(u) a
        read(u) b
        read(u) x0
        close(u)
    end subroutine

    subroutine report(name, b, st, x)
        character(len=*), intent(in) :: name
        type(iteration_behavior), intent(in) :: b
        integer(int32), intent(in) :: st
        real(real64), intent(in) :: x(:)
        print '(A,1X,I0,3(1X,I0),3(1X,L1),*(1X,Z16.16))', name, st, b%iter_count, b%fcn_count, b%jacobian_count, &
            b%converge_on_fcn, b%converge_on_chng, b%converge_on_zero_diff, x
    end subroutine

    subroutine run_lm(path)
        character(len=*), intent(in) :: path
        integer(int32) :: nprob, m, n, k
        real(real64) :: gamma
        real(real64), allocatable :: a(:,:,:), b(:,:), x0(:,:), x(:,:), f(:,:), x1(:), f1(:)
        type(vecfcn_helper) :: obj
        type(device_model_batch) :: batch
        type(least_squares_solver) :: lm
        type(iteration_behavior) :: ib
        type(iteration_behavior), allocatable :: ibs(:)
        integer(int32), allocatable :: st(:)

        call load(path, nprob, m, n, gamma, a, b, x0)
        call lm%set_max_fcn_evals(500)
        ! one problem through the reference's own call: solver%solve(obj, x, fvec, ib)
        call obj%set_device_model(NLH_MODEL_DENSE_QUADRATIC, a(:,:,1), b(:,1), gamma)
        if (.not.obj%is_fcn_defined() .or. .not.obj%is_device_model_defined()) error stop 3
        if (obj%get_equation_count() /= m .or. obj%get_variable_count() /= n) error stop 4
        allocate(x1(n), f1(m))
        x1 = x0(:,1)
        call obj%fcn(x1, f1)                                   ! vecfcn of the model (one evaluation on the GPU)
        print '(A,*(1X,Z16.16))', "dm_eval 0 0 0 0 F F F", f1(1), f1(m)
        call lm%solve(obj, x1, f1, ib)
        call report("dm_lm_single", ib, 0, x1)
        print '(A,*(1X,Z16.16))', "dm_lm_single_fvec 0 0 0 0 F F F", f1(1), f1(m)
        call obj%clear_device_model()
        ! all problems in one call
        call batch%create(NLH_MODEL_DENSE_QUADRATIC, a, b, gamma)
        allocate(x(n, nprob), f(m, nprob), ibs(nprob), st(nprob))
        x = x0
        call lm%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_lm_batch", ibs(k), st(k), x(:,k))
        end do
        call batch%destroy()
        ! the same batch dealt over a device set inside this process (here: two shares on GPU 0; on a multi-GPU node
        ! nlh_use_devices([0, 1, ...]) or NLH_DEVICES=all): every problem must come back with the same bits
        call nlh_use_devices([0, 0])
        call batch%create(NLH_MODEL_DENSE_QUADRATIC, a, b, gamma)
        x = x0
        call lm%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_lm_batch_set", ibs(k), st(k), x(:,k))
        end do
        call batch%destroy()
    end subroutine

    ! least_squares_solver%solve_batch with the extension component factor_policy = NLH_FACTOR_AUTO (J^T J on the fp64 MFMA +
    ! Cholesky step solve: the formulation the reference's doc-comment names, src/nonlin_least_squares.f90:21-24) and with
    ! the default NLH_FACTOR_EXACT, on zero-residual problems
    subroutine run_lm_auto_policy(path)
        character(len=*), intent(in) :: path
        integer(int32) :: nprob, m, n, k
        real(real64) :: gamma
        real(real64), allocatable :: a(:,:,:), b(:,:), x0(:,:), x(:,:), f(:,:)
        type(device_model_batch) :: batch
        type(least_squares_solver) :: lm
        type(iteration_behavior), allocatable :: ibs(:)
        integer(int32), allocatable :: st(:)

        call load(path, nprob, m, n, gamma, a, b, x0)
        call lm%set_max_fcn_evals(500)
        call batch%create(NLH_MODEL_DENSE_QUADRATIC, a, b, gamma)
        allocate(x(n, nprob), f(m, nprob), ibs(nprob), st(nprob))
        lm%factor_policy = NLH_FACTOR_AUTO
        x = x0
        call lm%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_lm_auto_zero_residual", ibs(k), st(k), x(:,k))
        end do
        lm%factor_policy = NLH_FACTOR_EXACT
        x = x0
        call lm%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_lm_exact_zero_residual", ibs(k), st(k), x(:,k))
        end do
        call batch%destroy()
    end subroutine

    subroutine run_newton(path)
        character(len=*), intent(in) :: path
        integer(int32) :: nprob, m, n, k
        real(real64) :: gamma
        real(real64), allocatable :: a(:,:,:), b(:,:), x0(:,:), x(:,:), f(:,:), x1(:), f1(:)
        type(vecfcn_helper) :: obj
        type(device_model_batch) :: batch
        type(newton_solver) :: nt
        type(iteration_behavior) :: ib
        type(iteration_behavior), allocatable :: ibs(:)
        integer(int32), allocatable :: st(:)

        call load(path, nprob, m, n, gamma, a, b, x0)
        call nt%set_max_fcn_evals(500)
        allocate(x1(n), f1(n))
        call obj%set_device_model(NLH_MODEL_DENSE_QUADRATIC, a(:,:,1), b(:,1), gamma, analytic = .true.)
        x1 = x0(:,1)
        call nt%solve(obj, x1, f1, ib)
        call report("dm_newton_an", ib, 0, x1)
        call obj%set_device_model(NLH_MODEL_DENSE_QUADRATIC, a(:,:,1), b(:,1), gamma)      ! forward differences
        x1 = x0(:,1)
        call nt%solve(obj, x1, f1, ib)
        call report("dm_newton_fd", ib, 0, x1)
        call batch%create(NLH_MODEL_DENSE_QUADRATIC, a, b, gamma, analytic = .true.)
        allocate(x(n, nprob), f(n, nprob), ibs(nprob), st(nprob))
        x = x0
        call nt%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_newton_batch", ibs(k), st(k), x(:,k))
        end do
        call run_broyden_batch(batch, x0)
        call batch%destroy()
    end subroutine

    ! quasi_newton_solver%solve_batch on the square problems of run_newton
    subroutine run_broyden_batch(batch, x0)
        type(device_model_batch), intent(in) :: batch
        real(real64), intent(in) :: x0(:,:)
        type(quasi_newton_solver) :: qn
        real(real64), allocatable :: x(:,:), f(:,:)
        type(iteration_behavior), allocatable :: ibs(:)
        integer(int32), allocatable :: st(:)
        integer(int32) :: k
        call qn%set_max_fcn_evals(500)
        allocate(x, source = x0)
        allocate(f(size(x0, 1), size(x0, 2)), ibs(size(x0, 2)), st(size(x0, 2)))
        call qn%solve_batch(batch, x, f, ibs, st)
        do k = 1, size(x0, 2)
            call report("dm_broyden_batch", ibs(k), st(k), x(:,k))
        end do
    end subroutine

    ! constrained_least_squares_solver%solve_batch (box -0.3 .. 0.25) and bfgs%solve_batch on the problems of run_lm
    subroutine run_box_and_bfgs(path)
        character(len=*), intent(in) :: path
        integer(int32) :: nprob, m, n, k
        real(real64) :: gamma
        real(real64), allocatable :: a(:,:,:), b(:,:), x0(:,:), x(:,:), f(:,:), fo(:), lo(:), hi(:)
        type(device_model_batch) :: batch
        type(constrained_least_squares_solver) :: tr
        type(bfgs) :: qb
        type(iteration_behavior), allocatable :: ibs(:)
        integer(int32), allocatable :: st(:)

        call load(path, nprob, m, n, gamma, a, b, x0)
        call batch%create(NLH_MODEL_DENSE_QUADRATIC, a, b, gamma)
        allocate(x(n, nprob), f(m, nprob), fo(nprob), ibs(nprob), st(nprob), lo(n), hi(n))
        lo = -0.3d0
        hi = 0.25d0
        call tr%set_lower_limits(lo)
        call tr%set_upper_limits(hi)
        call tr%set_max_fcn_evals(500)
        x = x0
        call tr%solve_batch(batch, x, f, ibs, st)
        do k = 1, nprob
            call report("dm_cls_batch", ibs(k), st(k), x(:,k))
        end do
        call qb%set_max_fcn_evals(300)
        call qb%set_tolerance(1.0d-8)
        call qb%set_var_tolerance(1.0d-12)
        x = x0
        call qb%solve_batch(batch, x, fo, ibs, st)
        do k = 1, nprob
            ibs(k)%jacobian_count = ibs(k)%gradient_count      ! (report prints the third counter)
            call report("dm_bfgs_batch", ibs(k), st(k), [x(:,k), fo(k)])
        end do
        call batch%destroy()
    end subroutine
end program
