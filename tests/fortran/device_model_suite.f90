! User-side program for the device-model extension of the drop-in layer (vecfcn_helper%set_device_model,
! device_model_batch, least_squares_solver%solve_batch / newton_solver%solve_batch): reads problem data written by
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
        call batch%destroy()
    end subroutine
end program
