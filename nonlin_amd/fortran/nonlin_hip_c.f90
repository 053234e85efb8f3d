! nonlin_hip_c.f90 -- ISO_C_BINDING view of include/nonlin_hip.h (libnonlin_hip.so).
! One interface body per C entry point the Fortran shim uses; struct layouts mirror the header.
module nonlin_hip_c
    use, intrinsic :: iso_c_binding
    implicit none
    public

    type, bind(C) :: nlh_iteration_behavior
        integer(c_int32_t) :: iter_count, fcn_count, jacobian_count, gradient_count
        integer(c_int32_t) :: converge_on_fcn, converge_on_chng, converge_on_zero_diff
    end type

    type, bind(C) :: nlh_options
        integer(c_int32_t) :: max_evals
        real(c_double) :: ftol, xtol, gtol
        integer(c_int32_t) :: print_status
        real(c_double) :: factor
        integer(c_int32_t) :: use_line_search, ls_max_evals
        real(c_double) :: ls_alpha, ls_factor
        integer(c_int32_t) :: factor_policy
        real(c_double) :: ne_pivot_tol
        integer(c_int32_t) :: fuse_fd
        integer(c_int32_t) :: sub_batches
    end type

    integer(c_int32_t), parameter :: NLH_FACTOR_AUTO = 0, NLH_FACTOR_QR = 1, NLH_FACTOR_EXACT = 2

    interface
        subroutine nlh_default_options(opts) bind(C, name="nlh_default_options")
            import :: nlh_options
            type(nlh_options), intent(out) :: opts
        end subroutine
        function nlh_create(h, device, stream) bind(C, name="nlh_create") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), intent(out) :: h
            integer(c_int32_t), value :: device
            type(c_ptr), value :: stream
            integer(c_int) :: rc
        end function
        subroutine nlh_destroy(h) bind(C, name="nlh_destroy")
            import :: c_ptr
            type(c_ptr), value :: h
        end subroutine
        function nlh_fd_jacobian(h, m, n, fcn, jacfcn, ctx, x, fv, jac) bind(C, name="nlh_fd_jacobian") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: m, n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            type(c_ptr), value :: fv
            real(c_double), intent(out) :: jac(*)
            integer(c_int) :: rc
        end function
        function nlh_lm_solve(h, opts, m, n, fcn, jacfcn, ctx, x, fvec, ib) bind(C, name="nlh_lm_solve") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: m, n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib
            integer(c_int) :: rc
        end function
        function nlh_newton_solve(h, opts, n, fcn, jacfcn, ctx, x, fvec, ib) bind(C, name="nlh_newton_solve") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib
            integer(c_int) :: rc
        end function
        function nlh_bfgs_solve(h, opts, n, fcn, gradfcn, ctx, x, fout, ib) bind(C, name="nlh_bfgs_solve") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: n
            type(c_funptr), value :: fcn, gradfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fout
            type(nlh_iteration_behavior), intent(out) :: ib
            integer(c_int) :: rc
        end function
        function nlh_fd_gradient(n, fcn, gradfcn, ctx, x, fv, g) bind(C, name="nlh_fd_gradient") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double
            integer(c_int32_t), value :: n
            type(c_funptr), value :: fcn, gradfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            type(c_ptr), value :: fv
            real(c_double), intent(out) :: g(*)
            integer(c_int) :: rc
        end function
        function nlh_poly_fit(h, npts, order, thru_zero, x, y, coef) bind(C, name="nlh_poly_fit") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: npts, order, thru_zero
            real(c_double), intent(in) :: x(*), y(*)
            real(c_double), intent(out) :: coef(*)
            integer(c_int) :: rc
        end function
        function nlh_cls_solve(h, opts, delta0, stepscale0, xl, xu, m, n, fcn, jacfcn, ctx, x, fvec, ib) &
                bind(C, name="nlh_cls_solve") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h
            type(nlh_options), intent(in) :: opts
            real(c_double), value :: delta0, stepscale0
            real(c_double), intent(in) :: xl(*), xu(*)
            integer(c_int32_t), value :: m, n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib
            integer(c_int) :: rc
        end function
        function nlh_quasi_newton_solve(h, opts, jdelta, n, fcn, jacfcn, ctx, x, fvec, ib) &
                bind(C, name="nlh_quasi_newton_solve") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: jdelta, n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib
            integer(c_int) :: rc
        end function
        ! ---- device residual models behind host arrays (include/nonlin_hip.h: nlh_dq_model_*) ----
        function nlh_dq_model_create(h, nprob, m, n, a, b, gamma, model) bind(C, name="nlh_dq_model_create") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: nprob, m, n
            real(c_double), intent(in) :: a(*), b(*)
            real(c_double), value :: gamma
            type(c_ptr), intent(out) :: model
            integer(c_int) :: rc
        end function
        ! a USER'S device residual (launchers, include/nonlin_hip.h: nlh_device_vecfcn / nlh_device_jacfcn) as a model object
        function nlh_device_fcn_model_create(nprob, m, n, fcn, jacfcn, ctx, model) &
                bind(C, name="nlh_device_fcn_model_create") result(rc)
            import :: c_ptr, c_funptr, c_int, c_int32_t
            integer(c_int32_t), value :: nprob, m, n
            type(c_funptr), value :: fcn, jacfcn
            type(c_ptr), value :: ctx
            type(c_ptr), intent(out) :: model
            integer(c_int) :: rc
        end function
        ! ---- several GPUs behind the boundary (include/nonlin_hip.h: nlh_device_set_*) ----
        function nlh_device_set_create(set, devices, ndev) bind(C, name="nlh_device_set_create") result(rc)
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), intent(out) :: set
            integer(c_int32_t), intent(in) :: devices(*)
            integer(c_int32_t), value :: ndev
            integer(c_int) :: rc
        end function
        subroutine nlh_device_set_destroy(set) bind(C, name="nlh_device_set_destroy")
            import :: c_ptr
            type(c_ptr), value :: set
        end subroutine
        function nlh_device_set_size(set) bind(C, name="nlh_device_set_size") result(n)
            import :: c_ptr, c_int32_t
            type(c_ptr), value :: set
            integer(c_int32_t) :: n
        end function
        function nlh_device_count() bind(C, name="nlh_device_count") result(n)
            import :: c_int
            integer(c_int) :: n
        end function
        function nlh_dq_model_create_on(set, nprob, m, n, a, b, gamma, model) bind(C, name="nlh_dq_model_create_on") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: set
            integer(c_int32_t), value :: nprob, m, n
            real(c_double), intent(in) :: a(*), b(*)
            real(c_double), value :: gamma
            type(c_ptr), intent(out) :: model
            integer(c_int) :: rc
        end function
        subroutine nlh_dq_model_destroy(model) bind(C, name="nlh_dq_model_destroy")
            import :: c_ptr
            type(c_ptr), value :: model
        end subroutine
        function nlh_dq_model_eval(h, model, x, f) bind(C, name="nlh_dq_model_eval") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h, model
            real(c_double), intent(in) :: x(*)
            real(c_double), intent(out) :: f(*)
            integer(c_int) :: rc
        end function
        function nlh_dq_model_lm_solve(h, opts, model, x, fvec, ib, status) bind(C, name="nlh_dq_model_lm_solve") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h, model
            type(nlh_options), intent(in) :: opts
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib(*)
            integer(c_int32_t), intent(out) :: status(*)
            integer(c_int) :: rc
        end function
        function nlh_dq_model_newton_solve(h, opts, model, analytic, x, fvec, ib, status) &
                bind(C, name="nlh_dq_model_newton_solve") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h, model
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: analytic
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib(*)
            integer(c_int32_t), intent(out) :: status(*)
            integer(c_int) :: rc
        end function
        function nlh_dq_model_quasi_newton_solve(h, opts, model, jdelta, analytic, x, fvec, ib, status) &
                bind(C, name="nlh_dq_model_quasi_newton_solve") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h, model
            type(nlh_options), intent(in) :: opts
            integer(c_int32_t), value :: jdelta, analytic
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib(*)
            integer(c_int32_t), intent(out) :: status(*)
            integer(c_int) :: rc
        end function
        function nlh_dq_model_cls_solve(h, opts, model, delta0, stepscale0, xl, xu, x, fvec, ib, status) &
                bind(C, name="nlh_dq_model_cls_solve") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h, model
            type(nlh_options), intent(in) :: opts
            real(c_double), value :: delta0, stepscale0
            real(c_double), intent(in) :: xl(*), xu(*)
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*)
            type(nlh_iteration_behavior), intent(out) :: ib(*)
            integer(c_int32_t), intent(out) :: status(*)
            integer(c_int) :: rc
        end function
        function nlh_dq_model_bfgs_solve(h, opts, model, x, fvec, fout, ib, status) &
                bind(C, name="nlh_dq_model_bfgs_solve") result(rc)
            import :: c_ptr, c_int, c_int32_t, c_double, nlh_options, nlh_iteration_behavior
            type(c_ptr), value :: h, model
            type(nlh_options), intent(in) :: opts
            real(c_double), intent(inout) :: x(*)
            real(c_double), intent(out) :: fvec(*), fout(*)
            type(nlh_iteration_behavior), intent(out) :: ib(*)
            integer(c_int32_t), intent(out) :: status(*)
            integer(c_int) :: rc
        end function
    end interface

    type(c_ptr), save, private :: default_handle = c_null_ptr
    type(c_ptr), save, private :: default_set = c_null_ptr
    logical, save, private :: default_set_decided = .false.

contains
    !> Extension: the GPUs that device_model_batch%create deals a batch over (solve_batch then runs one host thread
    !> per GPU inside this process).  devices = device ids, 0-based; an empty list selects every visible GPU.  Without
    !> this call the environment variable NLH_DEVICES decides ("all", or a comma-separated id list); unset: one GPU
    !> (device 0, the default handle).  Models created earlier keep the devices they were created on: the previous set is
    !> released here, and the library keeps its handles alive until the last model dealt over it has been destroyed.
    subroutine nlh_use_devices(devices)
        integer(c_int32_t), intent(in), dimension(:) :: devices
        integer(c_int) :: rc
        integer(c_int32_t) :: none(1)
        none = 0
        if (c_associated(default_set)) then
            call nlh_device_set_destroy(default_set)
            default_set = c_null_ptr
        end if
        if (size(devices) > 0) then
            rc = nlh_device_set_create(default_set, devices, int(size(devices), c_int32_t))
        else
            rc = nlh_device_set_create(default_set, none, 0_c_int32_t)
        end if
        if (rc /= 0) then
            print '(A,I0,A)', "nonlin_hip: nlh_device_set_create returned ", rc, " (no such HIP device?)"
            error stop 1
        end if
        default_set_decided = .true.
    end subroutine

    !> The device set models are dealt over, or c_null_ptr when a single GPU (the default handle) is in use.
    function nlh_default_device_set() result(set)
        type(c_ptr) :: set
        character(len=256) :: spec
        integer :: length, stat, i, k, cnt
        integer(c_int32_t) :: ids(64)
        if (.not.default_set_decided) then
            default_set_decided = .true.
            call get_environment_variable("NLH_DEVICES", spec, length, stat)
            if (stat == 0 .and. length > 0) then
                cnt = 0
                if (spec(1:length) /= "all") then
                    k = 1
                    do i = 1, length + 1
                        if (i > length .or. spec(i:min(i, length)) == ",") then
                            if (i > k) then
                                if (cnt >= size(ids)) then
                                    print '(A,I0,A)', "nonlin_hip: NLH_DEVICES lists more than ", size(ids), " devices"
                                    error stop 1
                                end if
                                cnt = cnt + 1
                                read (spec(k:i - 1), *, iostat = stat) ids(cnt)
                                if (stat /= 0 .or. verify(trim(adjustl(spec(k:i - 1))), "0123456789") /= 0) then
                                    print '(3A)', "nonlin_hip: NLH_DEVICES must be 'all' or a comma-separated list of device ids, got '", &
                                        spec(1:length), "'"
                                    error stop 1
                                end if
                            else if (i <= length .or. k > 1) then      ! an empty entry ("0,,1", a trailing comma)
                                print '(3A)', "nonlin_hip: NLH_DEVICES has an empty entry: '", spec(1:length), "'"
                                error stop 1
                            end if
                            k = i + 1
                        end if
                    end do
                end if
                call nlh_use_devices(ids(1:cnt))
            end if
        end if
        set = default_set
    end function

    !> Lazily created process-wide handle on device 0, default stream.
    function nlh_default_handle() result(h)
        type(c_ptr) :: h
        integer(c_int) :: rc
        if (.not.c_associated(default_handle)) then
            rc = nlh_create(default_handle, 0_c_int32_t, c_null_ptr)
            if (rc /= 0) then
                print '(A,I0)', "nonlin_hip: no usable HIP device (nlh_create returned ", rc
                error stop 1
            end if
        end if
        h = default_handle
    end function
end module
