! Golden vectors for the Fortran edit descriptor E10.3 / I0 as amdflang's runtime prints them
! (the formats of print_status, src/nonlin_helper.f90:29-32).  Build-owned program; not reference code.
program e103
    use iso_fortran_env
    implicit none
    real(real64) :: v(18)
    integer :: i
    v = [0.0d0, 1.23d0, -2.5d0, 1.23456d-4, 0.9996d0, 0.99949d0, 9.995d0, 1.0d100, 1.0d-100, 12345.678d0, &
         4.44089209850063d-16, 1.28602518018146d0, 0.506363030790737d0, 1.0d0, 0.1d0, 99.95d0, -1.0d-7, 5.0d-324]
    do i = 1, size(v)
        print 101, "Change in Variable: ", v(i)
    end do
    print *, ""
    print 100, "Iteration: ", 12
101 format(A, E10.3)
100 format(A, I0)
end program
