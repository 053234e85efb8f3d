program n2
use iso_fortran_env
implicit none
integer :: k, n, i
real(real64), allocatable :: x(:)
integer(int64) :: st
real(real64) :: r
st = 12345_int64
do k = 1, 300
  n = 1 + mod(k*37, 500)
  allocate(x(n))
  do i = 1, n
    st = mod(st * 6364136223846793005_int64 + 1442695040888963407_int64, 9223372036854775807_int64)
    x(i) = (real(mod(abs(st), 2000001_int64), real64) / 1000000.0d0 - 1.0d0) * 10.0d0**(mod(k,7)-3)
  end do
  r = norm2(x)
  write(*,'(I5,1X,Z16.16)') n, r
  write(*,'(*(Z16.16,1X))') x
  deallocate(x)
end do
end program
