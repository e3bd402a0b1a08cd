! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle), CROSS-CHECK ONLY -- a STAND-IN, not reference code.
!
! The reference's phy/mod_cmnfld_routines.F90 imports from mod_dia (netCDF-bound, hence
! not buildable here) the number of diagnostic groups `nphy` and nine integer arrays of
! accumulation flags (line 34-36); it only tests whether any mixed-layer-depth diagnostic
! is requested (cmnfld1, :1110-1120).  This file supplies a module of that name with one
! diagnostic group and all flags zero -- no diagnostics requested -- so that the
! reference's REAL mod_cmnfld_routines compiles and its cmnfld1 / cmnfld2 arithmetic can
! be compared with oracle/c/cmnfld.c and the device (oracle/Makefile configs *_xed,
! tests/test_xcheck_cmnfld.py).
!
! Because it is a stand-in for a reference module, results obtained through it do NOT pin
! cmnfld's parity (DESIGN.md): they replace "same author, same reading" with the
! reference's own arithmetic, no more.  Nothing else is built against this file.
! ------------------------------------------------------------------------------
module mod_dia
  implicit none
  integer, parameter :: nphymax = 1
  integer :: nphy = 1
  integer, dimension(nphymax) :: ACC_BFSQ = 0, ACC_MLDL82 = 0, ACC_MLDL82MN = 0, ACC_MLDL82MX = 0, ACC_MLDL82SQ = 0, &
                                 ACC_MLDB04 = 0, ACC_MLDB04MN = 0, ACC_MLDB04MX = 0, ACC_MLDB04SQ = 0
end module mod_dia
