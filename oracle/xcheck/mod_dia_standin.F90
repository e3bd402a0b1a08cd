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
!
! Round 3: also what phy/mod_ale_regrid_remap.F90 imports (:53-57) -- the number of diagnostic z-levels `ddm` (it sizes a
! remapping structure that is never used when no z-level diagnostic is requested), the alarm and accumulation flags of the
! z-level diagnostics (all zero), and the arrays those diagnostics would read or fill (present, never touched) -- so that the
! reference's REAL ale_regrid_remap compiles for the cross-check builds *_xale (tests/test_xcheck_ale.py).  Same caveat.
module mod_dia
  use dimensions, only: idm, jdm
  use mod_xc, only: nbdy
  implicit none
  integer, parameter :: nphymax = 1
  integer :: nphy = 1
  integer, dimension(nphymax) :: ACC_BFSQ = 0, ACC_MLDL82 = 0, ACC_MLDL82MN = 0, ACC_MLDL82MX = 0, ACC_MLDL82SQ = 0, &
                                 ACC_MLDB04 = 0, ACC_MLDB04MN = 0, ACC_MLDB04MX = 0, ACC_MLDB04SQ = 0
  integer, parameter :: ddm = 35
  integer, dimension(nphymax) :: alarm_phy = 0, acc_templvl = 0, acc_salnlvl = 0, acc_uvellvl = 0, acc_vvellvl = 0, &
                                 acc_idlagelvl = 0
  real, dimension(2,ddm) :: depthslev_bnds = 0.
  real, dimension(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy) :: pbath = 1., ubath = 1., vbath = 1.
  real, allocatable, dimension(:,:,:,:) :: phylvl
end module mod_dia
