! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle), CROSS-CHECK ONLY -- a STAND-IN, not reference code.
!
! The reference's channel/mod_thermf_channel.F90 imports two integers from ben02/mod_ben02.F90
! (`ntda`, the number of accumulated fields, which it increments, and `nrfets`, the e-folding
! time of the runoff reservoir in days, :108, :119 there); mod_ben02 reads its forcing with
! netCDF and is not buildable here.  This file supplies a module of that name holding just
! the two, so that the reference's REAL thermf_channel compiles for the cross-check builds
! *_xml (oracle/Makefile, tests/test_xcheck_thermf.py).  Because it is a stand-in for a
! reference module, results obtained through it do NOT pin thermf_channel's parity.
! ------------------------------------------------------------------------------
module mod_ben02
  implicit none
  integer :: ntda = 0
  integer :: nrfets = 7
end module mod_ben02
