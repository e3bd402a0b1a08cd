! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle), CROSS-CHECK ONLY -- a STAND-IN, not reference code.
!
! The reference's phy/mod_mxlayr.F90 has a bare `use mod_nctools` (line 54) and calls none of
! its routines -- but mod_nctools has `use mod_xc` and no `private`, so that line is also where
! mod_mxlayr gets ii, jj, kk, mnproc, lp from.  mod_nctools itself wraps the netCDF library,
! which is not in this image.  A module of that name that only passes mod_xc on lets the
! reference's REAL mod_mxlayr compile for the cross-check builds
! *_xml (oracle/Makefile, tests/test_xcheck_mxlayr.py).  Because it is a stand-in for a
! reference module, results obtained through it do NOT pin mxlayr's parity (DESIGN.md).
! ------------------------------------------------------------------------------
module mod_nctools
  use mod_xc
  implicit none
end module mod_nctools
