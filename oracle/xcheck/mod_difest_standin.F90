! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle), CROSS-CHECK ONLY -- a STAND-IN, not reference code.
!
! The reference's phy/mod_eddtra.F90 imports one array from mod_difest (line 41:
! `use mod_difest, only: OBLdepth`), and mod_difest needs the CVMix library, which is not
! in this image.  OBLdepth is read only on the eddtra_ale (hybrid coordinate) branch
! (:1063, :1087); the isopyc_bulkml branches this project builds (:152-1000) never touch
! it.  This file supplies a module of that name holding only that array so that the
! reference's REAL mod_eddtra can be compiled and its bulkml arithmetic compared with
! oracle/c/eddtra.c and the device (oracle/Makefile configs *_xed, tests/test_xcheck_eddtra.py).
!
! Because it is a stand-in for a reference module, results obtained through it do NOT pin
! eddtra's parity (DESIGN.md): they replace "same author, same reading" with the
! reference's own arithmetic, no more.  Nothing else is built against this file.
! ------------------------------------------------------------------------------
module mod_difest
  use mod_xc, only: idm, jdm, nbdy
  implicit none
  real(8), dimension(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy) :: OBLdepth = 0.0d0
end module mod_difest
