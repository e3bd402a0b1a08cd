! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle), CROSS-CHECK ONLY -- a STAND-IN, not reference code.
!
! The reference's phy/mod_ale_forcing.F90 imports from mod_swabs (which reads chlorophyll
! climatologies with netCDF and is therefore not buildable here) the maximum depth of
! shortwave penetration and the four arrays of its two-band absorption profile
!    E(z) = E(0)*(swfc1*exp(-z/swal1) + swfc2*exp(-z/swal2))      (phy/mod_swabs.F90:27-31, :179-193).
! This file supplies a module of that name holding just those five variables, which the
! tests fill, so that the reference's REAL ale_forcing compiles for the cross-check builds
! *_xale / *_xaln (oracle/Makefile, tests/test_xcheck_ale_forcing.py).  Because it is a
! stand-in for a reference module, results obtained through it do NOT pin ale_forcing's
! parity (DESIGN.md).  Nothing else is built against this file.
! ------------------------------------------------------------------------------
module mod_swabs
  use dimensions, only: idm, jdm
  use mod_xc, only: nbdy
  implicit none
  real :: swamxd = 200.
  real, dimension(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy) :: swfc1 = 0., swfc2 = 0., swal1 = 1., swal2 = 1.
end module mod_swabs
