! TEST INFRASTRUCTURE -- stand-in, NOT the reference's module.
!
! phy/mod_tidaldissip.F90 reads the tidal wave energy dissipation from a netCDF file (`use netcdf`, absent in this image).
! phy/mod_difest.F90 imports the one array `twedon` from it; difest_vertical_iso reads it when tidally driven mixing acts
! (tdmflg = 1, :2897-2935).  This module holds that array, as the reference declares it (phy/mod_tidaldissip.F90:36-38), to be
! filled by the harness; cross-check builds only (oracle/Makefile: *_xdf), not pins.
module mod_tidaldissip
  use mod_types, only: r8
  use mod_xc
  implicit none
  real(r8), dimension(1-nbdy:idm+nbdy, 1-nbdy:jdm+nbdy) :: twedon = 0._r8
  public :: twedon
end module mod_tidaldissip
