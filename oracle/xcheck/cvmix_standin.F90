! TEST INFRASTRUCTURE -- stand-in, NOT the CVMix library.
!
! phy/mod_difest.F90 imports, at module level, the procedures and types of CVMix-src (v0.98-beta, an un-vendored
! submodule of the reference: .gitmodules:1-6) that its HYBRID-coordinate routines call (init_difest :278-339,
! difest_vertical_hyb :1092-1364).  The routines of the isopycnic coordinate -- difest_isobml, difest_common_iso,
! difest_vertical_iso, difest_lateral_iso (:353-586, :735-809, :2040-3084) -- call none of them.  So that the REAL
! mod_difest.F90 can be compiled and its isopycnic routines run next to this project's kernels, the modules below
! declare the imported names with the argument names the call sites use and nothing behind them: every body is
! `error stop`.  Builds that contain this file (oracle/Makefile: *_xdf) are CROSS-CHECKS, NOT PINS: they say that the
! device reproduces the arithmetic of the reference's own difest_*_iso source as compiled here, not that a CVMix-linked
! BLOM would print the same bits (it would: the routines do not touch CVMix -- but that is an argument, not a test).
module CVMix_kinds_and_types
  implicit none
  type :: CVMix_global_params_type
    integer :: max_nlev = 0
    real(8) :: Prandtl = 0._8, FreshWaterDensity = 0._8, SaltWaterDensity = 0._8, Gravity = 0._8
  end type CVMix_global_params_type
end module CVMix_kinds_and_types

module CVMix_put_get
  use CVMix_kinds_and_types, only: CVMix_global_params_type
  implicit none
  interface CVMix_put
    module procedure put_int, put_real
  end interface CVMix_put
contains
  subroutine put_int(p, name, val)
    type(CVMix_global_params_type), intent(inout) :: p
    character(len=*), intent(in) :: name
    integer, intent(in) :: val
    error stop 'cvmix_standin: CVMix_put called'
  end subroutine put_int
  subroutine put_real(p, name, val)
    type(CVMix_global_params_type), intent(inout) :: p
    character(len=*), intent(in) :: name
    real(8), intent(in) :: val
    error stop 'cvmix_standin: CVMix_put called'
  end subroutine put_real
end module CVMix_put_get

module CVMix_kpp
  use CVMix_kinds_and_types, only: CVMix_global_params_type
  implicit none
  type :: CVMix_kpp_params_type
    integer :: unused = 0
  end type CVMix_kpp_params_type
contains
  subroutine CVMix_init_kpp(Ri_crit, minOBLdepth, minVtsqr, vonKarman, surf_layer_ext, interp_type, interp_type2, lEkman, &
                            lMonOb, MatchTechnique, lenhanced_diff, lnonzero_surf_nonlocal, lnoDGat1, Langmuir_mixing_str, &
                            Langmuir_entrainment_str, CVMix_kpp_params_user)
    real(8), optional :: Ri_crit, minOBLdepth, minVtsqr, vonKarman, surf_layer_ext
    character(len=*), optional :: interp_type, interp_type2, MatchTechnique, Langmuir_mixing_str, Langmuir_entrainment_str
    logical, optional :: lEkman, lMonOb, lenhanced_diff, lnonzero_surf_nonlocal, lnoDGat1
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    error stop 'cvmix_standin: CVMix_init_kpp called'
  end subroutine CVMix_init_kpp
  subroutine CVMix_put_kpp(name, val)
    character(len=*) :: name
    real(8) :: val
    error stop 'cvmix_standin: CVMix_put_kpp called'
  end subroutine CVMix_put_kpp
  subroutine CVMix_kpp_compute_turbulent_scales(sigma_coord, OBL_depth, surf_buoy_force, surf_fric_vel, w_m, w_s, &
                                                CVMix_kpp_params_user)
    real(8) :: sigma_coord, surf_fric_vel
    real(8), dimension(:) :: OBL_depth, surf_buoy_force
    real(8), dimension(:), optional :: w_m, w_s
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    error stop 'cvmix_standin: CVMix_kpp_compute_turbulent_scales called'
  end subroutine CVMix_kpp_compute_turbulent_scales
  function cvmix_kpp_EFactor_model(u10, ustar, hbl, CVmix_params_in) result(r)
    real(8) :: u10, ustar, hbl, r
    type(CVMix_global_params_type) :: CVmix_params_in
    r = 0._8
    error stop 'cvmix_standin: cvmix_kpp_EFactor_model called'
  end function cvmix_kpp_EFactor_model
  function CVmix_kpp_compute_unresolved_shear(zt_cntr, ws_cntr, N_iface, Nsqr_iface, EFactor, LaSL, bfsfc, ustar, &
                                              CVMix_kpp_params_user) result(r)
    real(8), dimension(:) :: zt_cntr, ws_cntr
    real(8), dimension(:), optional :: N_iface, Nsqr_iface
    real(8), optional :: EFactor, LaSL, bfsfc, ustar
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    real(8), dimension(size(zt_cntr)) :: r
    r = 0._8
    error stop 'cvmix_standin: CVmix_kpp_compute_unresolved_shear called'
  end function CVmix_kpp_compute_unresolved_shear
  function CVmix_kpp_compute_bulk_Richardson(zt_cntr, delta_buoy_cntr, delta_Vsqr_cntr, Vt_sqr_cntr, ws_cntr, N_iface, &
                                             Nsqr_iface, EFactor, LaSL, bfsfc, ustar, CVMix_kpp_params_user) result(r)
    real(8), dimension(:) :: zt_cntr, delta_buoy_cntr, delta_Vsqr_cntr
    real(8), dimension(:), optional :: Vt_sqr_cntr, ws_cntr, N_iface, Nsqr_iface
    real(8), optional :: EFactor, LaSL, bfsfc, ustar
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    real(8), dimension(size(zt_cntr)) :: r
    r = 0._8
    error stop 'cvmix_standin: CVmix_kpp_compute_bulk_Richardson called'
  end function CVmix_kpp_compute_bulk_Richardson
  subroutine CVMix_kpp_compute_OBL_depth(Ri_bulk, zw_iface, OBL_depth, kOBL_depth, zt_cntr, surf_fric, surf_buoy, Coriolis, &
                                         CVMix_kpp_params_user)
    real(8), dimension(:) :: Ri_bulk, zw_iface
    real(8) :: OBL_depth, kOBL_depth
    real(8), dimension(:), optional :: zt_cntr
    real(8), optional :: surf_fric, surf_buoy, Coriolis
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    error stop 'cvmix_standin: CVMix_kpp_compute_OBL_depth called'
  end subroutine CVMix_kpp_compute_OBL_depth
  function CVMix_kpp_compute_kOBL_depth(zw_iface, zt_cntr, OBL_depth) result(r)
    real(8), dimension(:) :: zw_iface, zt_cntr
    real(8) :: OBL_depth, r
    r = 0._8
    error stop 'cvmix_standin: CVMix_kpp_compute_kOBL_depth called'
  end function CVMix_kpp_compute_kOBL_depth
  subroutine CVMix_coeffs_kpp(Mdiff_out, Tdiff_out, Sdiff_out, zw, zt, old_Mdiff, old_Tdiff, old_Sdiff, OBL_depth, kOBL_depth, &
                              Tnonlocal, Snonlocal, surf_fric, surf_buoy, nlev, max_nlev, Langmuir_EFactor, &
                              CVMix_kpp_params_user)
    real(8), dimension(:) :: Mdiff_out, Tdiff_out, Sdiff_out, zw, zt, old_Mdiff, old_Tdiff, old_Sdiff, Tnonlocal, Snonlocal
    real(8) :: OBL_depth, kOBL_depth, surf_fric, surf_buoy
    integer :: nlev, max_nlev
    real(8), optional :: Langmuir_EFactor
    type(CVMix_kpp_params_type), optional :: CVMix_kpp_params_user
    error stop 'cvmix_standin: CVMix_coeffs_kpp called'
  end subroutine CVMix_coeffs_kpp
end module CVMix_kpp

module CVMix_shear
  implicit none
contains
  subroutine CVMix_init_shear(mix_scheme, KPP_nu_zero, KPP_Ri_zero, KPP_exp)
    character(len=*), optional :: mix_scheme
    real(8), optional :: KPP_nu_zero, KPP_Ri_zero, KPP_exp
    error stop 'cvmix_standin: CVMix_init_shear called'
  end subroutine CVMix_init_shear
  subroutine CVMix_coeffs_shear(Mdiff_out, Tdiff_out, RICH, nlev, max_nlev)
    real(8), dimension(:) :: Mdiff_out, Tdiff_out, RICH
    integer :: nlev, max_nlev
    error stop 'cvmix_standin: CVMix_coeffs_shear called'
  end subroutine CVMix_coeffs_shear
end module CVMix_shear

module CVMix_background
  implicit none
contains
  subroutine CVMix_init_bkgnd(max_nlev, zw, bl1, bl2, bl3, bl4, prandtl)
    integer :: max_nlev
    real(8), dimension(:) :: zw
    real(8) :: bl1, bl2, bl3, bl4, prandtl
    error stop 'cvmix_standin: CVMix_init_bkgnd called'
  end subroutine CVMix_init_bkgnd
  subroutine CVMix_coeffs_bkgnd(Mdiff_out, Tdiff_out, nlev, max_nlev)
    real(8), dimension(:) :: Mdiff_out, Tdiff_out
    integer :: nlev, max_nlev
    error stop 'cvmix_standin: CVMix_coeffs_bkgnd called'
  end subroutine CVMix_coeffs_bkgnd
end module CVMix_background

module CVMix_convection
  implicit none
contains
  subroutine CVMix_init_conv(convect_diff, convect_visc, lBruntVaisala, BVsqr_convect)
    real(8) :: convect_diff, convect_visc, BVsqr_convect
    logical :: lBruntVaisala
    error stop 'cvmix_standin: CVMix_init_conv called'
  end subroutine CVMix_init_conv
  subroutine CVMix_coeffs_conv(Mdiff_out, Tdiff_out, Nsqr, dens, dens_lwr, nlev, max_nlev, OBL_ind)
    real(8), dimension(:) :: Mdiff_out, Tdiff_out, Nsqr, dens, dens_lwr
    integer :: nlev, max_nlev, OBL_ind
    error stop 'cvmix_standin: CVMix_coeffs_conv called'
  end subroutine CVMix_coeffs_conv
end module CVMix_convection

module CVMix_tidal
  use CVMix_kinds_and_types, only: CVMix_global_params_type
  implicit none
  type :: CVMix_tidal_params_type
    integer :: unused = 0
  end type CVMix_tidal_params_type
contains
  subroutine CVMix_init_tidal(CVmix_tidal_params_user, mix_scheme, efficiency, local_mixing_frac)
    type(CVMix_tidal_params_type) :: CVmix_tidal_params_user
    character(len=*) :: mix_scheme
    real(8) :: efficiency, local_mixing_frac
    error stop 'cvmix_standin: CVMix_init_tidal called'
  end subroutine CVMix_init_tidal
  subroutine CVMix_compute_Simmons_invariant(nlev, energy_flux, rho, SimmonsCoeff, VertDep, zw, zt, CVmix_tidal_params_user)
    integer :: nlev
    real(8) :: energy_flux, rho, SimmonsCoeff
    real(8), dimension(:) :: VertDep, zw, zt
    type(CVMix_tidal_params_type) :: CVmix_tidal_params_user
    error stop 'cvmix_standin: CVMix_compute_Simmons_invariant called'
  end subroutine CVMix_compute_Simmons_invariant
  subroutine CVMix_coeffs_tidal(Mdiff_out, Tdiff_out, Nsqr, OceanDepth, SimmonsCoeff, vert_dep, nlev, max_nlev, cvmix_params, &
                                CVmix_tidal_params_user)
    real(8), dimension(:) :: Mdiff_out, Tdiff_out, Nsqr, vert_dep
    real(8) :: OceanDepth, SimmonsCoeff
    integer :: nlev, max_nlev
    type(CVMix_global_params_type) :: cvmix_params
    type(CVMix_tidal_params_type) :: CVmix_tidal_params_user
    error stop 'cvmix_standin: CVMix_coeffs_tidal called'
  end subroutine CVMix_coeffs_tidal
end module CVMix_tidal
