! ------------------------------------------------------------------------------
! TEST INFRASTRUCTURE (oracle) -- not product code.
!
! C-callable harness around the *reference's own* BLOM modules (compiled from
! /root/reference by oracle/Makefile into oracle/_ref/<cfg>/libblomref.so).
! It exposes
!   ref_dims        compile-time dimensions of this build
!   ref_setup       xcspmd + bigrid(depths) + the inivar_* of every module that
!                   builds here (the netCDF/CVMix dependent ones are absent)
!   ref_field       address/shape of a reference module array (numpy view)
!   ref_set_* / ref_get_*   scalar namelist-type options (mod_rdlim is netCDF
!                   bound, so options are poked directly into their modules)
!   ref_stage       call one reference stage with the (m,n,mm,nn,k1m,k1n) sextuple
!   ref_xctilr      the reference halo update
! The stages are the reference's own routines, called, never restated -- with four exceptions at the end of this file, marked
! there: the pseudo-stages `difest_ustar3`, `cmnfld2_kfpla`, `difest_p`, `mxlayr_tail` restate a few lines each of routines the PLAIN
! builds cannot compile (mod_difest / mod_cmnfld_routines / mod_mxlayr need CVMix or netCDF); the `_x*` builds run the real routines
! for the same pseudo-stages and agree bit for bit (tests/test_xcheck_*.py).
! ------------------------------------------------------------------------------
module ref_harness

  use iso_c_binding
  use dimensions,    only: idm, jdm, kdm, itdm, jtdm
  use mod_xc
  use mod_config,    only: expcnf
  use mod_time,      only: baclin, batrop, delt1, dlt, lstep, nstep, nday_in_year, nday_of_year, nstep_in_day, xmi, l1mi, l2mi, l3mi, l4mi, l5mi
  use mod_grid
  use mod_state
  use mod_eos,       only: pref, inieos
  use mod_bigrid,    only: bigrid
  use mod_checksum,  only: csdiag
  use mod_vcoord,    only: vcoord_tag, sigmar, sigint, inivar_vcoord
  use mod_pgforc
  use mod_momtum
  use mod_barotp
  use mod_tmsmt
  use mod_diffusion
  use mod_utility
  use mod_forcing
  use mod_advect,    only: advect, advmth
  use mod_cppm,      only: init_cppm, cppm_compatibility, cppm_limiting
  use mod_pbcor,     only: pbcor1, pbcor2, bmcmth
  use mod_diffus,    only: diffus
  use mod_diapfl,    only: diapfl
  use mod_convec,    only: convec
  use mod_idlage,    only: idlage_step
  use mod_budget,    only: budget_sums, cnsvdi
  use mod_tracers,   only: ntr, trc, trcold, uflxtr, vflxtr, trflx, inivar_tracers
  use mod_cmnfld,    only: inivar_cmnfld, nslpx, nslpy, nnslpx, nnslpy, bfsqi, bfsqf, bfsql, z, dz, mld, mldl82, dpml
#ifdef XCHECK_EDDTRA
  ! cross-check builds only (oracle/Makefile *_xed): the reference's real mod_eddtra, compiled against a stand-in for mod_difest
  use mod_eddtra,    only: eddtra, init_eddtra, inivar_eddtra, mlrmth, ce, tau_mlr, tau_growing_hbl, tau_decaying_hbl, &
                            tau_growing_hml, tau_decaying_hml, lfmin, mlbl_max_ratio, hbl_tf, hml_tf1, hml_tf, wpup_tf
  use mod_difest,    only: OBLdepth        ! the stand-in's array (oracle/xcheck/mod_difest_standin.F90)
  use mod_cmnfld_routines, only: cmnfld1, cmnfld2, cmnfld_bfsqi_ale
#endif
#ifdef WITH_ALE_VDIFF
  ! builds *_vdf: the reference's phy/mod_ale_vdiff.F90 (no stand-in involved: a pin)
  use mod_ale_vdiff, only: ale_vdifft, ale_vdiffm
#endif
#ifdef XCHECK_ALE
  ! cross-check builds only (oracle/Makefile *_xale): the reference's real mod_ale_regrid_remap against the mod_dia stand-in,
  ! its real mod_ale_forcing against the mod_swabs stand-in
  use mod_ale_regrid_remap, only: readnml_ale_regrid_remap, init_ale_regrid_remap, ale_regrid_remap
  use mod_vcoord,    only: plevel
  use mod_ale_forcing, only: ale_forcing
  use mod_swabs,     only: swamxd, swfc1, swfc2, swal1, swal2
#endif
#ifdef XCHECK_ML
  ! cross-check builds only (oracle/Makefile *_xml): the reference's real mod_mxlayr (stand-ins: mod_nctools, mod_swabs) and
  ! mod_thermf_channel (stand-in: mod_ben02), its real mod_niw
  use mod_mxlayr,    only: mxlayr, inivar_mxlayr, rm0, rm5, mlrttp, mtkeus, mtkeni, mtkebf, mtkers, mtkepe, mtkeke, pbrnda
  use mod_niw,       only: niwgf, niwbf, niwlf, idkedt, inivar_niw, niw_ke_tendency, uml, vml, umlres, vmlres
  use mod_swabs,     only: swamxd, swfc1, swfc2, swal1, swal2
  use mod_thermf_channel, only: thermf_channel
  use mod_ben02,     only: ntda
#endif
#ifdef XCHECK_DF
  ! cross-check builds only (oracle/Makefile *_xdf): the reference's real mod_difest (stand-ins: the CVMix modules, interface only,
  ! and mod_tidaldissip's one array); mod_seaice is the reference's own
  use mod_difest,    only: difest_isobml, inivar_difest
  use mod_tidaldissip, only: twedon
  use mod_seaice,    only: ficem
  use mod_tke,       only: initke, Prod, Buoy, Shear2, L_scale, sqrt2, cmu_fac1, cmu_fac2, cmu_fac3, tke_exp1, gls_exp1, gls_fac6, &
                            gls_s0, gls_s1, gls_s2, gls_s4, gls_s5, gls_s6, gls_b0, gls_b1, gls_b2, gls_b3, gls_b4, gls_b5
#endif
  use mod_ifdefs,    only: use_TRC
  use mod_temmin,    only: temmin

  implicit none
  private
  real(8), allocatable, target, save :: trflx_ij(:,:,:)
  integer :: nt_

  interface
    subroutine ref_capture_r8(a, out) bind(C, name='ref_capture_r8')
      import :: c_double, c_ptr
      real(c_double) :: a(*)
      type(c_ptr)    :: out
    end subroutine
    subroutine ref_capture_i4(a, out) bind(C, name='ref_capture_i4')
      import :: c_int, c_ptr
      integer(c_int) :: a(*)
      type(c_ptr)    :: out
    end subroutine
    subroutine ref_poke_i4(a, v) bind(C, name='ref_poke_i4')
      import :: c_int
      integer(c_int) :: a
      integer(c_int), value :: v
    end subroutine
  end interface

contains

  function cstr(s) result(f)
    character(kind=c_char), intent(in) :: s(*)
    character(len=80) :: f
    integer :: i
    f = ' '
    do i = 1, 80
      if (s(i) == c_null_char) exit
      f(i:i) = s(i)
    end do
  end function cstr

  subroutine ref_dims(d) bind(C, name='ref_dims')
    integer(c_int), intent(out) :: d(8)
    d(1) = idm; d(2) = jdm; d(3) = kdm; d(4) = nbdy
    d(5) = itdm; d(6) = jtdm; d(7) = ntr; d(8) = nreg
  end subroutine ref_dims

  subroutine ref_setup(depth) bind(C, name='ref_setup')
    real(c_double), intent(in) :: depth(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy)
    call xcspmd
    depths(:,:) = depth(:,:)
    call bigrid(depths)
    call inivar_tracers
    call inivar_vcoord
    call inivar_state
    call inivar_pgforc
    call inivar_momtum
    call inivar_barotp
    call inivar_tmsmt
    call inivar_diffusion
    call inivar_utility
    call inivar_forcing
    call inivar_cmnfld
    call inieos
    ! mask of the global sums: restatement of phy/mod_inigeo.F90:189-208 (mod_inigeo itself needs netCDF) --
    ! ip, without the seam and halo rows of an arctic patch
    ips(:,:) = ip(:,:)
    if (nreg == 2) ips(:,jj:jj+nbdy) = 0
  end subroutine ref_setup

  ! The reference's tracer count is a run-time quantity: ntr = ntrocn + ntrtke + ntrgls + ntriag + ntrbgc
  ! (trc/mod_tracers.F90:116-126), its arrays are allocated with it (:211-262) and every stage loops `do nt = 1,ntr`.
  ! The builds here have no iHAMOCC (ntrbgc = 0), so this gives the reference's own stages MORE tracers to carry: the
  ! tracer arrays are re-allocated for nnew tracers, the new ones start as copies of what inivar_tracers left in tracer 1
  ! (its spval / zero-flux patterns), and ntr -- PROTECTED -- is set through its address.  Tracers beyond the compiled
  ! ones are plain passive tracers to every stage (not TKE, not ideal age), as the bgc tracers are.
  subroutine ref_set_ntr(nnew) bind(C, name='ref_set_ntr')
    integer(c_int), value :: nnew
    real(8), allocatable :: t4(:,:,:,:), o4(:,:,:,:), u3(:,:,:), v3(:,:,:), f3(:,:,:), c3(:,:,:)
    integer :: nt, nold
    nold = ntr
    if (nnew < 1 .or. nold < 1 .or. nnew == nold) return
    call move_alloc(trc, t4); call move_alloc(trcold, o4)
    call move_alloc(uflxtr, u3); call move_alloc(vflxtr, v3); call move_alloc(trflx, f3)
    allocate(trc(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,2*kdm,nnew), trcold(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,kdm,nnew))
    allocate(uflxtr(nnew,1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy), vflxtr(nnew,1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy))
    allocate(trflx(nnew,1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy))
    do nt = 1, nnew
      trc(:,:,:,nt) = t4(:,:,:,min(nt,nold)); trcold(:,:,:,nt) = o4(:,:,:,min(nt,nold))
      uflxtr(nt,:,:) = u3(min(nt,nold),:,:); vflxtr(nt,:,:) = v3(min(nt,nold),:,:); trflx(nt,:,:) = f3(min(nt,nold),:,:)
    end do
    if (allocated(trc_corr)) then                       ! phy/mod_forcing.F90:272 (ale_vdifft accumulates into it per tracer)
      call move_alloc(trc_corr, c3)
      allocate(trc_corr(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nnew))
      do nt = 1, nnew
        trc_corr(:,:,nt) = c3(:,:,min(nt,nold))
      end do
    end if
    call ref_poke_i4(ntr, nnew)
  end subroutine ref_set_ntr

  subroutine ref_set_real(name, v, ierr) bind(C, name='ref_set_real')
    character(kind=c_char), intent(in) :: name(*)
    real(c_double), value :: v
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('brine_mlbase_frac'); brine_mlbase_frac = v
#if defined(XCHECK_ALE) || defined(XCHECK_ML)
      case ('swamxd'); swamxd = v
#endif
#ifdef XCHECK_ML
      case ('rm0'); rm0 = v
      case ('rm5'); rm5 = v
      case ('niwgf'); niwgf = v
      case ('niwbf'); niwbf = v
      case ('niwlf'); niwlf = v
      ! mod_forcing: the relaxation of thermf (phy/mod_forcing.F90:53-62, :84)
      case ('trxday'); trxday = v
      case ('srxday'); srxday = v
      case ('trxdpt'); trxdpt = v
      case ('srxdpt'); srxdpt = v
      case ('trxlim'); trxlim = v
      case ('srxlim'); srxlim = v
      case ('sref'); sref = v
      case ('area'); area = v
      case ('xmi'); xmi = v
#endif
#ifdef XCHECK_EDDTRA
      case ('ce'); ce = v
      case ('tau_mlr'); tau_mlr = v
      case ('lfmin'); lfmin = v
      case ('mlbl_max_ratio'); mlbl_max_ratio = v
      case ('tau_growing_hbl'); tau_growing_hbl = v
      case ('tau_decaying_hbl'); tau_decaying_hbl = v
      case ('tau_growing_hml'); tau_growing_hml = v
      case ('tau_decaying_hml'); tau_decaying_hml = v
#endif
#ifdef XCHECK_DF
      case ('egc'); egc = v
      case ('eggam'); eggam = v
      case ('eglsmn'); eglsmn = v
      case ('egmndf'); egmndf = v
      case ('egmxdf'); egmxdf = v
      case ('egidfq'); egidfq = v
      case ('rhiscf'); rhiscf = v
      case ('ri0'); ri0 = v
      case ('tkepf'); tkepf = v
#endif
      case ('baclin'); baclin = v
      case ('batrop'); batrop = v
      case ('delt1');  delt1 = v
      case ('dlt');    dlt = v
      case ('pref');   pref = v; call inieos   ! coefficients depend on pref (mod_eos.F90:105)
      case ('mdv2hi'); mdv2hi = v
      case ('mdv2lo'); mdv2lo = v
      case ('mdv4hi'); mdv4hi = v
      case ('mdv4lo'); mdv4lo = v
      case ('mdc2hi'); mdc2hi = v
      case ('mdc2lo'); mdc2lo = v
      case ('vsc2hi'); vsc2hi = v
      case ('vsc2lo'); vsc2lo = v
      case ('vsc4hi'); vsc4hi = v
      case ('vsc4lo'); vsc4lo = v
      case ('cbar');   cbar = v
      case ('cb');     cb = v
      case ('cwbdts'); cwbdts = v
      case ('cwbdls'); cwbdls = v
      case ('wuv1');   wuv1 = v
      case ('wuv2');   wuv2 = v
      case ('wts1');   wts1 = v
      case ('wts2');   wts2 = v
      case ('wbaro');  wbaro = v
      case ('bdmc1');  bdmc1 = v
      case ('bdmc2');  bdmc2 = v
      case ('iwdfac'); iwdfac = v
      case ('nubmin'); nubmin = v
      case ('vland');  vland = v
      case default; ierr = 1
    end select
  end subroutine ref_set_real

  subroutine ref_get_real(name, v, ierr) bind(C, name='ref_get_real')
    character(kind=c_char), intent(in) :: name(*)
    real(c_double), intent(out) :: v
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('baclin'); v = baclin
      case ('batrop'); v = batrop
      case ('delt1');  v = delt1
      case ('dlt');    v = dlt
      case ('pref');   v = pref
      case ('wbaro');  v = wbaro
      case ('wpgf');   v = wpgf
#ifdef XCHECK_DF
      case ('sqrt2'); v = sqrt2
      case ('cmu_fac1'); v = cmu_fac1
      case ('cmu_fac2'); v = cmu_fac2
      case ('cmu_fac3'); v = cmu_fac3
      case ('tke_exp1'); v = tke_exp1
      case ('gls_exp1'); v = gls_exp1
      case ('gls_fac6'); v = gls_fac6
      case ('gls_s0'); v = gls_s0
      case ('gls_s1'); v = gls_s1
      case ('gls_s2'); v = gls_s2
      case ('gls_s4'); v = gls_s4
      case ('gls_s5'); v = gls_s5
      case ('gls_s6'); v = gls_s6
      case ('gls_b0'); v = gls_b0
      case ('gls_b1'); v = gls_b1
      case ('gls_b2'); v = gls_b2
      case ('gls_b3'); v = gls_b3
      case ('gls_b4'); v = gls_b4
      case ('gls_b5'); v = gls_b5
#endif
      case default; ierr = 1; v = 0
    end select
  end subroutine ref_get_real

  ! 1-D module arrays (plevel: the pressure levels of vcoord_type = 'plevel', phy/mod_vcoord.F90:99)
  subroutine ref_set_vec(name, v, nv, ierr) bind(C, name='ref_set_vec')
    character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: nv
    real(c_double), intent(in) :: v(nv)
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
#ifdef XCHECK_ALE
      case ('plevel'); plevel(1:nv) = v(1:nv)
#endif
      case default; ierr = 1
    end select
  end subroutine ref_set_vec

  subroutine ref_set_int(name, v, ierr) bind(C, name='ref_set_int')
    character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: v
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('lstep');      lstep = v
      case ('nstep');      nstep = v
      case ('nday_in_year'); nday_in_year = v
      case ('vcoord_tag'); vcoord_tag = v
      case ('ltedtp_opt'); ltedtp_opt = v
      case ('ndiff_surface_align'); ndiff_surface_align = (v /= 0)
      case ('bdmtyp');     bdmtyp = v
      case ('iwdflg');     iwdflg = v
      case ('csdiag');     csdiag = (v /= 0)
      case ('cnsvdi');     cnsvdi = (v /= 0)
      case ('bdmldp');     bdmldp = (v /= 0)
#ifdef XCHECK_DF
      case ('eddf2d');     eddf2d = (v /= 0)
      case ('edsprs');     edsprs = (v /= 0)
      case ('edanis');     edanis = (v /= 0)
      case ('redi3d');     redi3d = (v /= 0)
      case ('rhsctp');     rhsctp = (v /= 0)
      case ('edfsmo');     edfsmo = (v /= 0)
      case ('edritp_opt'); edritp_opt = v        ! 1 shear, 2 large scale (phy/mod_diffusion.F90)
      case ('edwmth_opt'); edwmth_opt = v        ! 1 smooth, 2 step
#endif
#ifdef XCHECK_ML
      ! mod_forcing's switches of thermf and mod_time's calendar position (phy/mod_forcing.F90:43-47, phy/mod_time.F90)
      case ('aptflx');     aptflx = (v /= 0)
      case ('apsflx');     apsflx = (v /= 0)
      case ('ditflx');     ditflx = (v /= 0)
      case ('disflx');     disflx = (v /= 0)
      case ('srxbal');     srxbal = (v /= 0)
      case ('nday_of_year'); nday_of_year = v
      case ('nstep_in_day'); nstep_in_day = v
      case ('l1mi'); l1mi = v
      case ('l2mi'); l2mi = v
      case ('l3mi'); l3mi = v
      case ('l4mi'); l4mi = v
      case ('l5mi'); l5mi = v
      case ('ntda'); ntda = v
#endif
      case default; ierr = 1
    end select
  end subroutine ref_set_int

  subroutine ref_get_int(name, v, ierr) bind(C, name='ref_get_int')
    character(kind=c_char), intent(in) :: name(*)
    integer(c_int), intent(out) :: v
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('lstep'); v = lstep
      case ('nstep'); v = nstep
      case ('nreg');  v = nreg
      case ('ntr');   v = ntr
      case ('ii');    v = ii
      case ('jj');    v = jj
      case default; ierr = 1; v = 0
    end select
  end subroutine ref_get_int

  subroutine ref_set_str(name, s, ierr) bind(C, name='ref_set_str')
    character(kind=c_char), intent(in) :: name(*), s(*)
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('expcnf'); expcnf = trim(cstr(s))
      case ('mommth'); mommth = trim(cstr(s))
      case ('pgfmth'); pgfmth = trim(cstr(s))
      case ('advmth'); advmth = trim(cstr(s))
      case ('cppm_compatibility'); cppm_compatibility = trim(cstr(s))
      case ('cppm_limiting'); cppm_limiting = trim(cstr(s))
      case ('bmcmth'); bmcmth = trim(cstr(s))
#ifdef XCHECK_ML
      case ('mlrttp'); mlrttp = trim(cstr(s))
#endif
      case ('eitmth')            ! readnml_diffusion's translation, phy/mod_diffusion.F90:316-327
        if (trim(cstr(s)) == 'intdif') then
          eitmth_opt = eitmth_intdif
        else
          eitmth_opt = eitmth_gm
        end if
      case default; ierr = 1
    end select
  end subroutine ref_set_str

  ! kind: 0 = real(8), 1 = integer(4).  nlev = size of 3rd dimension (1 for 2-D).
  subroutine ref_field(name, ptr, nlev, kind) bind(C, name='ref_field')
    character(kind=c_char), intent(in) :: name(*)
    type(c_ptr), intent(out) :: ptr
    integer(c_int), intent(out) :: nlev, kind
    kind = 0
    nlev = 1
    ptr = c_null_ptr
#define R3(nm, nl) case (#nm); call ref_capture_r8(nm, ptr); nlev = nl
#define R2(nm) case (#nm); call ref_capture_r8(nm, ptr); nlev = 1
#define I2(nm) case (#nm); call ref_capture_i4(nm, ptr); nlev = 1; kind = 1
    select case (trim(cstr(name)))
      ! mod_state
      R3(u, 2*kdm)
      R3(v, 2*kdm)
      R3(dp, 2*kdm)
      R3(dpu, 2*kdm)
      R3(dpv, 2*kdm)
      R3(temp, 2*kdm)
      R3(saln, 2*kdm)
      R3(sigma, 2*kdm)
      R3(uflx, 2*kdm)
      R3(vflx, 2*kdm)
      R3(utflx, 2*kdm)
      R3(vtflx, 2*kdm)
      R3(usflx, 2*kdm)
      R3(vsflx, 2*kdm)
      R3(p, kdm+1)
      R3(pu, kdm+1)
      R3(pv, kdm+1)
      R3(phi, kdm+1)
      R3(cau, kdm)
      R3(cav, kdm)
      R3(ubflxs, 3)
      R3(vbflxs, 3)
      R3(ub, 2)
      R3(vb, 2)
      R3(pb, 2)
      R3(pbu, 2)
      R3(pbv, 2)
      R3(ubflxs_p, 2)
      R3(vbflxs_p, 2)
      R2(pb_p)
      R2(pbu_p)
      R2(pbv_p)
      R2(ubcors_p)
      R2(vbcors_p)
      R2(sealv)
      case ('kfpla'); call ref_capture_i4(kfpla, ptr); nlev = 2; kind = 1
      ! masks (mod_xc)
      I2(ip)
      I2(iu)
      I2(iv)
      I2(iq)
      ! mod_grid
      R2(scqx)
      R2(scqy)
      R2(scpx)
      R2(scpy)
      R2(scux)
      R2(scuy)
      R2(scvx)
      R2(scvy)
      R2(scq2)
      R2(scp2)
      R2(scu2)
      R2(scv2)
      R2(scq2i)
      R2(scp2i)
      R2(scuxi)
      R2(scuyi)
      R2(scvxi)
      R2(scvyi)
      R2(depths)
      R2(corioq)
      R2(coriop)
      R2(betafp)
      ! mod_pgforc
      R3(pgfx, 2*kdm)
      R3(pgfy, 2*kdm)
      R3(pgfx_o, kdm)
      R3(pgfy_o, kdm)
      R3(pgfxm, 2)
      R3(pgfym, 2)
      R3(xixp, 2)
      R3(xixm, 2)
      R3(xiyp, 2)
      R3(xiym, 2)
      R2(pgfxm_o)
      R2(pgfym_o)
      R2(xixp_o)
      R2(xixm_o)
      R2(xiyp_o)
      R2(xiym_o)
      ! mod_momtum
      R3(absvor, 2*kdm)
      R3(dpvor, 2*kdm)
      ! mod_barotp
      R3(ubflx, 2)
      R3(vbflx, 2)
      R3(pb_mn, 2)
      R3(ubflx_mn, 2)
      R3(vbflx_mn, 2)
      R3(pvtrop, 2)
      ! mod_tmsmt
      R3(dpold, 2*kdm)
      R3(dpuold, kdm)
      R3(dpvold, kdm)
      ! mod_vcoord, mod_temmin
      R3(sigmar, kdm)
      R3(sigint, kdm)
      R3(temmin, kdm)
      ! mod_diffusion
      R3(difint, kdm)
      R3(nslpx, kdm)
      R3(nslpy, kdm)
      R3(nnslpx, kdm)
      R3(nnslpy, kdm)
      R3(bfsqi, kdm+1)
      R3(bfsqf, kdm+1)
      R3(bfsql, kdm)
      R3(z, kdm+1)
      R3(dz, kdm)
      R3(difiso, kdm)
      R3(difdia, kdm)
      R2(difmxp)
      R2(difmxq)
      R2(difwgt)
      R3(umfltd, 2*kdm)
      R3(vmfltd, 2*kdm)
      R3(umflsm, 2*kdm)
      R3(vmflsm, 2*kdm)
      R3(utfltd, 2*kdm)
      R3(vtfltd, 2*kdm)
      R3(utflsm, 2*kdm)
      R3(vtflsm, 2*kdm)
      R3(utflld, 2*kdm)
      R3(vtflld, 2*kdm)
      R3(usfltd, 2*kdm)
      R3(vsfltd, 2*kdm)
      R3(usflsm, 2*kdm)
      R3(vsflsm, 2*kdm)
      R3(usflld, 2*kdm)
      R3(vsflld, 2*kdm)
      ! mod_utility
      R2(utotm)
      R2(vtotm)
      R2(utotn)
      R2(vtotn)
      R2(uflux)
      R2(vflux)
      R2(uflux2)
      R2(vflux2)
      R2(uflux3)
      R2(vflux3)
      R2(umax)
      R2(vmax)
      R2(util1)
      R2(util2)
      R2(util3)
      R2(util4)
      ! mod_forcing
      R2(taux)
      R2(tauy)
      R2(ustarb)
      ! mod_tracers (allocatable; ntr may be 0)
      case ('trc')
        if (allocated(trc)) then
          call ref_capture_r8(trc, ptr); nlev = 2*kdm*ntr
        end if
      case ('trcold')
        if (allocated(trcold)) then
          call ref_capture_r8(trcold, ptr); nlev = kdm*ntr
        end if
      ! inputs and accumulators of ale_vdifft / ale_vdiffm (mod_diffusion.F90:131-139, mod_forcing.F90:159-191)
      case ('kvisc_m'); call ref_capture_r8(Kvisc_m, ptr); nlev = kdm+1
      case ('kdiff_t'); call ref_capture_r8(Kdiff_t, ptr); nlev = kdm+1
      case ('kdiff_s'); call ref_capture_r8(Kdiff_s, ptr); nlev = kdm+1
      R3(mu_nonloc, kdm+1)
      R3(mv_nonloc, kdm+1)
      R3(t_ns_nonloc, kdm+1)
      R3(s_nb_nonloc, kdm+1)
      R3(t_sw_nonloc, kdm+1)
      R3(t_rs_nonloc, kdm+1)
      R3(s_br_nonloc, kdm+1)
      R3(s_rs_nonloc, kdm+1)
      R2(surflx)
      R2(sswflx)
      R2(surrlx)
      R2(salflx)
      R2(brnflx)
      R2(salrlx)
      R2(salt_corr)
      R3(buoyfl, kdm+1)
      R2(mld)
      R2(mldl82)
      R2(dpml)
#ifdef XCHECK_EDDTRA
      R2(hbl_tf)
      R2(wpup_tf)
      R2(hml_tf1)
      R2(hml_tf)
      R2(OBLdepth)
#endif
#if defined(XCHECK_ALE) || defined(XCHECK_ML)
      R2(swfc1)
      R2(swfc2)
      R2(swal1)
      R2(swal2)
#endif
#ifdef XCHECK_DF
      R2(twedon)
      R2(ficem)
      R2(plat)
      R2(betatp)
      R2(cosang)
      R2(sinang)
      R2(hangle)
      R3(Prod, kdm)
      R3(Buoy, kdm)
      R3(Shear2, kdm)
      R3(L_scale, kdm)
#endif
#ifdef XCHECK_ML
      R2(idkedt)
      R2(mtkeus)
      R2(mtkeni)
      R2(mtkebf)
      R2(mtkers)
      R2(mtkepe)
      R2(mtkeke)
      R2(pbrnda)
      R3(uml, 4)
      R3(vml, 4)
      R3(umlres, 2)
      R3(vmlres, 2)
#endif
      ! mod_forcing: friction velocity and the forcing fields of thermf (phy/mod_forcing.F90:100-175)
      R2(ustar)
      R2(ustar3)
      R2(wstar3)
      R2(ustarw)
      R2(swa)
      R2(nsf)
      R2(hmltfz)
      R2(lip)
      R2(sop)
      R2(eva)
      R2(rnf)
      R2(rfi)
      R2(fmltfz)
      R2(sfl)
      R3(sstclm, 12)
      R3(ricclm, 12)
      R3(sssclm, 12)
      R3(tflxap, 48)
      R3(sflxap, 48)
      R3(tflxdi, 48)
      R3(sflxdi, 48)
      case ('trc_corr')
        if (allocated(trc_corr)) then
          call ref_capture_r8(trc_corr, ptr); nlev = ntr
        end if
      ! trflx(ntr,i,j) has the tracer index first: it is shown as trflx_ij(i,j,ntr) and copied across around ale_vdifft
      case ('trflx')
        if (allocated(trflx)) then
          if (.not. allocated(trflx_ij)) allocate(trflx_ij(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,ntr))
          if (size(trflx_ij,3) /= ntr) then
            deallocate(trflx_ij); allocate(trflx_ij(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,ntr))
          end if
          do nt_ = 1, ntr
            trflx_ij(:,:,nt_) = trflx(nt_,:,:)
          end do
          call ref_capture_r8(trflx_ij, ptr); nlev = ntr
        end if
      case default
        nlev = 0
    end select
  end subroutine ref_field

  subroutine ref_stage(name, m, n, mm, nn, k1m, k1n, ierr) bind(C, name='ref_stage')
    character(kind=c_char), intent(in) :: name(*)
    integer(c_int), value :: m, n, mm, nn, k1m, k1n
    integer(c_int), intent(out) :: ierr
    ierr = 0
    select case (trim(cstr(name)))
      case ('init_fluxes'); call init_fluxes(m,n,mm,nn,k1m,k1n,.false.)
      case ('tmsmt1');  call tmsmt1(nn)
      case ('initms');  call initms(mm)
      case ('advect');  call advect(m,n,mm,nn,k1m,k1n)
      case ('pbcor1');  call pbcor1(m,n,mm,nn,k1m,k1n)
      case ('diffus');  call diffus(m,n,mm,nn,k1m,k1n)
      case ('pgforc');  call pgforc(m,n,mm,nn,k1m,k1n)
      case ('momtum');  call momtum(m,n,mm,nn,k1m,k1n)
      case ('convec');  call convec(m,n,mm,nn,k1m,k1n)
      ! updtrc (trc/mod_tracers_update.F90:152-170) = hamocc_step (not built) + idlage_step; the latter is called
      case ('updtrc');  call idlage_step(m,n,mm,nn,k1m,k1n)
      case ('diapfl');  call diapfl(n,nn,k1n)
      case ('barotp');  call barotp(m,n,mm,nn,k1m,k1n)
      case ('pbcor2');  call pbcor2(m,n,mm,nn,k1m,k1n)
      case ('tmsmt2');  call tmsmt2(m,mm,nn,k1m)
#ifdef XCHECK_EDDTRA
      case ('eddtra');  call eddtra(m,n,mm,nn,k1m,k1n)
      ! init_eddtra resolves the mixed layer restratification method from the string (phy/mod_eddtra.F90:1773-1806)
      case ('eddtra_init_fox08'); mlrmth = 'fox08'; call inivar_eddtra; call init_eddtra
      case ('eddtra_init_none');  mlrmth = 'none';  call inivar_eddtra; call init_eddtra
      case ('eddtra_init_bod23'); mlrmth = 'bod23'; call inivar_eddtra; call init_eddtra
      case ('cmnfld2'); call cmnfld2(m,n,mm,nn,k1m,k1n)
      case ('cmnfld1'); call cmnfld1(m,n,mm,nn,k1m,k1n)
      case ('cmnfld_bfsqi_ale'); call cmnfld_bfsqi_ale(m,n,mm,nn,k1m,k1n)
#endif
#ifdef XCHECK_DF
      ! difest_init: the module's arrays (inivar_difest; init_difest is CVMix's initialisation and is not called), mod_tke's arrays and
      ! derived constants (initke: cmu_fac1.., tke_exp1, gls_exp1, the stability function coefficients)
      case ('difest_init'); call inivar_difest; call initke
      case ('difest_isobml'); call difest_isobml(m,n,mm,nn,k1m,k1n)
#endif
#ifdef XCHECK_ML
      case ('mxlayr_init'); call inivar_mxlayr; call inivar_niw
      case ('mxlayr')
        if (allocated(trflx_ij)) then          ! trflx is shown as trflx_ij(i,j,ntr): see ref_field
          do nt_ = 1, ntr
            trflx(nt_,:,:) = trflx_ij(:,:,nt_)
          end do
        end if
        call mxlayr(m,n,mm,nn,k1m,k1n)
      case ('niw_ke_tendency'); call niw_ke_tendency(m,n,mm,nn,k1m,k1n)
      ! difest_isobml up to and including niw_ke_tendency (phy/mod_difest.F90:750-790; the diffusivity estimates behind it need
      ! CVMix): the halo updates and the pressure scan as under 'halo_difest', ustar3 RESTATED (:778-786), the real niw_ke_tendency
      case ('difest_isobml_pre')
        call xctilr(u, 1,2*kk, 2,2, halo_uv)
        call xctilr(v, 1,2*kk, 2,2, halo_vv)
        call xctilr(ubflxs_p, 1,2, 2,2, halo_uv)
        call xctilr(vbflxs_p, 1,2, 2,2, halo_vv)
        call xctilr(pbu, 1,2, 2,2, halo_us)
        call xctilr(pbv, 1,2, 2,2, halo_vs)
        call difest_p(nn)
        call difest_ustar3
        call niw_ke_tendency(m,n,mm,nn,k1m,k1n)
      case ('thermf')
        if (allocated(trflx_ij)) then
          do nt_ = 1, ntr
            trflx(nt_,:,:) = trflx_ij(:,:,nt_)
          end do
        end if
        call thermf_channel(m,n,mm,nn,k1m,k1n)
        if (allocated(trflx_ij)) then          ! show the tracer fluxes it computed (trflx_ij: see ref_field)
          do nt_ = 1, ntr
            trflx_ij(:,:,nt_) = trflx(nt_,:,:)
          end do
        end if
#endif
#ifdef WITH_ALE_VDIFF
      case ('ale_vdifft')
        if (allocated(trflx_ij)) then
          do nt_ = 1, ntr
            trflx(nt_,:,:) = trflx_ij(:,:,nt_)
          end do
        end if
        call ale_vdifft(m,n,mm,nn,k1m,k1n)
      case ('ale_vdiffm'); call ale_vdiffm(m,n,mm,nn,k1m,k1n)
#endif
#ifdef XCHECK_ALE
      ! ale_init: the group &ALE_REGRID_REMAP of the file `limits` in the working directory, then the reconstruction and
      ! remapping structures (mod_blom_init.F90 calls the two in this order); vcoord_tag must have been set before
      case ('ale_init')
        call readnml_ale_regrid_remap
        call init_ale_regrid_remap
      case ('ale_regrid_remap'); call ale_regrid_remap(m,n,mm,nn,k1m,k1n)
      case ('ale_forcing'); call ale_forcing(m,n,mm,nn,k1m,k1n)
#endif
      ! Halo updates the reference performs inside stages that cannot be built here
      ! (netCDF/CVMix).  Only the xctilr calls are reproduced, by calling xctilr.
      case ('init_cppm');  call init_cppm          ! phy/mod_cppm.F90:2504 (called from blom_init)
      case ('halo_cmnfld2')   ! phy/mod_cmnfld_routines.F90:1171-1196
        call xctilr(temp, 1, 2*kk, 3, 3, halo_ps)
        call xctilr(saln, 1, 2*kk, 3, 3, halo_ps)
        call cmnfld2_kfpla(n)
      case ('halo_difest')    ! phy/mod_difest.F90:750-772
        call xctilr(u, 1,2*kk, 2,2, halo_uv)
        call xctilr(v, 1,2*kk, 2,2, halo_vv)
        call xctilr(ubflxs_p, 1,2, 2,2, halo_uv)
        call xctilr(vbflxs_p, 1,2, 2,2, halo_vv)
        call xctilr(pbu, 1,2, 2,2, halo_us)
        call xctilr(pbv, 1,2, 2,2, halo_vs)
        call difest_p(nn)
      case ('mxlayr_tail');  call mxlayr_tail(nn, k1n)
      ! the halo updates of difest_lateral_hybrid and difest_vertical_hybrid (phy/mod_difest.F90:826-831, :877-878; the
      ! routines themselves need CVMix)
      case ('halo_difest_hyb')
        call xctilr(u, 1,2*kk, 2,2, halo_uv)
        call xctilr(v, 1,2*kk, 2,2, halo_vv)
        call xctilr(ubflxs_p, 1,2, 2,2, halo_uv)
        call xctilr(vbflxs_p, 1,2, 2,2, halo_vv)
        call xctilr(pbu, 1,2, 2,2, halo_us)
        call xctilr(pbv, 1,2, 2,2, halo_vs)
      case ('halo_difest_vert')
        call xctilr(u(1-nbdy,1-nbdy,k1n), 1,kk, 1,1, halo_uv)
        call xctilr(v(1-nbdy,1-nbdy,k1n), 1,kk, 1,1, halo_vv)
      case default; ierr = 1
    end select
  end subroutine ref_stage

#ifdef XCHECK_ML
  subroutine difest_ustar3
    ! RESTATEMENT of phy/mod_difest.F90:778-786
    integer :: i, j, l
    do j = 1,jj
      do l = 1,isp(j)
        do i = max(1,ifp(j,l)),min(ii,ilp(j,l))
          ustar3(i,j) = ustar(i,j)**3
        end do
      end do
    end do
  end subroutine difest_ustar3
#endif

  subroutine cmnfld2_kfpla(n)
    ! RESTATEMENT of the halo update of kfpla through util1, phy/mod_cmnfld_routines.F90:1176-1196
    ! (cmnfld2 itself is outside the hot path; eddtra reads kfpla(i-1,j,n), kfpla(i,j-1,n)).
    integer, intent(in) :: n
    integer :: i, j, l
    do j = 1, jj
      do l = 1, isp(j)
        do i = max(1, ifp(j,l)), min(ii, ilp(j,l))
          util1(i,j) = kfpla(i,j,n)
        enddo
      enddo
    enddo
    call xctilr(util1, 1, 1, 2, 2, halo_ps)
    do j = - 1, jj + 2
      do l = 1, isp(j)
        do i = max(- 1, ifp(j,l)), min(ii + 2, ilp(j,l))
          kfpla(i,j,n) = nint(util1(i,j))
        enddo
      enddo
    enddo
  end subroutine cmnfld2_kfpla

  subroutine difest_p(nn)
    ! RESTATEMENT of "Update layer interface pressure", phy/mod_difest.F90:761-772: difest_isobml
    ! itself is outside the hot path, but advect/remap consume p out to ii+3.
    integer, intent(in) :: nn
    integer :: i, j, k, l
    do j = -2,jj+3
      do k = 1,kk
        do l = 1,isp(j)
          do i = max(-2,ifp(j,l)),min(ii+3,ilp(j,l))
            p(i,j,k+1) = p(i,j,k)+dp(i,j,k+nn)
          end do
        end do
      end do
    end do
  end subroutine difest_p

  subroutine mxlayr_tail(nn, k1n)
    ! RESTATEMENT of phy/mod_mxlayr.F90:1266-1310 ("store 'new' layer thicknesses in
    ! -dpu,dpv-"): mod_mxlayr itself cannot be built here (netCDF), and its bulk mixed
    ! layer physics is out of scope, but the next step relies on this halo update of
    ! dp(:,:,k1n:) and on dpu/dpv at the new time level being consistent with it.
    ! The arithmetic is identical to the tail of tmsmt2 (phy/mod_tmsmt.F90:352-391).
    integer, intent(in) :: nn, k1n
    integer :: i, j, k, l, kn
    real(8) :: q
    call xctilr(dp(1-nbdy,1-nbdy,k1n), 1,kk, 3,3, halo_ps)
    do j = -2,jj+2
      do k = 1,kk
        kn = k+nn
        do l = 1,isp(j)
          do i = max(-2,ifp(j,l)),min(ii+2,ilp(j,l))
            p(i,j,k+1) = p(i,j,k)+dp(i,j,kn)
          end do
        end do
      end do
    end do
    do j = -1,jj+2
      do k = 1,kk
        kn = k+nn
        do l = 1,isu(j)
          do i = max(-1,ifu(j,l)),min(ii+2,ilu(j,l))
            q = min(p(i,j,kk+1),p(i-1,j,kk+1))
            dpu(i,j,kn)= &
                 .5*((min(q,p(i-1,j,k+1))-min(q,p(i-1,j,k))) &
                 +(min(q,p(i  ,j,k+1))-min(q,p(i  ,j,k))))
          end do
        end do
        do l = 1,isv(j)
          do i = max(-1,ifv(j,l)),min(ii+2,ilv(j,l))
            q = min(p(i,j,kk+1),p(i,j-1,kk+1))
            dpv(i,j,kn)= &
                 .5*((min(q,p(i,j-1,k+1))-min(q,p(i,j-1,k))) &
                    +(min(q,p(i,j  ,k+1))-min(q,p(i,j  ,k))))
          end do
        end do
      end do
    end do
  end subroutine mxlayr_tail

  ! The reference's own field checksum (chksum -> xccrc, phy/mod_checksum.F90:41-74,
  ! phy/mod_xc.F90:4164-4205): CRC-32 over the tile interior where the grid's mask is 1.
  subroutine ref_xccrc(a, nlev, itype, crc) bind(C, name='ref_xccrc')
    integer(c_int), value :: nlev, itype
    real(c_double), intent(in) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nlev)
    integer(c_int), intent(out) :: crc
    select case (itype)
      case (halo_ps, halo_pv); call xccrc(crc, a, nlev, ip, itype)
      case (halo_qs, halo_qv); call xccrc(crc, a, nlev, iq, itype)
      case (halo_us, halo_uv); call xccrc(crc, a, nlev, iu, itype)
      case default;            call xccrc(crc, a, nlev, iv, itype)
    end select
  end subroutine ref_xccrc

  ! xcsum (phy/mod_xc.F90:4116-4161): the reference's reproducible masked sum of a 2-D array; mask by grid type
  subroutine ref_xcsum(a, itype, s) bind(C, name='ref_xcsum')
    integer(c_int), value :: itype
    real(c_double), intent(inout) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy)
    real(c_double), intent(out) :: s
    select case (itype)
      case (halo_ps, halo_pv); call xcsum(s, a, ips)
      case (halo_qs, halo_qv); call xcsum(s, a, iq)
      case (halo_us, halo_uv); call xcsum(s, a, iu)
      case default;            call xcsum(s, a, iv)
    end select
  end subroutine ref_xcsum

  ! budget_sums (phy/mod_budget.F90:95): its sums are private to mod_budget; what it leaves behind in
  ! util1, util2 (the mass weighted column sums) is public and is what the tests compare
  subroutine ref_budget_sums(ncall, n, nn) bind(C, name='ref_budget_sums')
    integer(c_int), value :: ncall, n, nn
    call budget_sums(ncall, n, nn)
  end subroutine ref_budget_sums

  subroutine ref_xctilr(a, l1, ld, mh, nh, itype) bind(C, name='ref_xctilr')
    integer(c_int), value :: l1, ld, mh, nh, itype
    real(c_double), intent(inout) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,ld)
    call xctilr(a, l1, ld, mh, nh, itype)
  end subroutine ref_xctilr

end module ref_harness
