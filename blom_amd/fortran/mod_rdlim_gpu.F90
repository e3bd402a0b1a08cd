! ------------------------------------------------------------------------------
! module mod_rdlim_gpu -- the namelist input of the reference for the host of the device-resident dynamical core.
! Reads the file `limits` (or `ocn_in`) exactly as the reference's readers do -- groups &LIMITS
! (phy/mod_rdlim.F90:137-175), &VCOORD (phy/mod_vcoord.F90:818-823), &DIFFUSION (phy/mod_diffusion.F90:214-218) and, for the
! hybrid coordinates, &ALE_REGRID_REMAP (phy/mod_ale_regrid_remap.F90:1193-1201),
! each declared with the reference's full variable list so that its input files are accepted unchanged -- and hands
! the variables the dynamical core uses to the device library (blomgpu_set_*).  Everything else in the groups is
! read and ignored here: it belongs to parts of the model outside this path (I/O, forcing, restart, diagnostics).
! ------------------------------------------------------------------------------
module mod_rdlim_gpu

  use mod_blomgpu
  implicit none
  private
  public :: rdlim_gpu
  logical, public :: advect_cppm = .false.           ! advmth = 'cppm': blom_init calls init_cppm (phy/mod_blom_init.F90)
  logical, public :: hybrid_coordinate = .false.     ! vcoord_type /= 'isopyc_bulkml': blom_step takes the other branch

contains

  ! nstep1, nstep2: the model is integrated from time step nstep1 to nstep2 (phy/mod_rdlim.F90:1149-1156)
  subroutine rdlim_gpu(found, nstep1, nstep2, baclin_out)
    logical, intent(out) :: found
    integer, intent(inout) :: nstep1, nstep2
    real(8), intent(inout) :: baclin_out

    ! &LIMITS
    integer :: nday1 = 0, nday2 = 0, idate = 0, idate0 = 0, jwtype = 3, itest = 1, jtest = 1, rstfrq = 1, rstfmt = 1, &
               rstcmp = 1, iotype = 0
    character(len=256) :: runid = 'unset', runtyp = 'unset', rpoint = 'unset', expcnf = 'unset', grfile = 'unset', &
               icfile = 'unset', tdfile = 'unset', ccfile = 'unset', svfile = 'unset', scfile = 'unset', atm_path = 'unset'
    character(len=80) :: mommth = 'enscon', pgfmth = 'geopotential', bmcmth = 'uc', advmth = 'remap', &
               cppm_compatibility = 'full', cppm_limiting = 'non_oscillatory', mldmth = 'lev82', mlrmth = 'fox08', &
               mlrttp = 'constant', swamth = 'jerlov', chlopt = 'climatology', wavsrc = 'none'
    logical :: woa_nuopc_provided = .false., aptflx = .false., apsflx = .false., ditflx = .false., disflx = .false., &
               srxbal = .false., smtfrc = .false., sprfac = .false., cnsvdi = .false., csdiag = .false., &
               use_stream_relaxation = .false., use_stream_dust = .false., use_diag = .false.
    real(8) :: pref = 2000.d4, baclin = 0.d0, batrop = 0.d0, mdv2hi = 0.d0, mdv2lo = 0.d0, mdv4hi = 0.d0, mdv4lo = 0.d0, &
               mdc2hi = 0.d0, mdc2lo = 0.d0, vsc2hi = 0.d0, vsc2lo = 0.d0, vsc4hi = 0.d0, vsc4lo = 0.d0, cbar = 0.d0, &
               cb = 0.d0, cwbdts = 0.d0, cwbdls = 0.d0, &
               ! mixed layer restratification, defaults of phy/mod_eddtra.F90:53-94
               ce = .06d0, cl = .25d0, tau_mlr = 86400.d0, tau_growing_hbl = 300.d0, &
               tau_decaying_hbl = 86400.d0, tau_growing_hml = 3600.d0, tau_decaying_hml = 259200.d0, lfmin = 5.d3, mstar = .5d0, &
               nstar = .066d0, wpup_min = 1.d-3, mlbl_max_ratio = 3.d0, rm0 = 0.d0, rm5 = 0.d0, niwgf = 0.d0, niwbf = 0.d0, &
               niwlf = 0.d0, trxday = 0.d0, srxday = 0.d0, trxdpt = 0.d0, srxdpt = 0.d0, trxlim = 0.d0, srxlim = 0.d0, &
               brine_mlbase_frac = 0.d0
    ! &VCOORD
    character(len=80) :: vcoord_type = 'isopyc_bulkml', sigref_spec = 'inicon', plevel_spec = 'inflation', sigdia_spec = 'inicon'
    real(8) :: dpmin_surface = 1.5d0, dpmin_inflation_factor = 1.d0, sra_clim_ts = 5.d0, sra_param_ts1 = 5.d0, &
               sra_param_ts2 = 10.d0, sra_massfrac_bot = .01d0, sra_massfrac_eps = .0001d0
    real(8) :: sigref(512), plevel(512), sigdia(512), sigref_fun_spec(8), sigdia_fun_spec(8)
    logical :: sigref_adaption = .false.
    ! &DIFFUSION
    real(8) :: egc = 0.d0, eggam = 0.d0, eglsmn = 0.d0, egmndf = 0.d0, egmxdf = 0.d0, egidfq = 0.d0, rhiscf = 0.d0, &
               ri0 = 0.d0, bdmc1 = 5.d-8, bdmc2 = 1.d-5, iwdfac = .06d0, nubmin = 1.d-6, tkepf = 0.d0, lau10f = 0.d0
    integer :: bdmtyp = 2, iwdflg = 1
    logical :: eddf2d = .false., edsprs = .false., edanis = .false., redi3d = .false., rhsctp = .false., edfsmo = .false., &
               bdmldp = .false., smobld = .false., ndiff_surface_align = .false.
    character(len=256) :: tbfile = 'unset'
    character(len=80) :: lngmtp = 'none', eitmth = 'gm', edritp = 'large scale', edwmth = 'smooth', ltedtp = 'layer'

    ! &ALE_REGRID_REMAP, defaults of phy/mod_ale_regrid_remap.F90:69-95
    character(len=80) :: reconstruction_method = 'ppm', density_limiting = 'monotonic', tracer_limiting = 'non_oscillatory', &
               velocity_limiting = 'non_oscillatory', regrid_method = 'nudge'
    logical :: density_pc_upper_bndr = .false., density_pc_lower_bndr = .false., tracer_pc_upper_bndr = .true., &
               tracer_pc_lower_bndr = .false., velocity_pc_upper_bndr = .true., velocity_pc_lower_bndr = .false.
    integer :: upper_bndr_ord = 6, lower_bndr_ord = 4, k_range_plevel = 1, dktzu = 4, dktzl = 2
    real(8) :: dpmin_interior = .1d0, regrid_nudge_ts = 86400.d0, stab_fac_limit = .75d0, dpvar_fac = .75d0, &
               smooth_diff_max = 50000.d0
    real(8), parameter :: spval_ = huge(1.d0), onem = 9806.d0
    real(8) :: dpmin
    integer :: k

    character(len=80) :: nlfnm
    integer :: nfu, ios, lstep, nstep_in_day
    logical :: fexist
    real(8), parameter :: epsilt = 1.d-11              ! phy/mod_constants.F90

    namelist /limits/ nday1,nday2,idate,idate0,runid,runtyp, rpoint, expcnf, &
         grfile,icfile,woa_nuopc_provided,pref,baclin,batrop, &
         mdv2hi,mdv2lo,mdv4hi,mdv4lo,mdc2hi,mdc2lo, &
         vsc2hi,vsc2lo,vsc4hi,vsc4lo,cbar,cb,cwbdts,cwbdls, &
         mommth,pgfmth,bmcmth,advmth,cppm_compatibility,cppm_limiting, &
         mldmth,mlrmth,ce,cl,tau_mlr,tau_growing_hbl,tau_decaying_hbl, &
         tau_growing_hml,tau_decaying_hml,lfmin,mstar,nstar,wpup_min, &
         mlbl_max_ratio,mlrttp,rm0,rm5,tdfile,niwgf,niwbf,niwlf, &
         swamth,jwtype,chlopt,ccfile,svfile, &
         trxday,srxday,trxdpt,srxdpt,trxlim,srxlim, &
         aptflx,apsflx,ditflx,disflx,srxbal,scfile, &
         wavsrc,smtfrc,sprfac,brine_mlbase_frac, &
         atm_path, itest,jtest, cnsvdi, csdiag, &
         rstfrq,rstfmt,rstcmp,iotype,use_stream_relaxation, use_stream_dust, use_diag
    namelist /vcoord/ vcoord_type, dpmin_surface, dpmin_inflation_factor, sigref_spec, plevel_spec, sigdia_spec, &
         sigref_fun_spec, sigdia_fun_spec, sigref, plevel, sigdia, sigref_adaption, sra_clim_ts, sra_param_ts1, &
         sra_param_ts2, sra_massfrac_bot, sra_massfrac_eps
    namelist /diffusion/ egc, eggam, eglsmn, egmndf, egmxdf, egidfq, rhiscf, ri0, &
         bdmc1, bdmc2, bdmldp, iwdflg, iwdfac, nubmin, tkepf, lau10f, bdmtyp, &
         eddf2d, edsprs, edanis, redi3d, rhsctp, tbfile, edfsmo, smobld, &
         lngmtp, eitmth, edritp, edwmth, ltedtp, ndiff_surface_align

    namelist /ale_regrid_remap/ reconstruction_method, upper_bndr_ord, lower_bndr_ord, density_limiting, tracer_limiting, &
         velocity_limiting, density_pc_upper_bndr, density_pc_lower_bndr, tracer_pc_upper_bndr, tracer_pc_lower_bndr, &
         velocity_pc_upper_bndr, velocity_pc_lower_bndr, dpmin_interior, regrid_method, k_range_plevel, regrid_nudge_ts, &
         stab_fac_limit, dpvar_fac, smooth_diff_max, dktzu, dktzl

    found = .false.
    hybrid_coordinate = .false.
    plevel(:) = spval_
    nlfnm = 'ocn_in'                                   ! phy/mod_rdlim.F90:160-172
    inquire (file=nlfnm, exist=fexist)
    if (.not. fexist) then
      nlfnm = 'limits'
      inquire (file=nlfnm, exist=fexist)
    end if
    if (.not. fexist) return
    found = .true.

    open (newunit=nfu, file=nlfnm, status='old', action='read')
    read (unit=nfu, nml=limits, iostat=ios)
    if (ios /= 0) then
      write (*,*) 'rdlim: could not read the namelist group LIMITS of '//trim(nlfnm)
      error stop '(rdlim)'
    end if
    ! the other groups keep their defaults when ABSENT (end of file); a group that is present and malformed stops the
    ! run as in the reference (phy/mod_vcoord.F90:829-838, phy/mod_diffusion.F90:236-247)
    rewind (nfu)
    read (unit=nfu, nml=vcoord, iostat=ios)
    if (ios > 0) then
      write (*,*) 'readnml_vcoord: No vertical coordinate variable group found in namelist. ', &
                  'could not read the namelist group VCOORD of '//trim(nlfnm)
      error stop '(readnml_vcoord)'
    end if
    rewind (nfu)
    read (unit=nfu, nml=diffusion, iostat=ios)
    if (ios > 0) then
      write (*,*) 'readnml_diffusion: could not read the namelist group DIFFUSION of '//trim(nlfnm)
      error stop '(readnml_diffusion)'
    end if
    close (nfu)

    write (*,*) 'rdlim: BLOM LIMITS NAMELIST GROUP (dynamical core):'
    write (*,*) 'EXPCNF ', trim(expcnf), '  BACLIN', baclin, '  BATROP', batrop, '  PREF', pref
    write (*,*) 'MOMMTH ', trim(mommth), '  PGFMTH ', trim(pgfmth), '  BMCMTH ', trim(bmcmth), '  ADVMTH ', trim(advmth)
    write (*,*) 'VCOORD_TYPE ', trim(vcoord_type), '  EITMTH ', trim(eitmth), '  LTEDTP ', trim(ltedtp)

    select case (trim(vcoord_type))
      case ('isopyc_bulkml')
        call gpu_set('vcoord_type', 'isopyc_bulkml')
      case ('cntiso_hybrid', 'plevel')
        ! the step of the other coordinates as far as the device library has it (DESIGN.md 3h): the group &ALE_REGRID_REMAP
        ! (phy/mod_ale_regrid_remap.F90:1185-1355) and the pressure levels (phy/mod_vcoord.F90:948-976)
        hybrid_coordinate = .true.
        open (newunit=nfu, file=nlfnm, status='old', action='read')
        read (unit=nfu, nml=ale_regrid_remap, iostat=ios)
        close (nfu)
        ! phy/mod_ale_regrid_remap.F90:1226-1232: a missing or unreadable group is not an error there
        if (ios /= 0) write (*,*) 'readnml_ale_regrid_remap: No vertical coordinate variable group found in namelist. Using defaults.'

        call gpu_set('vcoord_type', trim(vcoord_type))
        call gpu_set('ale_reconstruction_method', trim(reconstruction_method))
        call gpu_set('ale_density_limiting', trim(density_limiting))
        call gpu_set('ale_tracer_limiting', trim(tracer_limiting))
        call gpu_set('ale_velocity_limiting', trim(velocity_limiting))
        call gpu_set('ale_regrid_method', trim(regrid_method))
        call gpu_set('ale_upper_bndr_ord', upper_bndr_ord); call gpu_set('ale_lower_bndr_ord', lower_bndr_ord)
        call gpu_set('ale_k_range_plevel', k_range_plevel); call gpu_set('ale_dktzu', dktzu); call gpu_set('ale_dktzl', dktzl)
        call gpu_set('ale_density_pc_upper_bndr', merge(1, 0, density_pc_upper_bndr))
        call gpu_set('ale_density_pc_lower_bndr', merge(1, 0, density_pc_lower_bndr))
        call gpu_set('ale_tracer_pc_upper_bndr', merge(1, 0, tracer_pc_upper_bndr))
        call gpu_set('ale_tracer_pc_lower_bndr', merge(1, 0, tracer_pc_lower_bndr))
        call gpu_set('ale_velocity_pc_upper_bndr', merge(1, 0, velocity_pc_upper_bndr))
        call gpu_set('ale_velocity_pc_lower_bndr', merge(1, 0, velocity_pc_lower_bndr))
        call gpu_set('ale_dpmin_interior', dpmin_interior)
        call gpu_set('ale_regrid_nudge_ts', regrid_nudge_ts); call gpu_set('ale_stab_fac_limit', stab_fac_limit)
        call gpu_set('ale_dpvar_fac', dpvar_fac); call gpu_set('ale_smooth_diff_max', smooth_diff_max)
        select case (trim(plevel_spec))
          case ('inflation')
            dpmin = dpmin_surface*onem                   ! :906
            plevel(1) = 0.d0
            do k = 1, kdm-1
              plevel(k+1) = plevel(k)+dpmin
              dpmin = dpmin*dpmin_inflation_factor
            end do
          case ('namelist')
            k = 1
            do while (plevel(k) /= spval_)
              k = k+1
              if (k > size(plevel)) exit
            end do
            if (k /= kdm+1) then
              write (*,*) ' readnml_vcoord: number of plevel values does not match vertical dimension!'
              error stop '(readnml_vcoord)'
            end if
            plevel(1:kdm) = plevel(1:kdm)*onem
          case default
            write (*,*) ' readnml_vcoord: plevel_spec = ', trim(plevel_spec), ' is unsupported!'
            error stop '(readnml_vcoord)'
        end select
        call gpu_set_vector('plevel', plevel(1:kdm))
        ! mixed layer restratification of eddtra_ale (phy/mod_eddtra.F90:1773-1806 init_eddtra)
        call gpu_set('mlrmth', trim(mlrmth))
        call gpu_set('ce', ce); call gpu_set('tau_mlr', tau_mlr); call gpu_set('lfmin', lfmin)
        call gpu_set('tau_growing_hbl', tau_growing_hbl); call gpu_set('tau_decaying_hbl', tau_decaying_hbl)
        call gpu_set('tau_growing_hml', tau_growing_hml); call gpu_set('tau_decaying_hml', tau_decaying_hml)
        call gpu_set('mlbl_max_ratio', mlbl_max_ratio)
        call gpu_set('brine_mlbase_frac', brine_mlbase_frac)
      case default
        write (*,*) ' readnml_vcoord: vcoord_type = ', trim(vcoord_type), ' is unsupported!'
        error stop '(readnml_vcoord)'
    end select
    call gpu_set('expcnf', trim(expcnf))
    call gpu_set('pref', pref)
    call gpu_set('baclin', baclin)
    call gpu_set('batrop', batrop)
    lstep = 2*ceiling(.5d0*baclin/batrop)              ! phy/mod_time.F90:139-142
    call gpu_set('lstep', lstep)
    call gpu_set('dlt', baclin/lstep)
    call gpu_set('delt1', baclin)
    call gpu_set('mdv2hi', mdv2hi); call gpu_set('mdv2lo', mdv2lo); call gpu_set('mdv4hi', mdv4hi); call gpu_set('mdv4lo', mdv4lo)
    call gpu_set('mdc2hi', mdc2hi); call gpu_set('mdc2lo', mdc2lo); call gpu_set('vsc2hi', vsc2hi); call gpu_set('vsc2lo', vsc2lo)
    call gpu_set('vsc4hi', vsc4hi); call gpu_set('vsc4lo', vsc4lo); call gpu_set('cbar', cbar); call gpu_set('cb', cb)
    call gpu_set('cwbdts', cwbdts); call gpu_set('cwbdls', cwbdls)
    call gpu_set('mommth', trim(mommth))
    call gpu_set('pgfmth', trim(pgfmth))
    call gpu_set('bmcmth', trim(bmcmth))
    call gpu_set('advmth', trim(advmth))
    advect_cppm = trim(advmth) == 'cppm'
    call gpu_set('cppm_compatibility', trim(cppm_compatibility))
    call gpu_set('cppm_limiting', trim(cppm_limiting))
    if (cnsvdi) then
      call gpu_set('cnsvdi', 1)
    else
      call gpu_set('cnsvdi', 0)
    end if
    call gpu_set('eitmth', trim(eitmth))
    call gpu_set('bdmtyp', bdmtyp); call gpu_set('bdmc1', bdmc1); call gpu_set('bdmc2', bdmc2)
    call gpu_set('iwdflg', iwdflg); call gpu_set('iwdfac', iwdfac); call gpu_set('nubmin', nubmin)
    if (bdmldp) then
      call gpu_set('bdmldp', 1)
    else
      call gpu_set('bdmldp', 0)
    end if
    select case (trim(ltedtp))
      case ('layer');   call gpu_set('ltedtp_opt', 1)
      case ('neutral'); call gpu_set('ltedtp_opt', 2)
      case default
        write (*,*) ' ltedtp = ', trim(ltedtp), ' is unsupported!'
        error stop '(readnml_diffusion)'
    end select
    if (ndiff_surface_align) then
      call gpu_set('ndiff_surface_align', 1)
    else
      call gpu_set('ndiff_surface_align', 0)
    end if
    baclin_out = baclin
    ! an integer number of baroclinic steps per day, phy/mod_time.F90:121-130
    nstep_in_day = nint(86400.d0/baclin)
    if (abs(86400.d0/baclin - nstep_in_day) > epsilt) then
      write (*,*) 'init_timevars: must have an integer number of baroclinic time steps pr. day!'
      error stop '(init_timevars)'
    end if
    ! integration from time step nstep1 to nstep2; with csdiag, five steps (phy/mod_rdlim.F90:1149-1156)
    nstep1 = nday1*nstep_in_day
    nstep2 = nday2*nstep_in_day
    if (csdiag) then
      nstep2 = nstep1+5
      call gpu_set('csdiag', 1)
    end if
  end subroutine rdlim_gpu

end module mod_rdlim_gpu
