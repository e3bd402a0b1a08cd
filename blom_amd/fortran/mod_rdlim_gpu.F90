! ------------------------------------------------------------------------------
! module mod_rdlim_gpu -- the namelist input of the reference for the host of the device-resident dynamical core.
! Reads the file `limits` (or `ocn_in`) exactly as the reference's readers do -- groups &LIMITS
! (phy/mod_rdlim.F90:137-175), &VCOORD (phy/mod_vcoord.F90:818-823) and &DIFFUSION (phy/mod_diffusion.F90:214-218),
! each declared with the reference's full variable list so that its input files are accepted unchanged -- and hands
! the variables the dynamical core uses to the device library (blomgpu_set_*).  Everything else in the groups is
! read and ignored here: it belongs to parts of the model outside this path (I/O, forcing, restart, diagnostics).
! ------------------------------------------------------------------------------
module mod_rdlim_gpu

  use mod_blomgpu
  implicit none
  private
  public :: rdlim_gpu

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
               cppm_compatibility = 'full', cppm_limiting = 'non_oscillatory', mldmth = 'lev82', mlrmth = 'none', &
               mlrttp = 'constant', swamth = 'jerlov', chlopt = 'climatology', wavsrc = 'none'
    logical :: woa_nuopc_provided = .false., aptflx = .false., apsflx = .false., ditflx = .false., disflx = .false., &
               srxbal = .false., smtfrc = .false., sprfac = .false., cnsvdi = .false., csdiag = .false., &
               use_stream_relaxation = .false., use_stream_dust = .false., use_diag = .false.
    real(8) :: pref = 2000.d4, baclin = 0.d0, batrop = 0.d0, mdv2hi = 0.d0, mdv2lo = 0.d0, mdv4hi = 0.d0, mdv4lo = 0.d0, &
               mdc2hi = 0.d0, mdc2lo = 0.d0, vsc2hi = 0.d0, vsc2lo = 0.d0, vsc4hi = 0.d0, vsc4lo = 0.d0, cbar = 0.d0, &
               cb = 0.d0, cwbdts = 0.d0, cwbdls = 0.d0, ce = 0.d0, cl = 0.d0, tau_mlr = 0.d0, tau_growing_hbl = 0.d0, &
               tau_decaying_hbl = 0.d0, tau_growing_hml = 0.d0, tau_decaying_hml = 0.d0, lfmin = 0.d0, mstar = 0.d0, &
               nstar = 0.d0, wpup_min = 0.d0, mlbl_max_ratio = 0.d0, rm0 = 0.d0, rm5 = 0.d0, niwgf = 0.d0, niwbf = 0.d0, &
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

    found = .false.
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

    if (trim(vcoord_type) /= 'isopyc_bulkml') then
      write (*,*) ' vcoord_type = ', trim(vcoord_type), ' is not built on the device (ALE stack)'
      error stop '(readnml_vcoord)'
    end if
    call gpu_set('vcoord_tag', 1)
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
