! ------------------------------------------------------------------------------
! mod_blomgpu -- Fortran host side of the MI355X-native BLOM dynamical core.
!
! Thin ISO_C_BINDING layer over include/blomgpu.h.  The stage wrappers keep the
! reference's names and argument lists (phy/mod_blom_step.F90:126-227):
!     call advect(m,n,mm,nn,k1m,k1n)      call diapfl(n,nn,k1n)
!     call tmsmt1(nn)                     call tmsmt2(m,mm,nn,k1m)  ...
! so that the stage sequence of blom_step reads exactly as in the reference.  The
! reference has no return codes (it prints and calls xchalt/xcstop, e.g.
! phy/mod_advect.F90:166-172); a non-zero C status is converted to that behaviour
! in gpu_check.
! ------------------------------------------------------------------------------
module mod_blomgpu

  use iso_c_binding
  implicit none
  private
  ! difest_isobml below runs the whole routine when set: public, and also set by gpu_set('difest_live', 1) -- the option that makes
  ! blomgpu_step estimate the diffusivities -- so that a host sequencing the stages itself and the device-resident loop agree
  logical, save, public :: difest_estimates = .false.

  integer, parameter, public :: nbdy = 4          ! phy/mod_xc.F90:45
  type(c_ptr), save :: ctx = c_null_ptr
  integer, public :: idm, jdm, kdm, ntr, nreg

  type, bind(C) :: blomgpu_dims
    integer(c_int) :: idm, jdm, kdm, nbdy, itdm, jtdm, i0, j0, nreg, ntr, device
  end type

  interface
    integer(c_int) function blomgpu_create(dims, ctxo) bind(C, name='blomgpu_create')
      import :: blomgpu_dims, c_ptr, c_int
      type(blomgpu_dims), intent(in) :: dims
      type(c_ptr), intent(out) :: ctxo
    end function
    integer(c_int) function blomgpu_destroy(c) bind(C, name='blomgpu_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: c
    end function
    type(c_ptr) function blomgpu_last_error(c) bind(C, name='blomgpu_last_error')
      import :: c_ptr
      type(c_ptr), value :: c
    end function
    integer(c_int) function blomgpu_set_real(c, name, v) bind(C, name='blomgpu_set_real')
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      real(c_double), value :: v
    end function
    integer(c_int) function blomgpu_set_int(c, name, v) bind(C, name='blomgpu_set_int')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: v
    end function
    integer(c_int) function blomgpu_set_str(c, name, v) bind(C, name='blomgpu_set_str')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*), v(*)
    end function
    integer(c_int) function blomgpu_upload(c, name, host, nlev) bind(C, name='blomgpu_upload')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      type(*), intent(in) :: host(*)
      integer(c_int), value :: nlev
    end function
    integer(c_int) function blomgpu_download(c, name, host, nlev) bind(C, name='blomgpu_download')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      type(*) :: host(*)
      integer(c_int), value :: nlev
    end function
    integer(c_int) function blomgpu_field_info(c, name, nlev, isint) bind(C, name='blomgpu_field_info')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: nlev, isint
    end function
    integer(c_int) function blomgpu_stage(c, stage, m, n, mm, nn, k1m, k1n) bind(C, name='blomgpu_stage')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: stage(*)
      integer(c_int), value :: m, n, mm, nn, k1m, k1n
    end function
    integer(c_int) function blomgpu_xctilr(c, name, lev0, l1, ld, mh, nh, itype) bind(C, name='blomgpu_xctilr')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: lev0, l1, ld, mh, nh, itype
    end function
    integer(c_int) function blomgpu_crc(c, name, lev0, nlev, itype, crc) bind(C, name='blomgpu_crc')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: lev0, nlev, itype
      integer(c_int), intent(out) :: crc
    end function
    integer(c_int) function blomgpu_xcsum(c, name, lev, itype, s) bind(C, name='blomgpu_xcsum')
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: lev, itype
      real(c_double), intent(out) :: s
    end function
    integer(c_int) function blomgpu_budget_sums(c, ncall, n, nn) bind(C, name='blomgpu_budget_sums')
      import :: c_ptr, c_int
      type(c_ptr), value :: c
      integer(c_int), value :: ncall, n, nn
    end function
    integer(c_int) function blomgpu_budget_get(c, which, ncall, n, v) bind(C, name='blomgpu_budget_get')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: c
      integer(c_int), value :: which, ncall, n
      real(c_double), intent(out) :: v
    end function
    integer(c_int) function blomgpu_sync(c) bind(C, name='blomgpu_sync')
      import :: c_ptr, c_int
      type(c_ptr), value :: c
    end function
    integer(c_int) function blomgpu_set_vector(c, name, v, nv) bind(C, name='blomgpu_set_vector')
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: c
      character(kind=c_char), intent(in) :: name(*)
      real(c_double), intent(in) :: v(*)
      integer(c_int), value :: nv
    end function
  end interface

  public :: gpu_init, gpu_finalize, gpu_set, gpu_upload, gpu_upload_int, gpu_download, gpu_nlev, &
            gpu_halo, gpu_chksum, gpu_sync, gpu_xcsum, budget_sums, gpu_budget
  public :: stage6
  public :: init_fluxes, tmsmt1, tmsmt2, advect, pbcor1, pbcor2, diffus, pgforc, momtum, &
            diapfl, barotp, eddtra, convec, sfcstr, updtrc, init_cppm, halo_cmnfld2, halo_difest, mxlayr_tail, &
            cmnfld1, cmnfld2, ale_regrid_remap, ale_vdifft, ale_vdiffm, ale_forcing, &
            cmnfld_bfsqi_ale, gpu_set_vector, difest_isobml, thermf, mxlayr

  interface gpu_set
    module procedure gpu_set_real, gpu_set_int, gpu_set_str
  end interface

contains

  function cz(s) result(z)
    character(len=*), intent(in) :: s
    character(kind=c_char, len=len_trim(s)+1) :: z
    z = trim(s)//c_null_char
  end function

  subroutine gpu_check(rc, where)
    ! the reference prints to lp and calls xchalt + stop '(name)' (phy/mod_xc.F90:516)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    p = blomgpu_last_error(ctx)
    call c_f_pointer(p, msg, [512])
    i = 1
    do while (i < 512 .and. msg(i) /= c_null_char)
      i = i+1
    end do
    write (*,*) '**************************************************'
    write (*,*) msg(1:i-1)
    write (*,*) '**************************************************'
    error stop '('//where//')'
  end subroutine

  subroutine gpu_init(idm_, jdm_, kdm_, ntr_, nreg_, device)
    integer, intent(in) :: idm_, jdm_, kdm_, ntr_, nreg_, device
    type(blomgpu_dims) :: d
    idm = idm_; jdm = jdm_; kdm = kdm_; ntr = ntr_; nreg = nreg_
    d = blomgpu_dims(idm, jdm, kdm, nbdy, idm, jdm, 0, 0, nreg, ntr, device)
    call gpu_check(blomgpu_create(d, ctx), 'gpu_init')
  end subroutine

  subroutine gpu_finalize()
    integer(c_int) :: rc
    rc = blomgpu_destroy(ctx)
    ctx = c_null_ptr
  end subroutine

  subroutine gpu_set_real(name, v)
    character(len=*), intent(in) :: name
    real(8), intent(in) :: v
    call gpu_check(blomgpu_set_real(ctx, cz(name), v), 'gpu_set')
  end subroutine
  subroutine gpu_set_int(name, v)
    character(len=*), intent(in) :: name
    integer, intent(in) :: v
    call gpu_check(blomgpu_set_int(ctx, cz(name), v), 'gpu_set')
    if (name == 'difest_live') difest_estimates = v /= 0
  end subroutine
  subroutine gpu_set_str(name, v)
    character(len=*), intent(in) :: name, v
    call gpu_check(blomgpu_set_str(ctx, cz(name), cz(v)), 'gpu_set')
  end subroutine

  integer function gpu_nlev(name, isint)
    character(len=*), intent(in) :: name
    logical, intent(out) :: isint
    integer(c_int) :: nl, ii_
    call gpu_check(blomgpu_field_info(ctx, cz(name), nl, ii_), 'gpu_nlev')
    gpu_nlev = nl
    isint = ii_ /= 0
  end function

  ! a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nlev): the reference's module-array layout
  subroutine gpu_upload(name, a, nlev)
    character(len=*), intent(in) :: name
    integer, intent(in) :: nlev
    real(8), intent(in) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nlev)
    call gpu_check(blomgpu_upload(ctx, cz(name), a, nlev), 'gpu_upload')
  end subroutine
  subroutine gpu_upload_int(name, a, nlev)
    character(len=*), intent(in) :: name
    integer, intent(in) :: nlev
    integer, intent(in) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nlev)
    call gpu_check(blomgpu_upload(ctx, cz(name), a, nlev), 'gpu_upload_int')
  end subroutine
  subroutine gpu_download(name, a, nlev)
    character(len=*), intent(in) :: name
    integer, intent(in) :: nlev
    real(8), intent(out) :: a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,nlev)
    call gpu_check(blomgpu_download(ctx, cz(name), a, nlev), 'gpu_download')
  end subroutine

  ! xctilr(a(1-nbdy,1-nbdy,lev0), l1, ld, mh, nh, itype), phy/mod_xc.F90:2342
  subroutine gpu_halo(name, lev0, l1, ld, mh, nh, itype)
    character(len=*), intent(in) :: name
    integer, intent(in) :: lev0, l1, ld, mh, nh, itype
    call gpu_check(blomgpu_xctilr(ctx, cz(name), lev0, l1, ld, mh, nh, itype), 'xctilr')
  end subroutine

  ! chksum(a, kcsd, itype, text), phy/mod_checksum.F90:41-74
  subroutine gpu_chksum(name, kcsd, itype, text)
    character(len=*), intent(in) :: name, text
    integer, intent(in) :: kcsd, itype
    integer(c_int) :: crc
    call gpu_check(blomgpu_crc(ctx, cz(name), 1, kcsd, itype, crc), 'chksum')
    write (*,'(3a,z8.8)') ' chksum: ', trim(text), ': 0x', crc
  end subroutine

  subroutine gpu_xcsum(s, name, lev, itype)    ! phy/mod_xc.F90:4116 xcsum(sum, a, mask) on a device field
    real(c_double), intent(out) :: s
    character(len=*), intent(in) :: name
    integer, intent(in) :: lev, itype
    call gpu_check(blomgpu_xcsum(ctx, cz(name), lev, itype, s), 'xcsum')
  end subroutine
  subroutine budget_sums(ncall, n, nn)         ! phy/mod_budget.F90:95
    integer, intent(in) :: ncall, n, nn
    call gpu_check(blomgpu_budget_sums(ctx, ncall, n, nn), 'budget_sums')
  end subroutine
  function gpu_budget(which, ncall, n) result(v)   ! sdp/tdp/trdp(ncall,n), phy/mod_budget.F90:50-59
    integer, intent(in) :: which, ncall, n
    real(c_double) :: v
    call gpu_check(blomgpu_budget_get(ctx, which, ncall, n, v), 'budget_get')
  end function

  subroutine gpu_sync()
    call gpu_check(blomgpu_sync(ctx), 'gpu_sync')
  end subroutine

  subroutine stage6(name, m, n, mm, nn, k1m, k1n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: m, n, mm, nn, k1m, k1n
    call gpu_check(blomgpu_stage(ctx, cz(name), m, n, mm, nn, k1m, k1n), name)
  end subroutine

  ! ---- stage API of the reference ------------------------------------------------------------
  subroutine init_fluxes(m,n,mm,nn,k1m,k1n)   ! phy/mod_state.F90:341
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('init_fluxes',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine tmsmt1(nn)                        ! phy/mod_tmsmt.F90:209
    integer, intent(in) :: nn
    call stage6('tmsmt1',0,0,0,nn,0,0)
  end subroutine
  subroutine tmsmt2(m,mm,nn,k1m)               ! phy/mod_tmsmt.F90:281
    integer, intent(in) :: m,mm,nn,k1m
    call stage6('tmsmt2',m,0,mm,nn,k1m,0)
  end subroutine
  subroutine advect(m,n,mm,nn,k1m,k1n)         ! phy/mod_advect.F90:59
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('advect',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine pbcor1(m,n,mm,nn,k1m,k1n)         ! phy/mod_pbcor.F90:66
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('pbcor1',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine pbcor2(m,n,mm,nn,k1m,k1n)         ! phy/mod_pbcor.F90:416
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('pbcor2',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine diffus(m,n,mm,nn,k1m,k1n)         ! phy/mod_diffus.F90:41
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('diffus',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine pgforc(m,n,mm,nn,k1m,k1n)         ! phy/mod_pgforc.F90:438
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('pgforc',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine momtum(m,n,mm,nn,k1m,k1n)         ! phy/mod_momtum.F90:215
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('momtum',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine diapfl(n,nn,k1n)                  ! phy/mod_diapfl.F90:49
    integer, intent(in) :: n,nn,k1n
    call stage6('diapfl',0,n,0,nn,0,k1n)
  end subroutine
  subroutine barotp(m,n,mm,nn,k1m,k1n)         ! phy/mod_barotp.F90:148
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('barotp',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine updtrc(m,n,mm,nn,k1m,k1n)         ! trc/mod_tracers_update.F90:152
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('updtrc',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine sfcstr(m,n,mm,nn,k1m,k1n)         ! phy/mod_sfcstr.F90:33
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('sfcstr',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine convec(m,n,mm,nn,k1m,k1n)         ! phy/mod_convec.F90:43
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('convec',m,n,mm,nn,k1m,k1n)
  end subroutine

  subroutine eddtra(m,n,mm,nn,k1m,k1n)         ! phy/mod_eddtra.F90:148
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('eddtra',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine init_cppm()                       ! phy/mod_cppm.F90:2504
    call stage6('init_cppm',0,0,0,0,0,0)
  end subroutine
  subroutine halo_cmnfld2(n)                   ! phy/mod_cmnfld_routines.F90:1171-1196
    integer, intent(in) :: n
    call stage6('halo_cmnfld2',0,n,0,0,0,0)
  end subroutine
  subroutine cmnfld1(m,n,mm,nn,k1m,k1n)         ! phy/mod_cmnfld_routines.F90:1090 (isopyc_bulkml: cmnfld_z)
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('cmnfld1',m,n,mm,nn,k1m,k1n)
  end subroutine cmnfld1
  subroutine cmnfld2(m,n,mm,nn,k1m,k1n)         ! phy/mod_cmnfld_routines.F90:1158 (halo updates, bfsqf, neutral slopes)
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('cmnfld2',m,n,mm,nn,k1m,k1n)
  end subroutine cmnfld2
  subroutine halo_difest(nn)                   ! phy/mod_difest.F90:750-772
    integer, intent(in) :: nn
    call stage6('halo_difest',0,0,0,nn,0,0)
  end subroutine
  subroutine difest_isobml(m,n,mm,nn,k1m,k1n)   ! phy/mod_difest.F90:735
    ! difest_estimates = .true.: the whole routine (difest_common_iso, difest_vertical_iso, difest_lateral_iso: difint, difiso, difdia,
    ! difwgt estimated every step) -- the host must have uploaded what they read beside the state: plat, cosang, sinang, twedon, ficem,
    ! buoyfl, the planes tdmls and bdmlq its own libm fills (include/blomgpu.h) and &DIFFUSION's variables through gpu_set;
    ! .false. (default): the part in front of the estimates, :750-790
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    if (difest_estimates) then
      call stage6('difest_isobml',m,n,mm,nn,k1m,k1n)
    else
      call stage6('difest_isobml_pre',m,n,mm,nn,k1m,k1n)
    end if
  end subroutine
  subroutine thermf(m,n,mm,nn,k1m,k1n)          ! phy/mod_thermf.F90:35
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('thermf',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine mxlayr(m,n,mm,nn,k1m,k1n)          ! phy/mod_mxlayr.F90:130
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('mxlayr',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine mxlayr_tail(nn,k1n)               ! phy/mod_mxlayr.F90:1266-1310
    integer, intent(in) :: nn,k1n
    call stage6('mxlayr_tail',0,0,0,nn,0,k1n)
  end subroutine
  subroutine ale_regrid_remap(m,n,mm,nn,k1m,k1n)   ! phy/mod_ale_regrid_remap.F90:1486 (vcoord 'plevel', 'cntiso_hybrid'/'direct')
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('ale_regrid_remap',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine ale_vdifft(m,n,mm,nn,k1m,k1n)          ! phy/mod_ale_vdiff.F90:50
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('ale_vdifft',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine ale_vdiffm(m,n,mm,nn,k1m,k1n)          ! phy/mod_ale_vdiff.F90:245
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('ale_vdiffm',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine cmnfld_bfsqi_ale(m,n,mm,nn,k1m,k1n)    ! phy/mod_cmnfld_routines.F90:352
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('cmnfld_bfsqi_ale',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine ale_forcing(m,n,mm,nn,k1m,k1n)         ! phy/mod_ale_forcing.F90:45
    integer, intent(in) :: m,n,mm,nn,k1m,k1n
    call stage6('ale_forcing',m,n,mm,nn,k1m,k1n)
  end subroutine
  subroutine gpu_set_vector(name, v)           ! 1-D module arrays: 'plevel' (phy/mod_vcoord.F90:99)
    character(len=*), intent(in) :: name
    real(c_double), intent(in) :: v(:)
    call gpu_check(blomgpu_set_vector(ctx, cz(name), v, size(v)), name)
  end subroutine

end module mod_blomgpu
