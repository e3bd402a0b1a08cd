! ------------------------------------------------------------------------------
! program blom_dyncore -- standalone Fortran driver of the device-resident
! dynamical core, in the shape of the reference's `program blom`
! (drivers/nocoupler/blom.F:20-66): initialise, loop blom_step until the last step,
! print the final field checksum of dp (blom.F:56-57), write 'success' to run.status.
!
! Options: the reference's `limits` namelist file (mod_rdlim_gpu.F90).
! Initial state: a raw binary dump (blom_amd/statefile.py writes it from the numpy host
! initialisation); the reference reads netCDF here, which is out of scope.
!   record 1: idm jdm kdm ntr nreg nsteps  (int32 x6), baclin (real64)
!   then per entry: name (16 chars), kind (int32: 0 real field, 1 int field, 2 real option,
!                   3 int option, 4 string option), nlev (int32), payload
! ------------------------------------------------------------------------------
program blom_dyncore

  use mod_blomgpu
  use mod_rdlim_gpu, only: rdlim_gpu, hybrid_coordinate, advect_cppm
  implicit none

  character(len=256) :: fname
  character(len=16)  :: name
  character(len=32)  :: sval
  integer :: u, ios, kind, nlev, nsteps, nstep, nstep1, nstep2, i4(6), ival
  real(8) :: baclin, rval
  logical :: have_limits, full_physics = .false.
  real(8), allocatable :: buf(:,:,:)
  integer, allocatable :: ibuf(:,:,:)

  call get_command_argument(1, fname)
  if (len_trim(fname) == 0) fname = 'blom_state.bin'
  open (newunit=u, file=trim(fname), access='stream', form='unformatted', status='old', action='read')
  read (u) i4, baclin
  nsteps = i4(6)
  call gpu_init(i4(1), i4(2), i4(3), i4(4), i4(5), 0)
  do
    read (u, iostat=ios) name, kind, nlev
    if (ios /= 0) exit
    select case (kind)
      case (0)
        allocate (buf(idm+2*nbdy, jdm+2*nbdy, nlev))
        read (u) buf
        call gpu_upload(trim(name), buf, nlev)
        deallocate (buf)
      case (1)
        allocate (ibuf(idm+2*nbdy, jdm+2*nbdy, nlev))
        read (u) ibuf
        call gpu_upload_int(trim(name), ibuf, nlev)
        deallocate (ibuf)
      case (2)
        read (u) rval
        call gpu_set(trim(name), rval)
      case (3)
        read (u) ival
        call gpu_set(trim(name), ival)
        ! config 2's step as far as the device library has it (thermf, mxlayr, the front of difest_isobml, cmnfld2, cmnfld1):
        ! the host sequences the stages itself, so it needs to know
        if (trim(name) == 'full_physics') full_physics = ival /= 0
      case (4)
        read (u) sval
        call gpu_set(trim(name), trim(sval))
    end select
  end do
  close (u)
  ! options: the reference's namelist file `limits` / `ocn_in` in the working directory, when there is one, decides them
  ! (rdlim, phy/mod_rdlim.F90:137-175) and the length of the run; the option records of the state file are the fall-back
  nstep1 = 0
  nstep2 = nsteps
  call rdlim_gpu(have_limits, nstep1, nstep2, baclin)
  call get_command_argument(2, sval)               ! optional: stop after this many steps (short test runs)
  if (len_trim(sval) > 0) then
    read (sval, *) nsteps
    nstep2 = nstep1+nsteps
  end if

  ! the step counter starts at nstep1 = nday1*nstep_in_day (its parity decides the time levels and the order of cppm's sweeps)
  nstep = nstep1
  if (advect_cppm) call init_cppm()
  if (hybrid_coordinate) then
    ! blom_init's cmnfld1 (phy/mod_blom_init.F90): the mixed layer depth the first eddtra and ale_forcing read
    ! with blom_init's time level indices (phy/mod_blom_init.F90:256-261: m = mod(nstep1+1,2)+1, n = mod(nstep1,2)+1)
    call cmnfld1(mod(nstep+1,2)+1, mod(nstep,2)+1, mod(nstep+1,2)*kdm, mod(nstep,2)*kdm, 1+mod(nstep+1,2)*kdm, 1+mod(nstep,2)*kdm)
  end if
  do while (nstep < nstep2)
    call blom_step(nstep)
  end do

  call gpu_sync()
  call gpu_chksum('dp', 2*kdm, 1, 'dp')          ! blom.F:56-57
  call gpu_chksum('temp', 2*kdm, 1, 'temp')
  call gpu_chksum('u', 2*kdm, 13, 'u')
  if (difest_estimates) call gpu_chksum('difint', kdm, 1, 'difint')     ! (the estimates ran: see mod_blomgpu)
  open (newunit=u, file='run.status', status='unknown')
  write (u,*) 'success'                          ! blom.F:59-61
  close (u)
  call gpu_finalize()

contains

  subroutine blom_step(nstep)
    ! stage sequence of phy/mod_blom_step.F90:89-253, dynamical core only
    integer, intent(inout) :: nstep
    integer :: m, n, mm, nn, k1m, k1n
    m = mod(nstep  ,2)+1
    n = mod(nstep+1,2)+1
    mm = (m-1)*kdm
    nn = (n-1)*kdm
    k1m = 1+mm
    k1n = 1+nn
    nstep = nstep+1                                ! step_time
    call gpu_set('nstep', nstep)
    if (hybrid_coordinate) then
      ! vcoord_type = 'cntiso_hybrid' or 'plevel', phy/mod_blom_step.F90:126-233 as far as the device library has it
      ! (DESIGN.md 3h; difest_*_hybrid and thermf are left out: what they produce stays as uploaded)
      call init_fluxes(m,n,mm,nn,k1m,k1n)
      call tmsmt1(nn)
      call ale_regrid_remap(m,n,mm,nn,k1m,k1n)
      call cmnfld2(m,n,mm,nn,k1m,k1n)
      call stage6('halo_difest_hyb',m,n,mm,nn,k1m,k1n)
      call eddtra(m,n,mm,nn,k1m,k1n)
      call advect(m,n,mm,nn,k1m,k1n)
      call pbcor1(m,n,mm,nn,k1m,k1n)
      call diffus(m,n,mm,nn,k1m,k1n)
      call sfcstr(m,n,mm,nn,k1m,k1n)
      call pgforc(m,n,mm,nn,k1m,k1n)
      call momtum(m,n,mm,nn,k1m,k1n)
      call cmnfld_bfsqi_ale(m,n,mm,nn,k1m,k1n)
      call ale_forcing(m,n,mm,nn,k1m,k1n)
      call stage6('halo_difest_vert',m,n,mm,nn,k1m,k1n)
      call ale_vdifft(m,n,mm,nn,k1m,k1n)
      call ale_vdiffm(m,n,mm,nn,k1m,k1n)
      call updtrc(m,n,mm,nn,k1m,k1n)
      call barotp(m,n,mm,nn,k1m,k1n)
      call pbcor2(m,n,mm,nn,k1m,k1n)
      call tmsmt2(m,mm,nn,k1m)
      call cmnfld1(m,n,mm,nn,k1m,k1n)
      call gpu_set('delt1', baclin+baclin)
      return
    end if
    if (full_physics) then
      ! phy/mod_blom_step.F90:96-253 for isopyc_bulkml, every stage the device library has (DESIGN.md 3i); of difest_isobml the
      ! part in front of the diffusivity estimates
      call init_fluxes(m,n,mm,nn,k1m,k1n)
      call tmsmt1(nn)
      call cmnfld2(m,n,mm,nn,k1m,k1n)
      call difest_isobml(m,n,mm,nn,k1m,k1n)
      call eddtra(m,n,mm,nn,k1m,k1n)
      call advect(m,n,mm,nn,k1m,k1n)
      call pbcor1(m,n,mm,nn,k1m,k1n)
      call diffus(m,n,mm,nn,k1m,k1n)
      call sfcstr(m,n,mm,nn,k1m,k1n)
      call pgforc(m,n,mm,nn,k1m,k1n)
      call momtum(m,n,mm,nn,k1m,k1n)
      call convec(m,n,mm,nn,k1m,k1n)
      call diapfl(n,nn,k1n)
      call thermf(m,n,mm,nn,k1m,k1n)
      call mxlayr(m,n,mm,nn,k1m,k1n)
      call updtrc(m,n,mm,nn,k1m,k1n)
      call barotp(m,n,mm,nn,k1m,k1n)
      call pbcor2(m,n,mm,nn,k1m,k1n)
      call tmsmt2(m,mm,nn,k1m)
      call cmnfld1(m,n,mm,nn,k1m,k1n)
      call gpu_set('delt1', baclin+baclin)
      return
    end if
    call init_fluxes(m,n,mm,nn,k1m,k1n)
    call tmsmt1(nn)
    call halo_cmnfld2(n)
    call halo_difest(nn)
    call eddtra(m,n,mm,nn,k1m,k1n)
    call advect(m,n,mm,nn,k1m,k1n)
    call pbcor1(m,n,mm,nn,k1m,k1n)
    call diffus(m,n,mm,nn,k1m,k1n)
    call sfcstr(m,n,mm,nn,k1m,k1n)
    call pgforc(m,n,mm,nn,k1m,k1n)
    call momtum(m,n,mm,nn,k1m,k1n)
    call convec(m,n,mm,nn,k1m,k1n)
    call diapfl(n,nn,k1n)
    call mxlayr_tail(nn,k1n)
    call updtrc(m,n,mm,nn,k1m,k1n)
    call barotp(m,n,mm,nn,k1m,k1n)
    call pbcor2(m,n,mm,nn,k1m,k1n)
    call tmsmt2(m,mm,nn,k1m)
    call gpu_set('delt1', baclin+baclin)           ! phy/mod_blom_step.F90:300
  end subroutine blom_step

end program blom_dyncore
