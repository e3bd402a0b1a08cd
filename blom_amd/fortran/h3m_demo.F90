! Stand-alone driver for mod_hor3map_gpu in the shape of the reference's call sites
! (phy/mod_ale_regrid_remap.F90:224-247, :1038-1046): analytic columns are reconstructed with
! BLOM's tracer settings and remapped; prints sums that tests/test_gpu_fortran_host.py compares
! with the Python-driven run of the same columns.
program h3m_demo
   use, intrinsic :: iso_fortran_env, only: real64
   use mod_hor3map_gpu
   implicit none
   integer, parameter :: r8 = real64, ncol = 96, ns = 20, nd = 15
   type(recon_grd_struct) :: rcgs
   type(recon_src_struct) :: trc_rcss
   type(remap_struct) :: rms
   real(r8) :: p_src(ns+1,ncol), trc(ns,ncol), p_dst(nd+1,ncol), trc_rm(nd,ncol), tpc(3,ns,ncol)
   integer :: i, k, errstat

   do i = 1, ncol
      p_src(1,i) = 0._r8
      do k = 1, ns
         if (mod(k + i, 5) == 0) then
            p_src(k+1,i) = p_src(k,i)                                     ! massless layer
         else
            p_src(k+1,i) = p_src(k,i) + 9806._r8*(1._r8 + 0.5_r8*sin(0.37_r8*k + 0.11_r8*i))
         endif
         trc(k,i) = 1._r8 + sin(0.5_r8*k + 0.05_r8*i)
      enddo
      do k = 1, nd + 1
         p_dst(k,i) = p_src(ns+1,i)*(real(k - 1, r8)/nd)**2
      enddo
      p_dst(nd+1,i) = p_src(ns+1,i)
   enddo

   rcgs%method = hor3map_ppm
   rcgs%left_bndr_ord = 6
   rcgs%right_bndr_ord = 4
   trc_rcss%limiting = hor3map_non_oscillatory_posdef
   trc_rcss%pc_left_bndr = .true.
   trc_rcss%pc_right_bndr = .false.

   errstat = prepare_reconstruction(rcgs, p_src)
   call check('prepare_reconstruction')
   errstat = reconstruct(rcgs, trc_rcss, trc)
   call check('reconstruct')
   errstat = extract_polycoeff(trc_rcss, tpc)
   call check('extract_polycoeff')
   errstat = prepare_remapping(rcgs, rms, p_dst)
   call check('prepare_remapping')
   errstat = remap(trc_rcss, rms, trc_rm)
   call check('remap')
   write (*, '(a,es24.16)') 'sum_polycoeff: ', sum(tpc)
   write (*, '(a,es24.16)') 'sum_remapped:  ', sum(trc_rm)

   ! error path: the reference's message for non-monotonic edges
   p_src(3,7) = p_src(2,7) - 1._r8
   errstat = prepare_reconstruction(rcgs, p_src)
   write (*, '(a,i0,2a)') 'errstat ', errstat, ': ', trim(hor3map_errstr(errstat))

   call free_rcss(trc_rcss)
   call free_rms(rms)
   call free_rcgs(rcgs)

contains
   subroutine check(what)
      character(len=*), intent(in) :: what
      if (errstat /= hor3map_noerr) then
         write (*, *) what, ': ', trim(hor3map_errstr(errstat))
         error stop 1
      endif
   end subroutine
end program h3m_demo
