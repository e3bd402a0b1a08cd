! TEST INFRASTRUCTURE (oracle).  C-callable batch driver around the REFERENCE's own
! phy/mod_hor3map.F90 (compiled from where it lies under /root/reference by oracle/Makefile,
! target hor3map).  One call runs the reference's public sequence
!   prepare_reconstruction -> reconstruct -> extract_polycoeff -> regrid -> prepare_remapping -> remap
! for every column of a slab, with a recon_grd_struct that has one slot per column (as BLOM
! dimensions it, phy/mod_ale_regrid_remap.F90:1404-1438), and records each call's errstat.
module ref_hor3map
   use, intrinsic :: iso_c_binding
   use mod_hor3map
   implicit none
   ! module storage: the reference's pointer components have no default initialisation and rely on
   ! static (zeroed) storage, as BLOM's own module-level instances have
   type(recon_grd_struct), target, save :: rcgs
   type(recon_src_struct), target, save :: rcss
   type(remap_struct), target, save :: rms
contains

   subroutine ref_h3m_run(method, lb_ord, rb_ord, limiting, pc_l, pc_r, ncol, n_src, n_dst, n_grd, &
                          regrid_method, x_src, u_src, x_dst, u_grd, missing, polycoeff, u_dst, x_grd, &
                          errs, n_act, m_act) bind(C, name='ref_h3m_run')
      integer(c_int), value :: method, lb_ord, rb_ord, limiting, pc_l, pc_r, ncol, n_src, n_dst, n_grd, &
                               regrid_method
      real(c_double), intent(in) :: x_src(n_src+1,ncol), u_src(n_src,ncol), x_dst(n_dst+1,ncol), &
                                    u_grd(n_grd,ncol)
      real(c_double), value :: missing
      real(c_double), intent(inout) :: polycoeff(*), u_dst(n_dst,ncol), x_grd(n_grd,ncol)
      integer(c_int), intent(inout) :: errs(6,ncol), n_act(ncol), m_act(ncol)

      integer :: i, np, off
      real(c_double), allocatable :: pc(:,:)

      rcgs%i_ubound = ncol
      rcgs%method = method
      rcgs%left_bndr_ord = lb_ord
      rcgs%right_bndr_ord = rb_ord
      rcss%limiting = limiting
      rcss%pc_left_bndr = pc_l /= 0
      rcss%pc_right_bndr = pc_r /= 0
      select case (method)
         case (hor3map_pcm); np = 1
         case (hor3map_plm); np = 2
         case (hor3map_ppm); np = 3
         case default;       np = 5
      end select
      allocate(pc(np,n_src))

      do i = 1, ncol
         errs(1,i) = prepare_reconstruction(rcgs, x_src(:,i), i, 1)
         errs(2,i) = reconstruct(rcgs, rcss, u_src(:,i), i, 1)
         errs(3,i) = extract_polycoeff(rcss, pc, i, 1)
         if (errs(3,i) == hor3map_noerr) then
            off = (i - 1)*np*n_src
            polycoeff(off+1:off+np*n_src) = reshape(pc, [np*n_src])
         endif
         errs(4,i) = regrid(rcss, u_grd(:,i), x_grd(:,i), missing, i, 1, regrid_method)
         errs(5,i) = prepare_remapping(rcgs, rms, x_dst(:,i), i, 1)
         errs(6,i) = remap(rcss, rms, u_dst(:,i), i, 1)
         if (errs(1,i) == hor3map_noerr) then
            n_act(i) = rcgs%n_src_actual
            m_act(i) = rcgs%method_actual
         endif
      enddo

      if (rcgs%initialized) call free_rcgs(rcgs)
   end subroutine ref_h3m_run

end module ref_hor3map
