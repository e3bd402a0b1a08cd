! ------------------------------------------------------------------------------
! mod_hor3map_gpu -- Fortran host side of the batched HOR3MAP library on MI355X.
!
! ISO_C_BINDING layer over include/blomgpu_hor3map.h that keeps the public names
! of the reference's phy/mod_hor3map.F90 (:268-277): the three derived types
! (with the same user-settable components: n_src/method/left_bndr_ord/
! right_bndr_ord, limiting/pc_left_bndr/pc_right_bndr, n_dst), the option and
! error parameters, and the procedures
!    prepare_reconstruction, reconstruct, extract_polycoeff, regrid,
!    prepare_remapping, remap, free_rcgs, free_rcss, free_rms, hor3map_errstr.
! The one difference is the batching: where the reference is called with ONE
! column and optional (i_index, j_index), these take the whole slab -- arrays
! carry a trailing column dimension, e.g. x_edge_src(n_src+1, ncol) -- and
! return the errstat of the first failing column (column order).
! ------------------------------------------------------------------------------
module mod_hor3map_gpu

   use, intrinsic :: iso_c_binding
   use, intrinsic :: iso_fortran_env, only: real64
   implicit none
   private

   integer, parameter :: r8 = real64

   integer, parameter, public :: &
      hor3map_pcm = 100, hor3map_plm = 101, hor3map_ppm = 102, hor3map_pqm = 103, &
      hor3map_no_limiting = 200, hor3map_monotonic = 201, hor3map_non_oscillatory = 203, &
      hor3map_non_oscillatory_posdef = 204, &
      hor3map_regrid_method_1 = 301, hor3map_regrid_method_2 = 302, &
      hor3map_noerr = 0

   type, public :: recon_grd_struct
      integer :: ncol = 1                    ! columns of the slab (the reference's i/j index range)
      integer :: method = hor3map_ppm, left_bndr_ord = 0, right_bndr_ord = 0
      integer :: device = 0
      integer :: n_src = 0
      logical :: initialized = .false.
      type(c_ptr) :: h = c_null_ptr
   end type
   type, public :: recon_src_struct
      integer :: limiting = hor3map_monotonic
      logical :: pc_left_bndr = .true., pc_right_bndr = .true.
      logical :: initialized = .false.
      type(c_ptr) :: h = c_null_ptr
      type(c_ptr) :: grid = c_null_ptr
   end type
   type, public :: remap_struct
      integer :: n_dst = 0
      logical :: initialized = .false.
      type(c_ptr) :: h = c_null_ptr
      type(c_ptr) :: grid = c_null_ptr
   end type

   interface
      integer(c_int) function blomgpu_h3m_grid_create(h, device, ncol, n_src, method, lbo, rbo) &
         bind(C, name='blomgpu_h3m_grid_create')
         import :: c_ptr, c_int
         type(c_ptr), intent(out) :: h
         integer(c_int), value :: device, ncol, n_src, method, lbo, rbo
      end function
      integer(c_int) function blomgpu_h3m_src_create(g, h, limiting, pcl, pcr) &
         bind(C, name='blomgpu_h3m_src_create')
         import :: c_ptr, c_int
         type(c_ptr), value :: g
         type(c_ptr), intent(out) :: h
         integer(c_int), value :: limiting, pcl, pcr
      end function
      integer(c_int) function blomgpu_h3m_map_create(g, h, n_dst) bind(C, name='blomgpu_h3m_map_create')
         import :: c_ptr, c_int
         type(c_ptr), value :: g
         type(c_ptr), intent(out) :: h
         integer(c_int), value :: n_dst
      end function
      subroutine blomgpu_h3m_src_free(h) bind(C, name='blomgpu_h3m_src_free')
         import :: c_ptr
         type(c_ptr), value :: h
      end subroutine
      subroutine blomgpu_h3m_map_free(h) bind(C, name='blomgpu_h3m_map_free')
         import :: c_ptr
         type(c_ptr), value :: h
      end subroutine
      subroutine blomgpu_h3m_grid_free(h) bind(C, name='blomgpu_h3m_grid_free')
         import :: c_ptr
         type(c_ptr), value :: h
      end subroutine
      integer(c_int) function blomgpu_h3m_prepare_reconstruction(g, x) &
         bind(C, name='blomgpu_h3m_prepare_reconstruction')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: g
         real(c_double), intent(in) :: x(*)
      end function
      integer(c_int) function blomgpu_h3m_reconstruct(g, s, u) bind(C, name='blomgpu_h3m_reconstruct')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: g, s
         real(c_double), intent(in) :: u(*)
      end function
      integer(c_int) function blomgpu_h3m_extract_polycoeff(s, pc) bind(C, name='blomgpu_h3m_extract_polycoeff')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: s
         real(c_double), intent(inout) :: pc(*)
      end function
      integer(c_int) function blomgpu_h3m_regrid(s, ng, u, x, missing, method) bind(C, name='blomgpu_h3m_regrid')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: s
         integer(c_int), value :: ng, method
         real(c_double), intent(in) :: u(*)
         real(c_double), intent(inout) :: x(*)
         real(c_double), value :: missing
      end function
      integer(c_int) function blomgpu_h3m_prepare_remapping(g, m, x) bind(C, name='blomgpu_h3m_prepare_remapping')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: g, m
         real(c_double), intent(in) :: x(*)
      end function
      integer(c_int) function blomgpu_h3m_remap(s, m, u) bind(C, name='blomgpu_h3m_remap')
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: s, m
         real(c_double), intent(inout) :: u(*)
      end function
      type(c_ptr) function blomgpu_h3m_errstr(e) bind(C, name='blomgpu_h3m_errstr')
         import :: c_ptr, c_int
         integer(c_int), value :: e
      end function
   end interface

   public :: prepare_reconstruction, prepare_remapping, reconstruct, extract_polycoeff, regrid, remap, &
             free_rcgs, free_rcss, free_rms, hor3map_errstr

contains

   function prepare_reconstruction(rcgs, x_edge_src) result(errstat)     ! mod_hor3map.F90:3834
      type(recon_grd_struct), intent(inout) :: rcgs
      real(r8), dimension(:,:), intent(in) :: x_edge_src                ! (n_src+1, ncol)
      integer :: errstat
      if (.not. rcgs%initialized) then
         rcgs%n_src = size(x_edge_src, 1) - 1
         rcgs%ncol = size(x_edge_src, 2)
         errstat = blomgpu_h3m_grid_create(rcgs%h, rcgs%device, rcgs%ncol, rcgs%n_src, rcgs%method, &
                                           rcgs%left_bndr_ord, rcgs%right_bndr_ord)
         if (errstat /= hor3map_noerr) return
         rcgs%initialized = .true.
      elseif (rcgs%n_src /= size(x_edge_src, 1) - 1 .or. rcgs%ncol /= size(x_edge_src, 2)) then
         errstat = 2                                                     ! hor3map_resizing_initialized_rcgs
         return
      endif
      errstat = blomgpu_h3m_prepare_reconstruction(rcgs%h, x_edge_src)
   end function

   function reconstruct(rcgs, rcss, u_src) result(errstat)               ! mod_hor3map.F90:4145
      type(recon_grd_struct), intent(inout) :: rcgs
      type(recon_src_struct), intent(inout) :: rcss
      real(r8), dimension(:,:), intent(in) :: u_src                     ! (n_src, ncol)
      integer :: errstat
      if (.not. rcgs%initialized) then
         errstat = 6                                                     ! hor3map_recon_not_prepared
         return
      endif
      if (size(u_src, 1) /= rcgs%n_src .or. size(u_src, 2) /= rcgs%ncol) then
         errstat = 11                                                    ! hor3map_src_size_mismatch
         return
      endif
      if (rcss%initialized .and. .not. c_associated(rcss%grid, rcgs%h)) call free_rcss(rcss)
      if (.not. rcss%initialized) then
         errstat = blomgpu_h3m_src_create(rcgs%h, rcss%h, rcss%limiting, merge(1, 0, rcss%pc_left_bndr), &
                                          merge(1, 0, rcss%pc_right_bndr))
         if (errstat /= hor3map_noerr) return
         rcss%grid = rcgs%h
         rcss%initialized = .true.
      endif
      errstat = blomgpu_h3m_reconstruct(rcgs%h, rcss%h, u_src)
   end function

   function extract_polycoeff(rcss, polycoeff) result(errstat)          ! mod_hor3map.F90:4274
      type(recon_src_struct), intent(inout) :: rcss
      real(r8), dimension(:,:,:), intent(out) :: polycoeff              ! (p_ord+1, n_src, ncol)
      integer :: errstat
      if (.not. rcss%initialized) then
         errstat = 16                                                    ! hor3map_recon_not_available
         return
      endif
      errstat = blomgpu_h3m_extract_polycoeff(rcss%h, polycoeff)
   end function

   function regrid(rcss, u_edge_grd, x_edge_grd, missing_value, regrid_method) result(errstat)   ! :4461
      type(recon_src_struct), intent(inout) :: rcss
      real(r8), dimension(:,:), intent(in) :: u_edge_grd
      real(r8), dimension(:,:), intent(out) :: x_edge_grd
      real(r8), intent(in) :: missing_value
      integer, optional, intent(in) :: regrid_method
      integer :: errstat, m
      if (.not. rcss%initialized) then
         errstat = 16
         return
      endif
      if (size(x_edge_grd, 1) /= size(u_edge_grd, 1)) then
         errstat = 18                                                    ! hor3map_grd_size_mismatch
         return
      endif
      m = hor3map_regrid_method_1
      if (present(regrid_method)) m = regrid_method
      errstat = blomgpu_h3m_regrid(rcss%h, size(u_edge_grd, 1), u_edge_grd, x_edge_grd, missing_value, m)
   end function

   function prepare_remapping(rcgs, rms, x_edge_dst) result(errstat)    ! mod_hor3map.F90:3947
      type(recon_grd_struct), intent(inout) :: rcgs
      type(remap_struct), intent(inout) :: rms
      real(r8), dimension(:,:), intent(in) :: x_edge_dst                ! (n_dst+1, ncol)
      integer :: errstat
      if (.not. rcgs%initialized) then
         errstat = 6
         return
      endif
      if (rms%initialized) then
         if (.not. c_associated(rms%grid, rcgs%h) .or. rms%n_dst /= size(x_edge_dst, 1) - 1) call free_rms(rms)
      endif
      if (.not. rms%initialized) then
         rms%n_dst = size(x_edge_dst, 1) - 1
         errstat = blomgpu_h3m_map_create(rcgs%h, rms%h, rms%n_dst)
         if (errstat /= hor3map_noerr) return
         rms%grid = rcgs%h
         rms%initialized = .true.
      endif
      errstat = blomgpu_h3m_prepare_remapping(rcgs%h, rms%h, x_edge_dst)
   end function

   function remap(rcss, rms, u_dst) result(errstat)                      ! mod_hor3map.F90:4559
      type(recon_src_struct), intent(inout) :: rcss
      type(remap_struct), intent(inout) :: rms
      real(r8), dimension(:,:), intent(out) :: u_dst                    ! (n_dst, ncol)
      integer :: errstat
      if (.not. rcss%initialized) then
         errstat = 16
         return
      endif
      if (.not. rms%initialized) then
         errstat = 19                                                    ! hor3map_remap_not_prepared
         return
      endif
      if (size(u_dst, 1) /= rms%n_dst) then
         errstat = 20                                                    ! hor3map_dst_size_mismatch
         return
      endif
      errstat = blomgpu_h3m_remap(rcss%h, rms%h, u_dst)
   end function

   subroutine free_rcss(rcss)                                            ! mod_hor3map.F90:4912
      type(recon_src_struct), intent(inout) :: rcss
      if (rcss%initialized) call blomgpu_h3m_src_free(rcss%h)
      rcss%h = c_null_ptr
      rcss%grid = c_null_ptr
      rcss%initialized = .false.
   end subroutine

   subroutine free_rms(rms)                                              ! mod_hor3map.F90:4937
      type(remap_struct), intent(inout) :: rms
      if (rms%initialized) call blomgpu_h3m_map_free(rms%h)
      rms%h = c_null_ptr
      rms%grid = c_null_ptr
      rms%initialized = .false.
   end subroutine

   ! free_rcgs (mod_hor3map.F90:4858) also releases the dependants on the device; their Fortran
   ! handles must be passed through free_rcss/free_rms BEFORE it, or simply be dropped after it.
   subroutine free_rcgs(rcgs)
      type(recon_grd_struct), intent(inout) :: rcgs
      if (rcgs%initialized) call blomgpu_h3m_grid_free(rcgs%h)
      rcgs%h = c_null_ptr
      rcgs%initialized = .false.
   end subroutine

   function hor3map_errstr(errstat) result(errstr)                       ! mod_hor3map.F90:4955
      integer, intent(in) :: errstat
      character(len=120) :: errstr
      character(kind=c_char), pointer :: p(:)
      type(c_ptr) :: cp
      integer :: i
      errstr = ' '
      cp = blomgpu_h3m_errstr(errstat)
      if (.not. c_associated(cp)) return
      call c_f_pointer(cp, p, [120])
      do i = 1, 120
         if (p(i) == c_null_char) exit
         errstr(i:i) = p(i)
      enddo
   end function

end module mod_hor3map_gpu
