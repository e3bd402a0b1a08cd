/* blomgpu_hor3map.h -- C ABI of the batched HOR3MAP library on MI355X.
 *
 * Replaces the public interface of the reference's phy/mod_hor3map.F90 (types :153,:207,:242;
 * procedures :3834 prepare_reconstruction, :3947 prepare_remapping, :4145 reconstruct,
 * :4274 extract_polycoeff, :4461 regrid, :4559 remap, :4858-:4953 free_*, :4955 hor3map_errstr)
 * for MANY columns at once: where the reference is called once per column inside i/j loops
 * (phy/mod_ale_regrid_remap.F90:224-247, :405-415, :1038-1046), here one call processes all
 * `ncol` columns of a slab, one GPU thread per column.
 *
 * Array arguments have the caller's Fortran shape with the column index last, e.g.
 * x_edge_src(n_src+1, ncol), u_src(n_src, ncol), polycoeff(p_ord+1, n_src, ncol): exactly what
 * `p_src(:,i)` / `trc_1d(:,nt)` / `tpc_src(:,:,nt,i)` are in the reference's callers.  Pointers
 * are host pointers unless blomgpu_h3m_set_io(grid, 1, ..) declared them device pointers.
 *
 * Return value of the compute entries: the reference's errstat (hor3map_noerr = 0, codes
 * :60-83) of the first column, in column order, that failed in this call -- what a host loop
 * `do i ...; errstat = reconstruct(...); if (errstat /= hor3map_noerr) stop` would have seen.
 * blomgpu_h3m_errstat returns the per-column codes of the last call.  Negative values are
 * failures of the device layer itself (no HIP device, allocation, bad handle); text from
 * blomgpu_h3m_errstr.  There is no host fallback. */
#ifndef BLOMGPU_HOR3MAP_H
#define BLOMGPU_HOR3MAP_H
#ifdef __cplusplus
extern "C" {
#endif

/* option and error codes: the reference's parameters (mod_hor3map.F90:47-83) */
enum {
  BLOMGPU_H3M_PCM = 100, BLOMGPU_H3M_PLM = 101, BLOMGPU_H3M_PPM = 102, BLOMGPU_H3M_PQM = 103,
  BLOMGPU_H3M_NO_LIMITING = 200, BLOMGPU_H3M_MONOTONIC = 201, BLOMGPU_H3M_NON_OSCILLATORY = 203,
  BLOMGPU_H3M_NON_OSCILLATORY_POSDEF = 204,
  BLOMGPU_H3M_REGRID_METHOD_1 = 301, BLOMGPU_H3M_REGRID_METHOD_2 = 302
};

typedef struct blomgpu_h3m_grid blomgpu_h3m_grid;   /* recon_grd_struct, all columns */
typedef struct blomgpu_h3m_src  blomgpu_h3m_src;    /* recon_src_struct */
typedef struct blomgpu_h3m_map  blomgpu_h3m_map;    /* remap_struct */

/* initialize_rcgs (:3607): method, left/right_bndr_ord as the type's components (0 = scheme maximum). */
int  blomgpu_h3m_grid_create(blomgpu_h3m_grid **out, int device, int ncol, int n_src, int method,
                             int left_bndr_ord, int right_bndr_ord);
/* initialize_rcss (:3731): limiting, pc_left_bndr, pc_right_bndr as the type's components. */
int  blomgpu_h3m_src_create(blomgpu_h3m_grid *grid, blomgpu_h3m_src **out, int limiting,
                            int pc_left_bndr, int pc_right_bndr);
/* initialize_rms (:3792) */
int  blomgpu_h3m_map_create(blomgpu_h3m_grid *grid, blomgpu_h3m_map **out, int n_dst);
void blomgpu_h3m_src_free(blomgpu_h3m_src *src);      /* free_rcss :4912 */
void blomgpu_h3m_map_free(blomgpu_h3m_map *map);      /* free_rms  :4937 */
void blomgpu_h3m_grid_free(blomgpu_h3m_grid *grid);   /* free_rcgs :4858 (frees dependants too) */

/* device_pointers: 1 = array arguments are device pointers (same shapes); 2 = device pointers to arrays that are already
 * in the library's own layout, [level][column] with the column fastest -- the layout of the model's 3-D fields (no transpose:
 * the dynamical core's ale_regrid_remap calls the engine this way; the kernels read and write those arrays where they lie,
 * in stream order: they must not be changed before the call's kernels have run).  check_errors = 0 defers
 * the per-call status read-back (calls then return 0 unless the device layer fails; use
 * blomgpu_h3m_errstat / blomgpu_h3m_sync). */
int  blomgpu_h3m_set_io(blomgpu_h3m_grid *grid, int device_pointers, int check_errors);

int  blomgpu_h3m_prepare_reconstruction(blomgpu_h3m_grid *grid, const double *x_edge_src);
int  blomgpu_h3m_reconstruct(blomgpu_h3m_grid *grid, blomgpu_h3m_src *src, const double *u_src);
int  blomgpu_h3m_extract_polycoeff(blomgpu_h3m_src *src, double *polycoeff);
int  blomgpu_h3m_regrid(blomgpu_h3m_src *src, int n_grd, const double *u_edge_grd, double *x_edge_grd,
                        double missing_value, int regrid_method);
int  blomgpu_h3m_prepare_remapping(blomgpu_h3m_grid *grid, blomgpu_h3m_map *map, const double *x_edge_dst);
int  blomgpu_h3m_remap(blomgpu_h3m_src *src, blomgpu_h3m_map *map, double *u_dst);

/* The same for up to 8 source fields on one grid in ONE launch -- the tracer loops of the reference's
 * callers (do nt = 1,ntr: reconstruct(rcgs, trc_rcss(nt), ...) / remap(trc_rcss(nt), rms, ...),
 * phy/mod_ale_regrid_remap.F90:231-243, :1044-1053).  The column routines are latency bound, so n fields
 * cost about the time of one.  Return value: the errstat of the lowest failing column over the fields. */
int  blomgpu_h3m_reconstruct_many(blomgpu_h3m_grid *grid, int nf, blomgpu_h3m_src *const *srcs,
                                  const double *const *u_srcs);
int  blomgpu_h3m_remap_many(int nf, blomgpu_h3m_src *const *srcs, blomgpu_h3m_map *map, double *const *u_dsts);

/* per-column status of the last call (ncol ints, host), and per-column n_src_actual / method_actual */
int  blomgpu_h3m_errstat(blomgpu_h3m_grid *grid, int *errstat);
int  blomgpu_h3m_grid_info(blomgpu_h3m_grid *grid, int *n_src_actual, int *method_actual);
int  blomgpu_h3m_sync(blomgpu_h3m_grid *grid);
/* time of the last compute kernel in ms (HIP events on the library's stream) */
int  blomgpu_h3m_last_kernel_ms(blomgpu_h3m_grid *grid, float *ms);
const char *blomgpu_h3m_errstr(int errstat);          /* hor3map_errstr :4955 */

#ifdef __cplusplus
}
#endif
#endif
