// what ale_regrid_remap hands to the neutral diffusion (stage_ndiff.hip): the reconstructed source columns and the regridded
// interfaces, as planes [level][column] over the model's padded plane
#pragma once
#include "blomgpu_internal.h"

struct NdArgs {
  const double *psrc, *pdst;   // source / destination interfaces, kk+1 planes each
  const int *ksmx, *kdmx;      // deepest source / destination layer with mass
  const double *tpc;           // polynomial coefficients of T, S, tracers: [field][layer][coefficient 0..npc-1] planes
  const double *tsd;           // T and S at both interfaces of every source layer: [field 0..1][layer][upper, lower]
  const double *drt, *drs;     // drho/dT, drho/dS there: [layer][upper, lower]
  double *flx;                 // flux convergence: [destination layer][field] planes
  double *puv;                 // pu, pv as they are when the stage starts ((kk+1) planes each): the velocity part of
                               // ale_regrid_remap, which runs beside the diffusion, rewrites them
  double *scr;                 // the flux kernel's per-face work arrays, ndiff_scratch_planes(kk) planes of 2 nplane faces
  // the fluxes a face found, in search order (u-faces first, then the v-faces: 2 nplane faces per plane)
  int *rec_n, *rec_k, *rec_s;  // their number; per record the destination layers kd_m | kd_p << 16; the source layers and how the
                               // values at the two neutral interfaces are taken: ks_m | ks_p << 8 | four 2-bit kinds << 16
  double *rec_g;               // per record: positions of the two interfaces in the source layers (m, m, p, p), thickness,
                               // mean pressure of the upper and of the lower interface
  double *rec_f;               // per record ntr_loc fluxes (NaN: withheld by the sign tests)
  int nrec_max;
  int flux_zero;               // utflx .. vsflx (level m) are zero when the stage starts (inside blomgpu_step: init_fluxes, nothing added since)
  long long *prof;             // debug (blomgpu_dbg_kprof, its own buffer: 6 words a wave, bound-checked against prof_words): per wave of
                               // k_ndiff_flux its start / end timestamp and record count
  int prof_words;
  int kk, npc, ntr_loc, mm, nn, surface_align;
};

size_t ndiff_scratch_planes(int kk);
int st_ndiff_prep_flux(blomgpu_ctx *c, hipStream_t st, hipEvent_t ev_snap, NdArgs A, int *ksmx, int *kdmx, double *tsd, double *drt, double *drs);
