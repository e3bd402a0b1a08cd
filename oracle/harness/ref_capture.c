/* TEST INFRASTRUCTURE (oracle): tiny C helper for the reference harness.
 * Fortran passes a (contiguous, statically allocated) module array by reference;
 * we hand the address back so Python can wrap it as a numpy view. */
#include <stddef.h>
void ref_capture_r8(double *a, void **out) { *out = (void *)a; }
void ref_capture_i4(int *a, void **out) { *out = (void *)a; }
/* writes through the address of a module scalar: the reference declares its run-time tracer count `ntr` PROTECTED
 * (trc/mod_tracers.F90:46), so the harness cannot assign it in Fortran (ref_set_ntr, ref_harness.F90) */
void ref_poke_i4(int *a, int v) { *a = v; }
