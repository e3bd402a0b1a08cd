/* TEST INFRASTRUCTURE (oracle): tiny C helper for the reference harness.
 * Fortran passes a (contiguous, statically allocated) module array by reference;
 * we hand the address back so Python can wrap it as a numpy view. */
#include <stddef.h>
void ref_capture_r8(double *a, void **out) { *out = (void *)a; }
void ref_capture_i4(int *a, void **out) { *out = (void *)a; }
