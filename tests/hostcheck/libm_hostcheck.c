/* TEST INFRASTRUCTURE: host build of blom_amd/csrc/pow_libm.h (and log_libm.h, tanh_libm.h), so that the functions the kernels
 * call can be compared with the host libm's where no GPU exists.  Never linked into the product. */
#include "../../blom_amd/csrc/pow_libm.h"
void pow_hostcheck(int n, const double *x, const double *y, double *z) {
  for (int i = 0; i < n; i++) z[i] = pow_libm(x[i], y[i]);
}
#include "../../blom_amd/csrc/sin_libm.h"
#include "../../blom_amd/csrc/atan2_libm.h"
void sin_hostcheck(int n, const double *x, double *z) {
  for (int i = 0; i < n; i++) z[i] = sin_libm(x[i]);
}
void atan2_hostcheck(int n, const double *y, const double *x, double *z) {
  for (int i = 0; i < n; i++) z[i] = atan2_libm(y[i], x[i]);
}
