/* TEST INFRASTRUCTURE: host build of blom_amd/csrc/exp_libm.h, so that the function the kernels call can be
 * compared with the host libm's exp() where no GPU exists.  Never linked into the product. */
#include "../../blom_amd/csrc/exp_libm.h"
void exp_hostcheck(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = exp_libm(x[i]);
}
