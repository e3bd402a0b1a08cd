// Stages of the hot path whose kernels are not written yet fail loudly.
#include "blomgpu_internal.h"
#define TODO6(nm) int st_##nm(blomgpu_ctx *c, int, int, int, int, int, int) { return ctx_fail(c, #nm ": HIP kernels not built yet"); }
TODO6(eddtra)
