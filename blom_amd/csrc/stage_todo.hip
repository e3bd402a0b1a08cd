// Stages of the hot path whose kernels are not written yet fail loudly (none at present).
#include "blomgpu_internal.h"
