// What a plain copy reaches on THIS box: 16 B/lane read + write over arrays of 0.95 GB each (>> the 256 MiB Infinity Cache), HIP events
// around 10 launches after 2 of warm-up.  The boxes of the pool differ by ~15 % here (5.2 - 6.2 TB/s seen in round 5), and every
// byte-bound kernel of the step with them: tools/bounds_table.py prices the byte bound of a profile with the rate printed by the
// same job.  Usage: tools/probes/copy_rate  ->  "copy_rate_TBps 5.93"
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n2) { double2 v = a[t]; v.x += 1.0; v.y += 1.0; b[t] = v; }
}
int main() {
  const size_t n = (size_t)216 * 520 * 1060, bytes = n * 8;
  double *a, *b;
  CHK(hipMalloc(&a, bytes));
  CHK(hipMalloc(&b, bytes));
  CHK(hipMemset(a, 0, bytes));
  CHK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int B = 256, reps = 10;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_copy16, dim3((n / 2 + B - 1) / B), dim3(B), 0, 0, (const double2 *)a, (double2 *)b, n / 2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy16, dim3((n / 2 + B - 1) / B), dim3(B), 0, 0, (const double2 *)a, (double2 *)b, n / 2);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("copy_rate_TBps %.3f\n", 2.0 * bytes * reps / (ms * 1e-3) / 1e12);
  hipFree(a);
  hipFree(b);
  return 0;
}
