// Probe: can a resident kernel hand work to another stream and get an answer back without ending?
//   kernel (stream A): for it = 1..N: store P = it (system scope); poll L >= it (bounded); record clock
//   stream B:          for it = 1..N: hipStreamWaitValue32(P >= it); tiny kernel; hipStreamWriteValue32(L = it)
// Prints the round-trip time per iteration.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_ping(unsigned *P, unsigned *L, int n, long long *t, int *fail) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int it = 1; it <= n; ++it) {
    const long long t0 = wall_clock64();
    __hip_atomic_store(P, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned spins = 0;
    while (__hip_atomic_load(L, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)it) {
      if (++spins > 20000000u) { *fail = it; return; }
      __builtin_amdgcn_s_sleep(2);
    }
    t[it - 1] = wall_clock64() - t0;
  }
}
__global__ void k_touch(double *x) { x[threadIdx.x] += 1.0; }

int main() {
  int can = 0;
  CHK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  unsigned *sig = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(signal): %s\n", hipGetErrorString(e));
  if (e != hipSuccess) CHK(hipMalloc((void **)&sig, 64));
  CHK(hipMemset(sig, 0, 64));
  unsigned *P = sig, *L = sig + 8;
  const int n = 20;
  long long *t; int *fail; double *x;
  CHK(hipMalloc((void **)&t, n * sizeof(long long)));
  CHK(hipMalloc((void **)&fail, sizeof(int)));
  CHK(hipMalloc((void **)&x, 64 * sizeof(double)));
  CHK(hipMemset(fail, 0, sizeof(int)));
  CHK(hipMemset(x, 0, 64 * sizeof(double)));
  hipStream_t A, B;
  CHK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_ping, dim3(1), dim3(64), 0, A, P, L, n, t, fail);
  for (int it = 1; it <= n; ++it) {
    CHK(hipStreamWaitValue32(B, P, (unsigned)it, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, B, x);
    CHK(hipStreamWriteValue32(B, L, (unsigned)it, 0));
  }
  CHK(hipStreamSynchronize(A));
  CHK(hipStreamSynchronize(B));
  long long ht[n]; int hf = 0;
  CHK(hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost));
  CHK(hipMemcpy(&hf, fail, sizeof(int), hipMemcpyDeviceToHost));
  printf("fail=%d  round trips [us at 100 MHz wall clock]:", hf);
  for (int i = 0; i < n; ++i) printf(" %.1f", ht[i] / 100.0);
  printf("\n");
  return hf ? 1 : 0;
}
