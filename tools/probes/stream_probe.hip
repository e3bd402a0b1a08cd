// What a streaming kernel over MANY arrays reaches on this box, by access width: NI input arrays and NO output arrays of n doubles each,
// one point per thread with 8-byte accesses (the form of the model's per-level kernels) against two points per thread with 16-byte
// accesses, and the same with the arrays' loads issued in groups.  Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Ptrs { const double *in[24]; double *out[8]; };
template <int NI, int NO>
__global__ void k8(Ptrs P, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double s = 0.;
#pragma unroll
  for (int a = 0; a < NI; a++) s += P.in[a][t];
#pragma unroll
  for (int o = 0; o < NO; o++) P.out[o][t] = s + o;
}
template <int NI, int NO>
__global__ void k16(Ptrs P, size_t n2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n2) return;
  double2 s = {0., 0.};
#pragma unroll
  for (int a = 0; a < NI; a++) { const double2 v = ((const double2 *)P.in[a])[t]; s.x += v.x; s.y += v.y; }
#pragma unroll
  for (int o = 0; o < NO; o++) { double2 w = {s.x + o, s.y + o}; ((double2 *)P.out[o])[t] = w; }
}
template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(a, 0); for (int r = 0; r < reps; r++) f(); hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e-3;
}
int main() {
  const size_t n = (size_t)216 * 520 * 53;            // one 3-D field of the channel
  Ptrs P;
  for (int a = 0; a < 24; a++) { double *p; hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8); P.in[a] = p; }
  for (int o = 0; o < 8; o++) { double *p; hipMalloc(&p, n * 8); P.out[o] = p; }
  const int B = 256;
#define RUN(NI, NO) { \
    double t8 = timeit([&] { hipLaunchKernelGGL((k8<NI, NO>), dim3((n + B - 1) / B), dim3(B), 0, 0, P, n); }, 20); \
    double t16 = timeit([&] { hipLaunchKernelGGL((k16<NI, NO>), dim3((n / 2 + B - 1) / B), dim3(B), 0, 0, P, n / 2); }, 20); \
    printf("in %2d out %d: 8 B/lane %7.1f us %5.2f TB/s | 16 B/lane %7.1f us %5.2f TB/s\n", NI, NO, t8 * 1e6, (NI + NO) * n * 8 / t8 / 1e12, t16 * 1e6, (NI + NO) * n * 8 / t16 / 1e12); }
  RUN(1, 1) RUN(2, 1) RUN(4, 2) RUN(8, 2) RUN(12, 4) RUN(18, 6) RUN(24, 8)
  return 0;
}
