// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a 16 B/lane stream ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel below moves a KNOWN number of bytes through arrays far larger than the 256 MiB Infinity Cache:
//   k_copy8     8 B/lane read + 8 B/lane write, unit stride (the fp64 point kernels)
//   k_copy16    16 B/lane read + write (the guide's reference pattern)
//   k_read8     8 B/lane read, one partial sum per wavefront written
//   k_stencil8  5-point stencil of 8 B/lane loads (c, c+-1, c+-ni) on (ni x nj) planes, one 8 B/lane store:
//               compulsory read bytes = one pass over the array
//   k_column8   thread per column walking nk planes (the column kernels' pattern): same bytes as k_copy8
// Usage: rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib ;  rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib
// Prints the byte counts each launch moves; tools/fetch_calib_summary.py divides the counters by them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_copy8(const double *__restrict__ a, double *__restrict__ b, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) b[t] = a[t] + 1.0;
}
__global__ void k_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n2) { double2 v = a[t]; v.x += 1.0; v.y += 1.0; b[t] = v; }
}
__global__ void k_read8(const double *__restrict__ a, double *__restrict__ out, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double v = t < n ? a[t] : 0.0;
  for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) out[t >> 6] = v;
}
__global__ void k_stencil8(const double *__restrict__ a, double *__restrict__ b, int ni, int nj, int nk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, np = ni * nj;
  if (t >= np) return;
  const int i = t % ni, j = t / ni;
  const size_t c = (size_t)blockIdx.y * np + t;
  double v = a[c];
  if (i > 0 && i < ni - 1 && j > 0 && j < nj - 1) v = v + 0.25 * (a[c - 1] + a[c + 1] + a[c - ni] + a[c + ni]);
  b[c] = v;
}
__global__ void k_column8(const double *__restrict__ a, double *__restrict__ b, int np, int nk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np) return;
  double acc = 0.0;
  for (int k = 0; k < nk; k++) { acc = acc + a[(size_t)k * np + t]; b[(size_t)k * np + t] = acc; }
}

// the column pattern with U levels' loads in flight per thread (manual software pipelining)
template <int U>
__global__ void k_column8_u(const double *__restrict__ a, double *__restrict__ b, int np, int nk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np) return;
  double acc = 0.0;
  int k = 0;
  for (; k + U <= nk; k += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = a[(size_t)(k + u) * np + t];
#pragma unroll
    for (int u = 0; u < U; u++) { acc = acc + v[u]; b[(size_t)(k + u) * np + t] = acc; }
  }
  for (; k < nk; k++) { acc = acc + a[(size_t)k * np + t]; b[(size_t)k * np + t] = acc; }
}

int main() {
  const int ni = 216, nj = 520, nk = 1060;                 // 1060 planes of the channel's padded size: 0.95 GB per array
  const size_t n = (size_t)ni * nj * nk, bytes = n * 8;
  double *a, *b;
  CHK(hipMalloc(&a, bytes));
  CHK(hipMalloc(&b, bytes));
  CHK(hipMemset(a, 0, bytes));
  CHK(hipMemset(b, 0, bytes));
  const int B = 256;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_copy8, dim3((n + B - 1) / B), dim3(B), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_copy16, dim3((n / 2 + B - 1) / B), dim3(B), 0, 0, (const double2 *)a, (double2 *)b, n / 2);
    hipLaunchKernelGGL(k_read8, dim3((n + B - 1) / B), dim3(B), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_stencil8, dim3((ni * nj + B - 1) / B, nk), dim3(B), 0, 0, a, b, ni, nj, nk);
    hipLaunchKernelGGL(k_column8, dim3((ni * nj + 63) / 64), dim3(64), 0, 0, a, b, ni * nj, nk);
    hipLaunchKernelGGL(k_column8_u<4>, dim3((ni * nj + 63) / 64), dim3(64), 0, 0, a, b, ni * nj, nk);
    hipLaunchKernelGGL(k_column8_u<8>, dim3((ni * nj + 63) / 64), dim3(64), 0, 0, a, b, ni * nj, nk);
    hipLaunchKernelGGL(k_column8_u<16>, dim3((ni * nj + 63) / 64), dim3(64), 0, 0, a, b, ni * nj, nk);
  }
  CHK(hipDeviceSynchronize());
  printf("bytes_per_array %zu\n", bytes);
  printf("k_copy8 read %zu write %zu\nk_copy16 read %zu write %zu\nk_read8 read %zu write %zu\n"
         "k_stencil8 read %zu write %zu\nk_column8 read %zu write %zu\n",
         bytes, bytes, bytes, bytes, bytes, n / 64 * 8, bytes, bytes, bytes, bytes);
  hipFree(a);
  hipFree(b);
  return 0;
}
