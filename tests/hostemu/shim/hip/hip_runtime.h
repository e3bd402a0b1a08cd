// TEST INFRASTRUCTURE -- host emulation of the small part of the HIP runtime and device language that
// blom_amd/csrc/*.hip uses.  With this header first on the include path the device library's OWN sources compile with
// g++ into tests/hostemu/libblomgpu_hostemu.so: kernels become ordinary functions, a launch runs the grid block by
// block with one fiber per thread (so __syncthreads(), LDS and the per-thread index arithmetic behave as on the
// device), device memory is host memory.  Purpose: check a kernel's LOGIC (index ranges, stencil reach, LDS rings,
// pack/unpack of the tile exchange with several ranks in one process) bit for bit against the oracle on the CPU,
// before GPU minutes are spent.  It says nothing about speed, memory-model hazards between workgroups or wavefront
// effects; the `-m gpu` tests on the real device remain the parity tests.
// Nothing under blom_amd/, bench.py or __graft_entry__.smoke() may load the emulated library: the product has no
// CPU path (blomgpu_create fails without a HIP device).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define BLOM_HOSTEMU 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hostemu::dyn_lds();

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };

namespace hostemu {
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void *dyn_lds();
void syncthreads();
// runs body(tid-linear) for every thread of every block; fibers are used as soon as a thread reaches a barrier
struct KernelCall { virtual void run() = 0; virtual ~KernelCall() {} };
void launch(KernelCall &k, dim3 grid, dim3 block, size_t shmem);
}  // namespace hostemu
#define threadIdx hostemu::t_threadIdx
#define blockIdx hostemu::t_blockIdx
#define blockDim hostemu::t_blockDim
#define gridDim hostemu::t_gridDim
#define warpSize 64
inline void __syncthreads() { hostemu::syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

#include <functional>
template <class K, class... A>
inline void hostemu_launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  struct C : hostemu::KernelCall {
    std::function<void()> fn;
    void run() override { fn(); }
  } c;
  c.fn = [=]() { kernel(args...); };
  hostemu::launch(c, grid, block, shmem);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hostemu_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

// ---- runtime API (synchronous; "device" memory is host memory) --------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct hostemu_stream *hipStream_t;
typedef struct hostemu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
inline const char *hipGetErrorString(hipError_t) { return "hostemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 0; strcpy(p->name, "hostemu"); return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = *tot = (size_t)1 << 46; return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
double hostemu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hostemu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = hostemu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
// graphs: capture is not emulated; the library checks hostemu and steps eagerly
typedef struct hostemu_graph *hipGraph_t;
typedef struct hostemu_graphexec *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorUnknown; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipErrorUnknown; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return hipErrorUnknown; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorUnknown; }

// ---- device functions ---------------------------------------------------------------------------------
inline long long wall_clock64() { return 0; }
inline long long clock64() { return 0; }
inline unsigned long long __double_as_longlong(double x) { unsigned long long b; memcpy(&b, &x, 8); return b; }
inline double __longlong_as_double(long long b) { double x; memcpy(&x, &b, 8); return x; }
// atomics: blocks run one after the other on the launching host thread; tiles of one process run on several host
// threads but never share these words
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
// wavefront shuffles are not emulated (threads of a block run one at a time): code paths that use them are
// optional variants (diffus_shfl) and are not reachable under the emulation
template <class T> inline T __shfl_up(T v, int) { fprintf(stderr, "hostemu: __shfl_up is not emulated\n"); abort(); return v; }
template <class T> inline T __shfl_down(T v, int) { fprintf(stderr, "hostemu: __shfl_down is not emulated\n"); abort(); return v; }
template <class T> inline T __shfl(T v, int) { fprintf(stderr, "hostemu: __shfl is not emulated\n"); abort(); return v; }
inline int __all(int pred) { return pred; }      // only the persistent barotp kernel votes; it is never launched here
inline int __any(int pred) { return pred; }
#include <algorithm>
using std::max;
using std::min;
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }
