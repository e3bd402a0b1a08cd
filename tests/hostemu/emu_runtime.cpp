// TEST INFRASTRUCTURE -- runtime of the host emulation (see shim/hip/hip_runtime.h): grid loop, one ucontext fiber
// per thread of a block, barrier = every live fiber has yielded.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <time.h>
#include <ucontext.h>
#include <vector>

namespace hostemu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber { ucontext_t ctx; bool done = false, started = false; };
struct Sched {
  ucontext_t main;
  std::vector<Fiber> fib;
  char *stacks = nullptr;
  size_t nstacks = 0;
  int cur = -1;             // fiber that is running, -1: none (plain mode)
  bool in_fibers = false;
  KernelCall *call = nullptr;
  dim3 block;
  void *lds = nullptr;
  size_t lds_cap = 0;
};
thread_local Sched S;

void set_tid(int t) {
  t_threadIdx.x = t % S.block.x;
  t_threadIdx.y = (t / S.block.x) % S.block.y;
  t_threadIdx.z = t / (S.block.x * S.block.y);
}
void fiber_entry() {
  Sched &s = S;
  s.call->run();
  s.fib[s.cur].done = true;
  swapcontext(&s.fib[s.cur].ctx, &s.main);
}
void need_stacks(size_t n) {
  if (n <= S.nstacks) return;
  if (S.stacks) munmap(S.stacks, S.nstacks * STACK);
  S.stacks = (char *)mmap(nullptr, n * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (S.stacks == MAP_FAILED) { perror("hostemu: mmap"); abort(); }
  S.nstacks = n;
}
}  // namespace

void *dyn_lds() { return S.lds; }

void syncthreads() {
  Sched &s = S;
  if (!s.in_fibers) { fprintf(stderr, "hostemu: __syncthreads reached by a thread after thread 0 of its block finished without one\n"); abort(); }
  swapcontext(&s.fib[s.cur].ctx, &s.main);      // yield; resumed when every live fiber has arrived
}

void launch(KernelCall &k, dim3 grid, dim3 block, size_t shmem) {
  Sched &s = S;
  const int nthr = block.x * block.y * block.z;
  s.call = &k;
  s.block = block;
  t_blockDim = block;
  t_gridDim = grid;
  if (shmem > s.lds_cap) { free(s.lds); s.lds = malloc(shmem); s.lds_cap = shmem; }
  need_stacks(nthr);
  s.fib.resize(nthr);
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        t_blockIdx = dim3(bx, by, bz);
        if (shmem) memset(s.lds, 0xff, shmem);           // NaN-poison the dynamic LDS of every block
        auto mk = [&](int t) {
          Fiber &f = s.fib[t];
          f.done = false;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = s.stacks + (size_t)t * STACK;
          f.ctx.uc_stack.ss_size = STACK;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_entry, 0);
        };
        // thread 0 runs as a fiber; if it finishes without reaching a barrier the kernel has none (every thread of a
        // block must reach the same barriers) and the other threads run as plain calls
        s.in_fibers = true;
        mk(0);
        s.cur = 0;
        set_tid(0);
        swapcontext(&s.main, &s.fib[0].ctx);
        if (s.fib[0].done) {
          s.in_fibers = false;
          s.cur = -1;
          for (int t = 1; t < nthr; t++) { set_tid(t); k.run(); }
          continue;
        }
        for (int t = 1; t < nthr; t++) mk(t);
        int live = nthr;
        bool first = true;
        while (live > 0) {
          live = 0;
          for (int t = 0; t < nthr; t++) {
            if (first && t == 0) { live++; continue; }      // thread 0 already waits at the first barrier
            if (s.fib[t].done) continue;
            s.cur = t;
            set_tid(t);
            swapcontext(&s.main, &s.fib[t].ctx);
            if (!s.fib[t].done) live++;
          }
          first = false;
        }
        s.in_fibers = false;
        s.cur = -1;
      }
}
}  // namespace hostemu

double hostemu_now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
