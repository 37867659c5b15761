// launch_rate.hip -- what bounds a step of ~40 dependent small kernels (the reference's own 8 000-particle scenes: 0.21 ms/step)?
//   hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip && ./launch_rate
// (a) dependent launches of a trivial kernel in one stream, kernel argument of 8 / 256 / 704 bytes (Dev is ~700 B, by value);
// (b) the same launches captured once in a hipGraph and replayed;
// (c) one cooperative launch with grid-wide barriers (cooperative groups grid.sync()) at 32 / 128 / 512 / 1024 blocks of 256;
// (d) a hand-written grid barrier (one device-scope atomic counter + spin on a sense word) in a persistent kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
#include <chrono>
namespace cg = cooperative_groups;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int N> struct Arg { int *p; char pad[N - 8]; };
template <> struct Arg<8> { int *p; };
template <int N> __global__ void k_touch(Arg<N> a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += 1; }
__global__ void k_coop(int *p, int iters) {
  cg::grid_group g = cg::this_grid();
  for (int k = 0; k < iters; k++) { if (g.thread_rank() == 0) p[0] += 1; g.sync(); }
}
// sense-reversing barrier: count arrivals with a device-scope atomic, the last one flips the generation word everybody spins on
__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
__global__ void k_persist(int *p, unsigned *bar, int iters) {
  for (int k = 0; k < iters; k++) { if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1; grid_barrier(bar, bar + 32, gridDim.x); }
}
template <int N> static double run_launches(hipStream_t s, int *d, int n) {
  Arg<N> a{}; a.p = d;
  for (int k = 0; k < 50; k++) hipLaunchKernelGGL(k_touch<N>, dim3(32), dim3(256), 0, s, a);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_touch<N>, dim3(32), dim3(256), 0, s, a);
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int *d; CHK(hipMalloc(&d, 4096)); CHK(hipMemset(d, 0, 4096));
  unsigned *bar; CHK(hipMalloc(&bar, 4096)); CHK(hipMemset(bar, 0, 4096));
  const int n = 2000;
  printf("dependent launches in one stream, us per launch: arg 8 B %.2f, 256 B %.2f, 704 B %.2f\n", run_launches<8>(s, d, n), run_launches<256>(s, d, n), run_launches<704>(s, d, n));
  // graph of 40 kernel nodes (a step's worth), replayed
  for (int nn : {8, 704}) {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 40; k++) { if (nn == 8) { Arg<8> a{}; a.p = d; hipLaunchKernelGGL(k_touch<8>, dim3(32), dim3(256), 0, s, a); } else { Arg<704> a{}; a.p = d; hipLaunchKernelGGL(k_touch<704>, dim3(32), dim3(256), 0, s, a); } }
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 5; k++) CHK(hipGraphLaunch(ge, s));
    CHK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 50; k++) CHK(hipGraphLaunch(ge, s));
    CHK(hipStreamSynchronize(s));
    printf("hipGraph of 40 kernel nodes (arg %d B), replayed: %.2f us per node\n", nn, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (50 * 40));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  // (e) re-capture every "step" (kernel arguments may change), hipGraphExecUpdate the instantiated graph, launch
  {
    hipGraph_t g; hipGraphExec_t ge = nullptr;
    double t_cap = 0, t_upd = 0, t_tot = 0;
    const int reps = 60;
    for (int it = 0; it < reps + 5; it++) {
      auto t0 = std::chrono::steady_clock::now();
      CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < 40; k++) { Arg<704> a{}; a.p = d; a.pad[0] = (char)it; hipLaunchKernelGGL(k_touch<704>, dim3(32), dim3(256), 0, s, a); }
      CHK(hipStreamEndCapture(s, &g));
      auto t1 = std::chrono::steady_clock::now();
      if (!ge) CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      else {
        hipGraphNode_t errn; hipGraphExecUpdateResult res;
        hipError_t e = hipGraphExecUpdate(ge, g, &errn, &res);
        if (e != hipSuccess) { printf("hipGraphExecUpdate: %s (result %d)\n", hipGetErrorString(e), (int)res); break; }
      }
      auto t2 = std::chrono::steady_clock::now();
      CHK(hipGraphLaunch(ge, s));
      CHK(hipStreamSynchronize(s));
      auto t3 = std::chrono::steady_clock::now();
      CHK(hipGraphDestroy(g));
      if (it >= 5) { t_cap += std::chrono::duration<double, std::micro>(t1 - t0).count(); t_upd += std::chrono::duration<double, std::micro>(t2 - t1).count(); t_tot += std::chrono::duration<double, std::micro>(t3 - t0).count(); }
    }
    printf("re-capture 40 nodes + hipGraphExecUpdate + launch + sync, per step: capture %.1f us, update %.1f us, total %.1f us (= %.2f us per node)\n", t_cap / reps, t_upd / reps, t_tot / reps, t_tot / reps / 40);
    if (ge) hipGraphExecDestroy(ge);
  }
  // the same 40 dependent launches + sync, directly
  {
    Arg<704> a{}; a.p = d;
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 60; it++) { for (int k = 0; k < 40; k++) hipLaunchKernelGGL(k_touch<704>, dim3(32), dim3(256), 0, s, a); CHK(hipStreamSynchronize(s)); }
    printf("40 direct launches + sync, per step: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 60);
  }
  for (int nb : {32, 128, 512, 1024}) {
    int iters = 2000;
    void *args[] = {&d, &iters};
    hipError_t e = hipLaunchCooperativeKernel((const void *)k_coop, dim3(nb), dim3(256), args, 0, s);
    if (e != hipSuccess) { printf("cooperative launch with %d blocks: %s\n", nb, hipGetErrorString(e)); continue; }
    CHK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CHK(hipLaunchCooperativeKernel((const void *)k_coop, dim3(nb), dim3(256), args, 0, s));
    CHK(hipStreamSynchronize(s));
    printf("cooperative grid.sync(), %4d blocks of 256: %.2f us per barrier\n", nb, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
  }
  for (int nb : {32, 128, 512, 1024}) {
    int iters = 2000;
    CHK(hipMemset(bar, 0, 4096));
    hipLaunchKernelGGL(k_persist, dim3(nb), dim3(256), 0, s, d, bar, 10);
    CHK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_persist, dim3(nb), dim3(256), 0, s, d, bar, iters);
    CHK(hipStreamSynchronize(s));
    printf("hand-written barrier (atomic counter + generation word), %4d blocks of 256: %.2f us per barrier\n", nb, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
  }
  return 0;
}
