// prio_probe.hip -- does a small kernel launched on ANOTHER stream get compute units while a big grid fills the GPU?
//   hipcc --offload-arch=gfx950 -O3 -o prio_probe prio_probe.hip && ./prio_probe
// The halo exchange of the slab step (csrc/api_comm.h) is a kernel of 8-24 workgroups (RCCL: 256 threads, 132 VGPRs, 19.7 KB LDS)
// launched ~10 us after a neighbour pass of ~8000 workgroups has started on the context's stream.  Measured in round 6
// (profiles/r06_overlap.md): it ends when the pass ends, whatever the stream priority.  This probe isolates the question:
//   big   = 8192 workgroups x 256 threads, each ~25 us of dependent loads (a neighbour pass in miniature: 48 VGPRs, no LDS)
//   small = 16 workgroups x 256 threads with 20 KB LDS and ~130 VGPRs, ~10 us alone
// launched `delay` us after big, on (a) a default-priority stream, (b) the highest-priority stream, (c) a stream restricted to a
// CU mask, with big on a stream masked to the complement; reported: small's completion relative to big's start and big's duration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#include <thread>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_big(const int *chain, int *out, int hops) {
  int j = (blockIdx.x * 256 + threadIdx.x) & 0xfffff;
  for (int k = 0; k < hops; k++) j = chain[j];
  if (j == -1) out[0] = j;
}
__global__ void k_spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void __launch_bounds__(256) k_small(const float *in, float *out, int n, long long *stamp) {
  __shared__ float sh[5000];
  if (threadIdx.x == 0) stamp[8 + 2 * blockIdx.x] = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = wall_clock64();
  float acc[24];
  for (int q = 0; q < 24; q++) acc[q] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float v = in[i];
    sh[(threadIdx.x * 19 + i) % 5000] = v;
#pragma unroll
    for (int q = 0; q < 24; q++) acc[q] += v * (float)(q + 1);
  }
  __syncthreads();
  float s = sh[threadIdx.x];
  for (int q = 0; q < 24; q++) s += acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
  if (threadIdx.x == 0) stamp[9 + 2 * blockIdx.x] = wall_clock64();
}
__global__ void k_stamp(long long *stamp, int slot) { if (threadIdx.x == 0) stamp[slot] = wall_clock64(); }
int main(int argc, char **argv) {
  const int N = 1 << 20;
  std::vector<int> h(N);
  const bool friendly = argc > 1 && argv[1][0] == 'f';   // f: the chain stays within a few cache lines (a pass whose gathers hit the L1)
  for (int i = 0; i < N; i++) h[i] = friendly ? (int)((i & ~255) | ((i * 37 + 11) & 255)) : (int)(((long long)i * 7919 + 12345) & (N - 1));
  printf("big kernel: %s chain\n", friendly ? "L1-friendly" : "random (L2-bound)");
  int *chain, *out; float *fin, *fout; long long *stamp;
  CHK(hipMalloc(&chain, N * 4)); CHK(hipMemcpy(chain, h.data(), N * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&fin, 4 << 20)); CHK(hipMemset(fin, 0, 4 << 20)); CHK(hipMalloc(&fout, 1 << 20));
  CHK(hipHostMalloc(&stamp, 64 * 8));
  int lo, hi; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: least %d, greatest %d\n", lo, hi);
  hipStream_t sb, s_def, s_hi, sb_mask, s_mask;
  CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&s_def, hipStreamNonBlocking));
  CHK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
  // CU masks: 256 CUs = 8 words of 32 bits; reserve the first CU of every 32 for the small stream
  uint32_t m_small[8], m_big[8];
  for (int w = 0; w < 8; w++) { m_small[w] = 0x00000001u; m_big[w] = ~m_small[w]; }
  hipError_t em = hipExtStreamCreateWithCUMask(&s_mask, 8, m_small);
  hipError_t em2 = hipExtStreamCreateWithCUMask(&sb_mask, 8, m_big);
  printf("CU-masked streams: %s / %s\n", hipGetErrorString(em), hipGetErrorString(em2));
  hipEvent_t e0, e1, e2, e3;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2)); CHK(hipEventCreate(&e3));
  const int hops = 60;
  auto run = [&](const char *label, hipStream_t big, hipStream_t small, bool with_big) -> int {
    float tb = 0, ts = 0, t_small_end = 0; long long d_start = 0, d_len = 0, first = 0, last = 0, wend = 0;
    const int reps = 10;
    for (int r = 0; r < reps + 2; r++) {
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0, big));
      if (with_big) hipLaunchKernelGGL(k_big, dim3(8192), dim3(256), 0, big, chain, out, hops);
      CHK(hipEventRecord(e1, big));
      // the small kernel follows an event of the big stream recorded BEFORE the big kernel, as the exchange follows the pack
      CHK(hipStreamWaitEvent(small, e0, 0));
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, small, 1500LL);   // 15 us: the big grid has filled the GPU by now
      hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, small, stamp, 2);
      hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, small, fin, fout, 1 << 19, stamp);
      CHK(hipEventRecord(e3, small));
      CHK(hipDeviceSynchronize());
      float a, b;
      CHK(hipEventElapsedTime(&a, e0, e1)); CHK(hipEventElapsedTime(&b, e0, e3));
      if (r >= 2) {
        tb += a; t_small_end += b; d_len += stamp[1] - stamp[0]; d_start += stamp[0] - stamp[2];
        long long f = 1LL << 62, l = 0, we = 0;
        for (int q = 0; q < 16; q++) { f = stamp[8 + 2 * q] < f ? stamp[8 + 2 * q] : f; l = stamp[8 + 2 * q] > l ? stamp[8 + 2 * q] : l; we = stamp[9 + 2 * q] > we ? stamp[9 + 2 * q] : we; }
        first += f - stamp[2]; last += l - stamp[2]; wend += we - stamp[2];
      }
    }
    printf("%-46s big %.1f us; small: first / last workgroup starts %.1f / %.1f us after its stream woke, all done at %.1f us (block 0 runs %.1f us)\n",
           label, tb / reps * 1e3, first / (double)reps / 100.0, last / (double)reps / 100.0, wend / (double)reps / 100.0, d_len / (double)reps / 100.0);
    return 0;
  };
  if (run("small alone, default stream", sb, s_def, false)) return 1;
  if (run("big + small on a default-priority stream", sb, s_def, true)) return 1;
  if (run("big + small on the highest-priority stream", sb, s_hi, true)) return 1;
  if (em == hipSuccess && em2 == hipSuccess) {
    if (run("big on 248 CUs + small on 8 reserved CUs", sb_mask, s_mask, true)) return 1;
    if (run("big on ALL CUs + small on a masked stream", sb, s_mask, true)) return 1;
  }
  return 0;
}
