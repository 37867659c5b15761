// L1 / LDS gather-rate probe for gfx950: CU cycles per 64-lane gather instruction by load width and address pattern,
// all addresses inside one 8 KB region (always L1 hits).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/gather_rate tools/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v3 __attribute__((ext_vector_type(3)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int W> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef v2 T; };
template <> struct Vec<3> { typedef v3 T; };
template <> struct Vec<4> { typedef v4 T; };
template <int W> __device__ inline float first(typename Vec<W>::T v) { if constexpr (W == 1) return v; else return v.x; }
// pattern 0: lane l reads record l + c (contiguous, shifted); 1: brick -- 4 rows of 16 lanes, rows 40 records apart;
// 2: pseudo-random records within the region; 3: every lane its own record (c = 0, aligned)
__device__ inline int pat(int P, int lane, int it, int nrec) {
  int r;
  if (P == 0) r = lane + 3 + it * 5;
  else if (P == 1) r = (lane & 15) + (lane >> 4) * 40 + 3 + it * 5;
  else if (P == 2) r = (lane * 37 + it * 11) ^ ((lane >> 2) * 13);
  else r = lane;
  return r & (nrec - 1);
}
template <int W, int STRIDE, int P, bool LDS, int ACTIVE = 64>
__global__ void __launch_bounds__(256) k(const char *buf, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) char sh[8192];
  const int nrec = 8192 / STRIDE >= 512 ? 512 : 256;   // records in the region (power of two)
  if (LDS) { for (int t = threadIdx.x; t < 2048; t += 256) ((float *)sh)[t] = ((const float *)buf)[t]; __syncthreads(); }
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  typedef typename Vec<W>::T T;
  // ACTIVE < 64: only lanes whose (lane * 37 & 63) < ACTIVE run the loop (scattered over the wave, like short lists among long ones);
  // ACTIVE < 0: the first -ACTIVE lanes (a contiguous prefix)
  if (ACTIVE > 0 ? ((lane * 37) & 63) >= ACTIVE : lane >= -ACTIVE) iters = 0;
  for (int it = 0; it < iters; it++) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int r = pat(P, lane, it * 8 + u, nrec);
      if (LDS) v[u] = *reinterpret_cast<const T *>(sh + r * STRIDE);
      else v[u] = *reinterpret_cast<const T *>(buf + (unsigned)r * STRIDE);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += first<W>(v[u]);
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int W, int STRIDE, int P, bool LDS, int ACTIVE = 64>
void run(const char *name, const char *buf, float *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 8;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, STRIDE, P, LDS, ACTIVE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%-44s %6.2f CU-cycles per wave-gather (2.4 GHz), %.3f ms\n", name, ms * 1e-3 * 2.4e9 / instr_per_cu, ms);
}
int main() {
  char *buf; float *out; hipMalloc(&buf, 16384); hipMemset(buf, 0, 16384); hipMalloc(&out, 4);
#define R(W, S, P, L, n) run<W, S, P, L>(n, buf, out)
  R(1, 4, 0, false, "L1 x1 stride 4  contiguous");
  R(2, 8, 0, false, "L1 x2 stride 8  contiguous");
  R(3, 12, 0, false, "L1 x3 stride 12 contiguous");
  R(3, 16, 0, false, "L1 x3 stride 16 contiguous");
  R(4, 16, 0, false, "L1 x4 stride 16 contiguous");
  R(4, 16, 3, false, "L1 x4 stride 16 own record (aligned)");
  R(1, 4, 1, false, "L1 x1 stride 4  brick");
  R(2, 8, 1, false, "L1 x2 stride 8  brick");
  R(3, 12, 1, false, "L1 x3 stride 12 brick");
  R(3, 16, 1, false, "L1 x3 stride 16 brick");
  R(4, 16, 1, false, "L1 x4 stride 16 brick");
  R(1, 4, 2, false, "L1 x1 stride 4  scattered");
  R(3, 16, 2, false, "L1 x3 stride 16 scattered");
  R(4, 16, 2, false, "L1 x4 stride 16 scattered");
#define RA(W, S, P, L, A, n) run<W, S, P, L, A>(n, buf, out)
  RA(4, 16, 1, false, 48, "L1 x4 brick, 48 lanes active (scattered)");
  RA(4, 16, 1, false, 32, "L1 x4 brick, 32 lanes active (scattered)");
  RA(4, 16, 1, false, 16, "L1 x4 brick, 16 lanes active (scattered)");
  RA(4, 16, 1, false, 4, "L1 x4 brick,  4 lanes active (scattered)");
  RA(4, 16, 1, false, -32, "L1 x4 brick, lanes 0..31 active");
  RA(4, 16, 1, false, -16, "L1 x4 brick, lanes 0..15 active");
  RA(4, 16, 2, false, 32, "L1 x4 scattered, 32 lanes active (scattered)");
  RA(4, 16, 2, false, 16, "L1 x4 scattered, 16 lanes active (scattered)");
  RA(4, 16, 1, true, 32, "LDS b128 brick, 32 lanes active (scattered)");
  R(1, 4, 0, true, "LDS b32  contiguous");
  R(2, 8, 0, true, "LDS b64  contiguous");
  R(3, 16, 0, true, "LDS b96  stride 16 contiguous");
  R(4, 16, 0, true, "LDS b128 contiguous");
  R(3, 16, 1, true, "LDS b96  stride 16 brick");
  R(4, 16, 1, true, "LDS b128 brick");
  R(3, 16, 2, true, "LDS b96  stride 16 scattered");
  R(4, 16, 2, true, "LDS b128 scattered");
  R(1, 4, 2, true, "LDS b32  scattered");
  return 0;
}
