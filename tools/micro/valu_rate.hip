// VALU issue-rate probe for gfx950: wave-instructions per second for v_fma_f32 and v_pk_fma_f32 at full occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = {a1, a0}, b5 = {a3, a2}, b6 = {a5, a4}, b7 = {a7, a6};
  v2f sv = {s, s};
  for (int i = 0; i < iters; i++) {
    if (PK) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        b0 = __builtin_elementwise_fma(b0, sv, sv); b1 = __builtin_elementwise_fma(b1, sv, sv); b2 = __builtin_elementwise_fma(b2, sv, sv); b3 = __builtin_elementwise_fma(b3, sv, sv);
        b4 = __builtin_elementwise_fma(b4, sv, sv); b5 = __builtin_elementwise_fma(b5, sv, sv); b6 = __builtin_elementwise_fma(b6, sv, sv); b7 = __builtin_elementwise_fma(b7, sv, sv);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = fmaf(a0, s, s); a1 = fmaf(a1, s, s); a2 = fmaf(a2, s, s); a3 = fmaf(a3, s, s);
        a4 = fmaf(a4, s, s); a5 = fmaf(a5, s, s); a6 = fmaf(a6, s, s); a7 = fmaf(a7, s, s);
      }
    }
  }
  float r = PK ? (b0.x + b1.x + b2.x + b3.x + b4.y + b5.y + b6.y + b7.y) : (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  if (r == 12345.678f) out[0] = r;
}
int main() {
  float *d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 256 * 8;
  for (int pk = 0; pk < 2; pk++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double winst = (double)blocks * 4 * iters * 64;   // waves * instr
      if (rep) printf("%s: %.1f G wave-instr/s (%.1f TFLOP/s), %.3f ms\n", pk ? "v_pk_fma_f32" : "v_fma_f32", winst / ms / 1e6, winst * 64 * (pk ? 4 : 2) / ms / 1e9, ms);
    }
  }
  return 0;
}
