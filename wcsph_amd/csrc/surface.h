// surface.h -- ParticleData.compute_color_map / cal_anistropic_kernel (ref: ParticleData.py:187-298), the
// per-particle half of the Yu & Turk surface reconstruction.  Post-processing, not on the
// simulation hot path: these kernels re-walk the cells (walk_cells) with their own radius instead of using the
// compact per-step lists, and their fields are allocated on first use.
#pragma once
#include "device.h"

struct SurfDev { float *color; float4 *color_grad, *pos_avr, *g0, *g1, *g2; float R2; };

// ref: ParticleData.py:188-201
__global__ void __launch_bounds__(BLOCK) k_surf_color(Dev d, SurfDev s) {
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  const float4 pi = d.pos[i];
  float cl = 0.f, cs = 0.f;
  walk_cells(d, i, pi, true, [&](int j, float3, float r2, float m) { cl += m * d.mass / d.rho[j] * cubic_W(d, r2); },
             [&](int, float3, float r2, float m) { cs += m * cubic_W(d, r2); });
  s.color[i] = d.mass / d.rho[i] * d.mk_h3 + cl + d.VS0 * cs;
}
// ref: ParticleData.py:203-215
__global__ void __launch_bounds__(BLOCK) k_surf_color_grad(Dev d, SurfDev s) {
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  const float4 pi = d.pos[i];
  float3 g = f3(0, 0, 0);
  walk_cells(d, i, pi, true,
             [&](int j, float3 r, float r2, float m) { g = g + r * (m * d.mass / d.rho[j] * s.color[j] * cubic_gradW_scale(d, r2)); },
             [&](int, float3, float, float) {});
  s.color_grad[i] = f4(g * (1.0f / s.color[i]), 0);
}
// ref: ParticleData.py:291-298 weight_func, support 2 * mc_grid.searchR
__device__ __forceinline__ float aniso_w(float r2, float R2) {
  const float q = sqrtf(r2) / R2;
  return (q < 1.0f) ? 1.0f - q * q * q : 0.0f;
}
// ref: ParticleData.py:222-241  (d.list_r2 = R2^2 for this launch)
__global__ void __launch_bounds__(BLOCK) k_surf_pos_avr(Dev d, SurfDev s) {
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  const float4 pi = d.pos[i];
  float sw = 0.f;
  float3 sx = f3(0, 0, 0);
  walk_cells(d, i, pi, true,
             [&](int j, float3, float r2, float m) { const float w = m * aniso_w(r2, s.R2); sw += w; sx = sx + xyz(d.pos[j]) * w; },
             [&](int, float3, float, float) {});
  s.pos_avr[i] = (sw > 0.0f) ? f4(sx * (1.0f / sw), 0) : make_float4(pi.x, pi.y, pi.z, 0);
}
// symmetric 3x3 eigen-decomposition, cyclic Jacobi (the reference's ti.svd of the covariance: U = V = Q for w > 0)
__device__ __forceinline__ void jacobi3(float a00, float a01, float a02, float a11, float a12, float a22, float w[3], float Q[3][3]) {
  float A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  float V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll 1
  for (int sweep = 0; sweep < 8; sweep++) {
#pragma unroll
    for (int pq = 0; pq < 3; pq++) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
      const float apq = A[p][q];
      if (fabsf(apq) > 1e-30f) {
        const float th = (A[q][q] - A[p][p]) / (2.0f * apq);
        const float t = copysignf(1.0f, th) / (fabsf(th) + sqrtf(th * th + 1.0f));
        const float c = rsqrtf(t * t + 1.0f), sn = t * c;
#pragma unroll
        for (int k = 0; k < 3; k++) { const float x = A[k][p], y = A[k][q]; A[k][p] = c * x - sn * y; A[k][q] = sn * x + c * y; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const float x = A[p][k], y = A[q][k]; A[p][k] = c * x - sn * y; A[q][k] = sn * x + c * y; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const float x = V[k][p], y = V[k][q]; V[k][p] = c * x - sn * y; V[k][q] = sn * x + c * y; }
      }
    }
  }
  float e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  float v0[3] = {V[0][0], V[1][0], V[2][0]}, v1[3] = {V[0][1], V[1][1], V[2][1]}, v2[3] = {V[0][2], V[1][2], V[2][2]};
#define SWAPC(ea, va, eb, vb) if (eb > ea) { float t_ = ea; ea = eb; eb = t_; for (int k_ = 0; k_ < 3; k_++) { float u_ = va[k_]; va[k_] = vb[k_]; vb[k_] = u_; } }
  SWAPC(e0, v0, e1, v1) SWAPC(e0, v0, e2, v2) SWAPC(e1, v1, e2, v2)
  w[0] = e0; w[1] = e1; w[2] = e2;
  for (int k = 0; k < 3; k++) { Q[k][0] = v0[k]; Q[k][1] = v1[k]; Q[k][2] = v2[k]; }
}
// ref: ParticleData.py:268-280: C = sum_ci / sum_wij -> eigen-decomposition -> G = R inv(sigmas) R^T
__device__ __forceinline__ void aniso_G(float sw, float c00, float c01, float c02, float c11, float c12, float c22, float4 &r0, float4 &r1, float4 &r2v) {
  const float kr = 4.0f, ks = 1400.0f;
  const float is = 1.0f / sw;
  float w[3], Q[3][3];
  jacobi3(c00 * is, c01 * is, c02 * is, c11 * is, c12 * is, c22 * is, w, Q);
  if (w[0] > 0.0f) {
    const float i0 = 1.0f / (ks * w[0]), i1 = 1.0f / (ks * fmaxf(w[1], w[0] / kr)), i2 = 1.0f / (ks * fmaxf(w[2], w[0] / kr));
    float G[3][3];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) G[a][b] = Q[a][0] * i0 * Q[b][0] + Q[a][1] * i1 * Q[b][1] + Q[a][2] * i2 * Q[b][2];
    r0 = make_float4(G[0][0], G[0][1], G[0][2], 0); r1 = make_float4(G[1][0], G[1][1], G[1][2], 0); r2v = make_float4(G[2][0], G[2][1], G[2][2], 0);
  }
}
// ref: ParticleData.py:244-280
__global__ void __launch_bounds__(BLOCK) k_surf_aniso(Dev d, SurfDev s) {
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  const float kn = 0.5f, ne = 25.0f;
  float4 r0 = make_float4(kn, 0, 0, 0), r1 = make_float4(0, kn, 0, 0), r2v = make_float4(0, 0, kn, 0);
  if ((float)d.ncount[i] > ne) {
    const float4 pi = d.pos[i];
    const float3 pa = xyz(s.pos_avr[i]);
    float sw = 0.f, c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    walk_cells(d, i, pi, true,
               [&](int j, float3, float r2, float m) {
                 const float w = m * aniso_w(r2, s.R2);
                 const float3 r = xyz(d.pos[j]) - pa;
                 sw += w;
                 c00 += w * r.x * r.x; c01 += w * r.x * r.y; c02 += w * r.x * r.z;
                 c11 += w * r.y * r.y; c12 += w * r.y * r.z; c22 += w * r.z * r.z;
               },
               [&](int, float3, float, float) {});
    aniso_G(sw, c00, c01, c02, c11, c12, c22, r0, r1, r2v);
  }
  s.g0[i] = r0; s.g1[i] = r1; s.g2[i] = r2v;
}
