// ablate.h -- MEASUREMENT ONLY (wcsph_time_kernel groups "ablate_drho:<n>" / "ablate_vel:<n>", tools/ablate.py):
// the two neighbour passes of pressure_iter with one ingredient removed at a time, to tell which resource a pass
// waits for.  The results of these kernels are meaningless and are written to scratch fields (adv_rho is
// rewritten by the next real pass; d_vel is scratch between steps).
//   0 full replica            1 indices loaded but every gather reads the lane's own record (index stream + math)
//   2 no index loads, gathers read consecutive records (math + coalesced gathers)
//   3 full memory traffic, no kernel math (sum of the gathered components)
//   4 like 0 but the payload gather (vel_j) is dropped (one gather per entry)
//   6 = 0 + hop-slot test   7 = 6 + block reduction and one double atomic per block   8 = 6 + ownership test (oid)
//   5 the two gathers of an entry read the two halves of ONE 32-byte aligned record (emulates interleaved pos|vel)
// (hop slots are treated as entries here: timing only)
// Measured on the 32-bit index lists this replaced (C3, faithful, us): vel 79 -> own-record 70, no-index 38, no-math 77;
// drho 127 -> own-record 110, no-index 68, no-math 122, one gather 84, interleaved record 110; half the index bytes:
// vel 43, drho 79 -- the index stream was the bound, hence the 16-bit delta streams.
#pragma once
#include "device.h"

template <int ABL, bool TWO, class FL>
__device__ __forceinline__ void walk_abl(const Dev &d, int i, float4 pi, const float4 *P, const float4 *Q, FL &&fl) {
  const float3 xi = xyz(pi);
  const size_t st = (size_t)d.nbr_stride;
  const int2 head = d.nhead[i];
  const int cl = head.x & 0xffff, cs = (int)((unsigned)head.x >> 16);
  const uint4 *L = d.nbl + i;
  const int top = d.nl - 1;
  int run = i;
  for (int k = 0; k < cl; k += 8) {
    int j0, j1, j2, j3, j4, j5, j6, j7, dd0 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0, dd5 = 0, dd6 = 0, dd7 = 0;
    if (ABL == 2) {
      j0 = min(i + k + 1, top); j1 = min(i + k + 2, top); j2 = min(i + k + 3, top); j3 = min(i + k + 4, top);
      j4 = min(i + k + 5, top); j5 = min(i + k + 6, top); j6 = min(i + k + 7, top); j7 = min(i + k + 8, top);
    } else {
      const uint4 e = L[(size_t)(k >> 3) * st];
      dd0 = NB_LO(e.x); dd1 = NB_HI(e.x); dd2 = NB_LO(e.y); dd3 = NB_HI(e.y); dd4 = NB_LO(e.z); dd5 = NB_HI(e.z); dd6 = NB_LO(e.w); dd7 = NB_HI(e.w);
      j0 = run + dd0; j1 = j0 + dd1; j2 = j1 + dd2; j3 = j2 + dd3; j4 = j3 + dd4; j5 = j4 + dd5; j6 = j5 + dd6; j7 = j6 + dd7;
      run = j7;
      if (ABL == 1) {   // keep a data dependence on the loaded slots without following them
        j0 = i + (j0 >> 31); j1 = i + (j1 >> 31); j2 = i + (j2 >> 31); j3 = i + (j3 >> 31);
        j4 = i + (j4 >> 31); j5 = i + (j5 >> 31); j6 = i + (j6 >> 31); j7 = i + (j7 >> 31);
      }
    }
    float4 p0, p1, p2, p3, p4, p5, p6, p7, b0, b1, b2, b3, b4, b5, b6, b7;
    if (TWO && ABL == 5) {
      p0 = P[j0 & ~1]; p1 = P[j1 & ~1]; p2 = P[j2 & ~1]; p3 = P[j3 & ~1]; p4 = P[j4 & ~1]; p5 = P[j5 & ~1]; p6 = P[j6 & ~1]; p7 = P[j7 & ~1];
      b0 = P[j0 | 1]; b1 = P[j1 | 1]; b2 = P[j2 | 1]; b3 = P[j3 | 1]; b4 = P[j4 | 1]; b5 = P[j5 | 1]; b6 = P[j6 | 1]; b7 = P[j7 | 1];
    } else {
      p0 = P[j0]; p1 = P[j1]; p2 = P[j2]; p3 = P[j3];
      b0 = p0; b1 = p1; b2 = p2; b3 = p3;
      if (TWO && ABL != 4) { b0 = Q[j0]; b1 = Q[j1]; b2 = Q[j2]; b3 = Q[j3]; }
      p4 = P[j4]; p5 = P[j5]; p6 = P[j6]; p7 = P[j7];
      b4 = p4; b5 = p5; b6 = p6; b7 = p7;
      if (TWO && ABL != 4) { b4 = Q[j4]; b5 = Q[j5]; b6 = Q[j6]; b7 = Q[j7]; }
    }
#define ABL_ONE(u, pp, bb) if (k + u < cl && (ABL < 6 || NB_IS_ENTRY(dd##u))) { const float3 r = xi - xyz(pp); fl(r, dot3(r, r), bb); }
    ABL_ONE(0, p0, b0) ABL_ONE(1, p1, b1) ABL_ONE(2, p2, b2) ABL_ONE(3, p3, b3)
    ABL_ONE(4, p4, b4) ABL_ONE(5, p5, b5) ABL_ONE(6, p6, b6) ABL_ONE(7, p7, b7)
  }
  const uint4 *S = d.nbs + i;
  run = head.y;
  for (int k = 0; k < cs; k += 8) {
    int j0 = 0, j1 = 1, j2 = 2, j3 = 3, j4 = 4, j5 = 5, j6 = 6, j7 = 7;
    if (ABL != 2) {
      const uint4 e = S[(size_t)(k >> 3) * st];
      j0 = run + NB_LO(e.x); j1 = j0 + NB_HI(e.x); j2 = j1 + NB_LO(e.y); j3 = j2 + NB_HI(e.y);
      j4 = j3 + NB_LO(e.z); j5 = j4 + NB_HI(e.z); j6 = j5 + NB_LO(e.w); j7 = j6 + NB_HI(e.w);
      run = j7;
      if (ABL == 1) { j0 = -(j0 >> 31); j1 = -(j1 >> 31); j2 = -(j2 >> 31); j3 = -(j3 >> 31); j4 = -(j4 >> 31); j5 = -(j5 >> 31); j6 = -(j6 >> 31); j7 = -(j7 >> 31); }
    }
    const float4 q0 = d.spos[j0], q1 = d.spos[j1], q2 = d.spos[j2], q3 = d.spos[j3], q4 = d.spos[j4], q5 = d.spos[j5], q6 = d.spos[j6], q7 = d.spos[j7];
    const float4 z = make_float4(0, 0, 0, 0);
#undef ABL_ONE
#define ABL_ONE(u, pp, bb) if (k + u < cs) { const float3 r = xi - xyz(pp); fl(r, dot3(r, r), bb); }
    ABL_ONE(0, q0, z) ABL_ONE(1, q1, z) ABL_ONE(2, q2, z) ABL_ONE(3, q3, z) ABL_ONE(4, q4, z) ABL_ONE(5, q5, z) ABL_ONE(6, q6, z) ABL_ONE(7, q7, z)
#undef ABL_ONE
  }
}

template <int ABL>
__global__ void __launch_bounds__(BLOCK) kx_drho(Dev d, float dt, int i0 = 0, int n = 1 << 30) {
  const int ii = sorted_index();
  const int i = ii + i0;
  if (ii >= n || i >= d.nl) return;
  const float4 pi = d.pos[i];
  const float3 vi = xyz(d.vel[i]);
  float t = 0.f;
  walk_abl<ABL, true>(d, i, pi, d.pos, d.vel, [&](float3 r, float r2, float4 vj) {
    if (ABL == 3) t += r.x + r.y + r.z + vj.x + vj.y + vj.z;
    else t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(vj), r);
  });
  const float a = fmaxf(1.0f, d.rho[i] / d.rho_L0 + dt * t);
  if (ABL == 8 && d.oid[i] >= d.n_owned) return;
  d.adv_rho[i] = a;
  if (ABL == 7) block_accumulate(ACC_SLOT(d, ACC_ERR), a - 1.0f);
  // striped accumulators: 64 targets 8 B / 256 B / 4 KB apart (d.kbuf is scratch here)
  if (ABL == 9) block_accumulate(reinterpret_cast<double *>(d.kbuf) + (blockIdx.x & 63), a - 1.0f);
  if (ABL == 10) block_accumulate(reinterpret_cast<double *>(d.kbuf) + (blockIdx.x & 63) * 32, a - 1.0f);
  if (ABL == 11) block_accumulate(reinterpret_cast<double *>(d.kbuf) + (blockIdx.x & 63) * 512, a - 1.0f);
  if (ABL == 12) {   // no atomics: one plain store per block
    __shared__ float sh2[BLOCK / WAVE];
    float v = wave_sum(a - 1.0f);
    if ((threadIdx.x & (WAVE - 1)) == 0) sh2[threadIdx.x / WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) d.kbuf[blockIdx.x] = sh2[0] + sh2[1] + sh2[2] + sh2[3];
  }
}

template <int ABL>
__global__ void __launch_bounds__(BLOCK) kx_vel(Dev d, float dt, int i0 = 0, int n = 1 << 30) {
  const int ii = sorted_index();
  const int i = ii + i0;
  if (ii >= n || i >= d.nl) return;
  const float4 pi = d.pk[i];
  const float ki = pi.w;
  float3 acc = f3(0, 0, 0);
  walk_abl<ABL, false>(d, i, pi, d.pk, d.pk, [&](float3 r, float r2, float4 b) {
    if (ABL == 3) acc = acc + r * b.w;
    else {
      const float s = ki + b.w;
      if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
    }
  });
  d.d_vel[i] = make_float4(acc.x * dt, acc.y * dt, acc.z * dt, 0);
}
