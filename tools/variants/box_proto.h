// box_proto.h -- PROTOTYPE (development builds only: -DWCSPH_BOX_PROTO, tools/; not part of the product library).
//
// Third look at LDS staging, with the numbers of tools/micro/gather_rate.hip in hand: a 64-lane gather through the vector
// memory pipeline costs 16 CU-cycles coalesced and ~35 scattered (the developed flow), an LDS read of the same 16 bytes
// 5-10.  Design (the "shared box" the round-2 write-up asked for): one workgroup = 256 consecutive particles; its BOX =
// the cell bounding box of those particles + 2 cells, cut into PIECES = (x, y, z-tile) rows, each one contiguous particle
// range in the tile order; a per-step PLAN (k_box_plan) gives every piece its LDS offset, so a pass stages the box with
// plain coalesced copies (no scans, no searches at pass time) and the neighbour list of a particle holds 16-bit LDS SLOT
// numbers (no deltas to decode).  Here only the two passes of the pressure iteration are converted, the slot lists are
// produced from the product's delta streams by a converter (k_box_convert) and blocks whose box does not fit keep the
// gather walk: enough to measure what the passes would gain, at rest and in the developed flow.
#pragma once
#include "../../wcsph_amd/csrc/device.h"

#define BOX_CAP 2048     /* LDS slots per box */
#define BOX_PIECES 256   /* pieces per box (one plan thread each) */

struct BoxPlan {
  int *hdr;        // [nblocks][8]: ok, npieces, total, x0, y0, zt0, nyr, nzt
  int2 *pieces;    // [nblocks][BOX_PIECES + 1]: (first particle, first LDS slot); entry npieces = (0, total)
  uint4 *nbx;      // slot lists, 8 x 16 bit per block, block-major like the streams
  uint16_t *nxc;   // entries per particle
};

__global__ void __launch_bounds__(BLOCK) k_box_plan(Dev d, BoxPlan bp) {
  const int b = blockIdx.x, tid = threadIdx.x, i = b * BLOCK + tid;
  const int cw = (i < d.nl) ? __float_as_int(d.pos[i].w) : -1;
  __shared__ int mn[3], mx[3];
  if (tid < 3) { mn[tid] = 1 << 30; mx[tid] = -1; }
  __syncthreads();
  if (cw >= 0) {
    const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
    atomicMin(&mn[0], cx); atomicMax(&mx[0], cx); atomicMin(&mn[1], cy); atomicMax(&mx[1], cy); atomicMin(&mn[2], cz); atomicMax(&mx[2], cz);
  }
  __syncthreads();
  int *h = bp.hdr + (size_t)b * 8;
  int2 *pc = bp.pieces + (size_t)b * (BOX_PIECES + 1);
  if (mx[0] < 0) { if (tid == 0) { h[0] = 0; h[1] = 0; h[2] = 0; } return; }
  const int x0 = max(mn[0] - 2, 0), x1 = min(mx[0] + 2, d.nx - 1), y0 = max(mn[1] - 2, 0), y1 = min(mx[1] + 2, d.ny - 1);
  const int za = max(mn[2] - 2, 0), zb = min(mx[2] + 2, d.nz - 1) + 1;
  const int nxr = x1 - x0 + 1, nyr = y1 - y0 + 1, zt0 = za >> T2B, nzt = ((zb - 1) >> T2B) - zt0 + 1;
  const int np = nxr * nyr * nzt;
  bool ok = np <= BOX_PIECES;
  int lo = 0, len = 0;
  if (ok && tid < np) {
    const int rz = tid % nzt, ry = (tid / nzt) % nyr, rx = tid / (nzt * nyr);
    const int x = x0 + rx, y = y0 + ry, zt = zt0 + rz;
    const int zl = max(za, zt << T2B), zh = min(zb, (zt + 1) << T2B);
    lo = d.cs_l[cell_lin(d, x, y, zl)];
    len = d.cs_l[cell_lin(d, x, y, zh - 1) + 1] - lo;
  }
  int tot;
  const int off = block_exclusive_scan(len, &tot);
  ok = ok && tot <= BOX_CAP;
  pc[tid] = make_int2(lo, off);
  if (tid == 0) {
    pc[BOX_PIECES] = make_int2(0, tot);
    h[0] = ok ? 1 : 0; h[1] = np; h[2] = tot; h[3] = x0; h[4] = y0; h[5] = zt0; h[6] = nyr; h[7] = nzt;
  }
}

__device__ __forceinline__ int box_slot(const int *h, const int2 *pc, int j, int cwj) {
  const int cx = cwj & 1023, cy = (cwj >> 10) & 1023, cz = cwj >> 20;
  const int pid = ((cx - h[3]) * h[6] + (cy - h[4])) * h[7] + ((cz >> T2B) - h[5]);
  const int2 a = pc[pid];
  return j - a.x + a.y;
}

// the product's liquid delta stream of every particle -> LDS slot numbers of its block's box
__global__ void k_box_convert(Dev d, BoxPlan bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  const int b = i >> 8;
  const int *h = bp.hdr + (size_t)b * 8;
  if (!h[0]) { bp.nxc[i] = 0; return; }
  const int2 *pc = bp.pieces + (size_t)b * (BOX_PIECES + 1);
  const int cl = d.nhead[i] & 0xffff;
  const size_t st = (size_t)d.nbr_stride;
  const unsigned short *L = reinterpret_cast<const unsigned short *>(d.nbl + i);
  const int own = box_slot(h, pc, i, __float_as_int(d.pos[i].w));
  unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  int n = 0, run = i;
  uint4 *X = bp.nbx + i;
  auto put = [&](int s) {
    w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 = __builtin_amdgcn_alignbit(w2, w1, 16);
    w2 = __builtin_amdgcn_alignbit(w3, w2, 16); w3 = (w3 >> 16) | ((unsigned)s << 16);
    n++;
    if ((n & 7) == 0) X[(size_t)((n >> 3) - 1) * st] = make_uint4(w0, w1, w2, w3);
  };
  for (int k = 0; k < cl; k++) {
    const int dd = (int)(short)L[(size_t)(k >> 3) * st * 8 + (k & 7)];
    run += dd;
    if (!NB_IS_ENTRY(dd)) continue;
    put(box_slot(h, pc, run, __float_as_int(d.pos[run].w)));
  }
  const int ne = n;
  while (n & 7) put(own);
  bp.nxc[i] = (uint16_t)ne;
}

// the solid part of walk() alone
template <class FS>
__device__ __forceinline__ void walk_solid(const Dev &d, int i, float4 pi, FS &&fs) {
  const int cs = (int)((unsigned)d.nhead[i] >> 16);
  if (cs == 0) return;
  const float3 xi = xyz(pi);
  const size_t st = (size_t)d.nbr_stride;
  const uint4 *S = d.nbs + i;
  int run = d.nsbase[i];
  for (int k = 0; k < cs; k += 8) {
    const uint4 e = *(S + (size_t)(k >> 3) * st);
    const int dd[8] = {NB_LO(e.x), NB_HI(e.x), NB_LO(e.y), NB_HI(e.y), NB_LO(e.z), NB_HI(e.z), NB_LO(e.w), NB_HI(e.w)};
#pragma unroll
    for (int u = 0; u < 8; u++) {
      run += dd[u];
      if (k + u < cs && NB_IS_ENTRY(dd[u])) { const float3 r = xi - xyz(nb_ld(d.spos, run)); fs(run, r, dot3(r, r)); }
    }
  }
}

// copies the box into LDS: 8 lanes per piece, `put(slot, particle)` stores one record
template <class PUT>
__device__ __forceinline__ void box_stage(const int2 *sPiece, int np, PUT &&put) {
  for (int p = threadIdx.x >> 3; p < np; p += BLOCK / 8) {
    const int2 a = sPiece[p];
    const int len = sPiece[p + 1].y - a.y;
    for (int t = threadIdx.x & 7; t < len; t += 8) put(a.y + t, a.x + t);
  }
}

#define BOX_SLOTS(e) const unsigned s0 = (e).x & 0xffffu, s1 = (e).x >> 16, s2 = (e).y & 0xffffu, s3 = (e).y >> 16, \
                                    s4 = (e).z & 0xffffu, s5 = (e).z >> 16, s6 = (e).w & 0xffffu, s7 = (e).w >> 16;

// ---- velocity pass (k_vel_from_k<0> of the fused pressure loop) ----
__global__ void __launch_bounds__(BLOCK) k_vel_from_k_box(Dev d, BoxPlan bp, const float4 *pk, float dt, float *kap) {
  if (kap) acc_clear(ACC_SLOT(d, ACC_ERR));
  __shared__ float4 sP[BOX_CAP];
  __shared__ int2 sPiece[BOX_PIECES + 1];
  const int i = sorted_index(d);
  const int b = i >> 8;
  const int *h = bp.hdr + (size_t)b * 8;
  const bool ok = b < (d.nl + BLOCK - 1) / BLOCK && h[0] != 0;   // block-uniform
  if (ok) {
    const int np = h[1];
    const int2 *pc = bp.pieces + (size_t)b * (BOX_PIECES + 1);
    sPiece[threadIdx.x] = pc[threadIdx.x];                      // entries >= np are (0, total): the end of the last piece
    if (threadIdx.x == 0) sPiece[BOX_PIECES] = pc[BOX_PIECES];
    __syncthreads();
    box_stage(sPiece, np, [&](int s, int j) { sP[s] = pk[j]; });
    __syncthreads();
  }
  if (i >= d.nl) return;
  const float4 pi = pk[i];
  const float ki = pi.w;
  if (kap) kap[i] += ki;
  const float3 xi = xyz(pi);
  float3 acc = f3(0, 0, 0);
  const bool ks = fabsf(ki) > d.eps;
  auto fl = [&](float3 r, float r2, float kj) {
    float s = ki + kj;
    if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
  };
  auto fs = [&](int, float3 r, float r2) {
    if (ks) acc = acc + r * (ki * (d.VS0 / d.VL0) * cubic_gradW_scale(d, r2));
  };
  if (ok) {
    const int ne = bp.nxc[i];
    const size_t st = (size_t)d.nbr_stride;
    const uint4 *X = bp.nbx + i;
    uint4 en = ne > 0 ? *X : make_uint4(0, 0, 0, 0);
#define VB_ONE(u) if (k + u < ne) { const float3 r = xi - xyz(p##u); fl(r, dot3(r, r), p##u.w); }
    for (int k = 0; k < ne; k += 8) {
      const uint4 e = en;
      if (k + 8 < ne) en = *(X + (size_t)((k >> 3) + 1) * st);
      BOX_SLOTS(e)
      const float4 p0 = sP[s0], p1 = sP[s1], p2 = sP[s2], p3 = sP[s3], p4 = sP[s4], p5 = sP[s5], p6 = sP[s6], p7 = sP[s7];
      VB_ONE(0) VB_ONE(1) VB_ONE(2) VB_ONE(3) VB_ONE(4) VB_ONE(5) VB_ONE(6) VB_ONE(7)
    }
#undef VB_ONE
    walk_solid(d, i, pi, fs);
  } else {
    walk<8>(d, i, pi, pk, [](int, float4 p) { return p.w; }, [&](int, float3 r, float r2, float kj) { fl(r, r2, kj); }, fs);
  }
  float4 v = d.vel[i];
  const float f = dt * d.VL0;
  v.x += f * acc.x; v.y += f * acc.y; v.z += f * acc.z;
  d.vel[i] = v;
}

// ---- density-change pass (k_drho_press<DRHO_ITER, true>) ----
__global__ void __launch_bounds__(BLOCK) k_drho_press_box(Dev d, BoxPlan bp, float dt) {
  __shared__ float2 sA[BOX_CAP], sB[BOX_CAP], sC[BOX_CAP];   // (px, py) (pz, vx) (vy, vz): three 8-byte reads per neighbour
  __shared__ int2 sPiece[BOX_PIECES + 1];
  const int i = sorted_index(d);
  const bool live = i < d.nl;
  const int b = i >> 8;
  const int *h = bp.hdr + (size_t)b * 8;
  const bool ok = b < (d.nl + BLOCK - 1) / BLOCK && h[0] != 0;
  if (ok) {
    const int np = h[1];
    const int2 *pc = bp.pieces + (size_t)b * (BOX_PIECES + 1);
    sPiece[threadIdx.x] = pc[threadIdx.x];                      // entries >= np are (0, total): the end of the last piece
    if (threadIdx.x == 0) sPiece[BOX_PIECES] = pc[BOX_PIECES];
    __syncthreads();
    box_stage(sPiece, np, [&](int s, int j) {
      const float4 p = d.pk[j], v = d.vel[j];
      sA[s] = make_float2(p.x, p.y); sB[s] = make_float2(p.z, v.x); sC[s] = make_float2(v.y, v.z);
    });
    __syncthreads();
  }
  float e = 0.f;
  if (live) {
    const float4 pi = d.pk[i];
    const float3 xi = xyz(pi), vi = xyz(d.vel[i]);
    float t = 0.f;
    auto fl = [&](float3 r, float r2, float3 vj) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - vj, r); };
    auto fs = [&](int, float3 r, float r2) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi, r); };
    if (ok) {
      const int ne = bp.nxc[i];
      const size_t st = (size_t)d.nbr_stride;
      const uint4 *X = bp.nbx + i;
      uint4 en = ne > 0 ? *X : make_uint4(0, 0, 0, 0);
#define DB_LD(u) const float2 a##u = sA[s##u], b##u = sB[s##u], c##u = sC[s##u];
#define DB_ONE(u) if (k + u < ne) { const float3 r = xi - f3(a##u.x, a##u.y, b##u.x); fl(r, dot3(r, r), f3(b##u.y, c##u.x, c##u.y)); }
      for (int k = 0; k < ne; k += 8) {
        const uint4 e4 = en;
        if (k + 8 < ne) en = *(X + (size_t)((k >> 3) + 1) * st);
        BOX_SLOTS(e4)
        DB_LD(0) DB_LD(1) DB_LD(2) DB_LD(3) DB_LD(4) DB_LD(5) DB_LD(6) DB_LD(7)
        DB_ONE(0) DB_ONE(1) DB_ONE(2) DB_ONE(3) DB_ONE(4) DB_ONE(5) DB_ONE(6) DB_ONE(7)
      }
#undef DB_LD
#undef DB_ONE
      walk_solid(d, i, pi, fs);
    } else {
      walk<8>(d, i, pi, d.pk, [&](int j, float4) { return nb_ld(d.vel, j); }, [&](int, float3 r, float r2, float4 vj) { fl(r, r2, xyz(vj)); }, fs);
    }
    const float a = fmaxf(1.0f, d.rho[i] / d.rho_L0 + dt * t);
    d.adv_rho[i] = a;
    e = a - 1.0f;
    reinterpret_cast<float *>(d.pk + i)[3] = e * d.alpha[i];
  }
  block_accumulate(ACC_SLOT(d, ACC_ERR), e);
}
