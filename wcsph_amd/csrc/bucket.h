// bucket.h -- the reference's hash table itself (HashGrid.grid / gridCount, ref: HashGrid.py:20-21,67-76), built on
// demand in faithful mode for the few consumers that need the RAW neighbour row rather than its in-support part:
//   * HashGrid.neighbor rows exactly as the reference stores them, far aliased entries included (HashGrid.py:88-106);
//   * the literal compute_tension loop (dfsph.py:264-305): `normal *= searchR` once per raw entry (Q8) and the curvature
//     term over every raw liquid entry;
//   * ParticleData.cal_anistropic_kernel in faithful mode (support 0.18 > h: a far aliased entry can weigh in).
// The solver passes never touch it: every kernel-weighted sum has support h <= the walk radius.
//
// Layout: bk_start[T + 1] (exclusive scan of gridCount), bk_item[] = global particle ids (original liquid index, or
// n_liquid + original solid index), ascending inside a bucket.  The reference's slot order is the arrival order of its
// atomic_add (Q16: schedule dependent); ascending ids is the order a serial insert produces (one valid reference run).
#pragma once
#include "device.h"
#include "grid.h"
#include "surface.h"

struct BucketTab {
  int *start;     // [T + 1]
  int *item;      // [n_liquid + n_solid]
  int *cursor;    // [T] scatter cursors
  int *swhere;    // [n_solid] original solid index -> sorted solid index
  // derived per step for the kernels that walk whole rows (compute_tension literally): the items of every cell's bucket
  // once more, as SORTED indices (bit 31: boundary particle), laid out in plain (x, y, z) CELL order -- the five cells of
  // one (x, y) column of a row are then ONE contiguous run [coff[c(za)], coff[c(zb) + 1]), neighbouring lanes read
  // neighbouring runs, and a row walk is 25 runs + one record gather per entry instead of 125 hashes, 125 slices
  // scattered all over the table by the hash, and two dependent look-ups per entry
  int *coff;      // [ncell + 1] exclusive offsets into citem; cells further than 2 from the liquid's box are empty
  int *citem;     // [citem_cap]: sum over the in-box cells of their bucket lengths -- about (cells within 2 of the liquid's box) x (particles per
                  // bucket), i.e. the CELL count of the grown box once that box is the whole domain; sized from coff[ncell] before it is filled (api_grid.h)
  int citem_cap;
};
#define BK_SOLID 0x80000000u

__global__ void k_bk_swhere(Dev d, BucketTab b) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < d.ns) b.swhere[__float_as_int(d.spos[s].w)] = s;
}
__global__ void k_bk_scatter_liquid(Dev d, BucketTab b) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  const int w = __float_as_int(d.pos[i].w);
  if (w < 0) return;   // outside the box: not inserted (HashGrid.py:68-69)
  const int h = ref_hash((w & 1023) + d.x0, (w >> 10) & 1023, w >> 20, d.table_size);
  b.item[atomicAdd(&b.cursor[h], 1)] = d.oid[i];
}
__global__ void k_bk_scatter_solid(Dev d, BucketTab b) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.ns) return;
  const float4 p = d.spos[s];
  int lin;
  const int w = cell_packed(d, xyz(p), &lin);
  if (w < 0) return;
  const int h = ref_hash((w & 1023) + d.x0, (w >> 10) & 1023, w >> 20, d.table_size);
  b.item[atomicAdd(&b.cursor[h], 1)] = d.nl + __float_as_int(p.w);
}
__global__ void k_bk_sort(Dev d, BucketTab b) {   // one thread per bucket: insertion sort (<= maxInGrid items, ~1 on average)
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= d.table_size) return;
  const int lo = b.start[h], hi = b.start[h + 1];
  for (int a = lo + 1; a < hi; a++) {
    const int v = b.item[a];
    int k = a - 1;
    while (k >= lo && b.item[k] > v) { b.item[k + 1] = b.item[k]; k--; }
    b.item[k + 1] = v;
  }
}

// coff (before its scan): the length of cell c's bucket, for the cells within 2 of the liquid's box (plain cell index)
__device__ __forceinline__ int bk_cell_bucket(const Dev &d, int c, bool *inbox) {
  const int *bb = d.bbox;
  const int cz = c % d.nz, cy = (c / d.nz) % d.ny, cx = c / (d.nz * d.ny);
  *inbox = bb[3] >= 0 && !(cx < bb[0] - 2 || cx > bb[3] + 2 || cy < bb[1] - 2 || cy > bb[4] + 2 || cz < bb[2] - 2 || cz > bb[5] + 2);
  return ref_hash(cx + d.x0, cy, cz, d.table_size);
}
__global__ void k_bk_cellcount(Dev d, BucketTab b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > d.ncell) return;
  int n = 0;
  if (c < d.ncell) {
    bool in;
    const int h = bk_cell_bucket(d, c, &in);
    if (in) n = b.start[h + 1] - b.start[h];
  }
  b.coff[c] = n;
}
// after the scan: copy every cell's bucket behind its offset, as sorted indices
__global__ void k_bk_cellfill(Dev d, BucketTab b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.ncell) return;
  bool in;
  const int h = bk_cell_bucket(d, c, &in);
  if (!in) return;
  const int lo = b.start[h], n = b.start[h + 1] - lo, o = b.coff[c];
  if (n > 0 && o + n > b.citem_cap) { atomicOr(&d.flags[FLAG_ERR], ERRBIT_ROWITEMS); return; }
  for (int k = 0; k < n; k++) {
    const int g = b.item[lo + k];
    b.citem[o + k] = g < d.nl ? d.where[g] : (int)((unsigned)b.swhere[g - d.nl] | BK_SOLID);
  }
}
// the raw row of liquid particle i through the cell-ordered item runs, in the reference's order (REV: in exactly the
// reverse order): use(js, ld(js)) for every entry but the particle itself, js = sorted index | BK_SOLID; use returns
// false to stop.  Per (x, y) column: one run, walked B entries at a time -- B item loads in flight, then B record loads
// ld(js), then the B bodies in order.  Loads past the end of a run re-read its first item and are discarded: no branch
// between the loads of a chunk.
template <bool REV, int B, class LD, class F>
__device__ __forceinline__ void raw_row_s(const Dev &d, const BucketTab &b, int i, float4 pi, LD &&ld, F &&use) {
  const int cw = __float_as_int(pi.w);
  if (cw < 0) return;
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const int za = max(cz - 2, 0), zb = min(cz + 2, d.nz - 1);
  auto column = [&](int u, int &lo, int &hi) {
    const int q = REV ? 24 - u : u;
    const int x = cx + q / 5 - 2, y = cy + q % 5 - 2;
    const bool in = u < 25 && (unsigned)x < (unsigned)d.nx && (unsigned)y < (unsigned)d.ny;
    const int *col = b.coff + (size_t)((in ? x : cx) * d.ny + (in ? y : cy)) * d.nz;   // outside the grid: any valid address, emptied below
    lo = col[za]; hi = col[zb + 1];
    if (!in) hi = lo;
  };
  // software pipeline over the columns: while column u's records are in flight, the items of column u + 1 and the run
  // bounds of column u + 2 are requested -- one exposed memory latency per column instead of three
  auto items = [&](int lo, int hi, int e0, int (&js)[B]) {
    const int tot = hi - lo;
#pragma unroll
    for (int k = 0; k < B; k++) {
      const int e = e0 + k;
      const int ec = e < tot ? e : (e0 < tot ? e0 : 0);          // past the end: a valid address (the run's first item, or
      const int idx = tot > 0 ? (REV ? hi - 1 - ec : lo + ec) : 0;   // the table's first when the run is empty), discarded
      const int v = b.citem[idx];
      js[k] = e < tot ? v : i;   // i: skipped like the particle itself
    }
  };
  int lo, hi, nlo, nhi;
  column(0, lo, hi);
  column(1, nlo, nhi);
  int js[B], jn[B];
  items(lo, hi, 0, js);
#pragma unroll 1
  for (int u = 0; u < 25; u++) {
    int n2lo, n2hi;
    column(u + 2, n2lo, n2hi);
    items(nlo, nhi, 0, jn);
    const int tot = hi - lo;
#pragma unroll 1
    for (int e0 = 0; e0 < tot; e0 += B) {
      if (e0 > 0) items(lo, hi, e0, js);   // a run longer than one chunk (crowded cells): the rest is not pipelined
      decltype(ld(0)) rec[B];
#pragma unroll
      for (int k = 0; k < B; k++) rec[k] = ld(js[k]);   // unconditional (i is a valid index)
#pragma unroll
      for (int k = 0; k < B; k++) nb_pin(rec[k]);       // keep LLVM from sinking a load into the body that consumes it
#pragma unroll
      for (int k = 0; k < B; k++) if (js[k] != i && !use(js[k], rec[k])) return;
    }
    lo = nlo; hi = nhi; nlo = n2lo; nhi = n2hi;
#pragma unroll
    for (int k = 0; k < B; k++) js[k] = jn[k];
  }
}

// position (and kind) of a global id
__device__ __forceinline__ float4 bk_pos(const Dev &d, const BucketTab &b, int g, int *sorted) {
  if (g < d.nl) { const int j = d.where[g]; *sorted = j; return d.pos[j]; }
  const int s = b.swhere[g - d.nl];
  *sorted = s;
  return d.spos[s];
}

// the raw row of liquid particle i (sorted index): f(g) for every entry in the reference's order -- cells (m, n, q) with
// x outermost (HashGrid.py:82-85), bucket slots in ascending id, the particle itself skipped (HashGrid.py:97)
template <class F>
__device__ __forceinline__ void raw_row(const Dev &d, const BucketTab &b, int i, float4 pi, F &&f) {
  const int cw = __float_as_int(pi.w);
  if (cw < 0) return;
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const int self = d.oid[i];
#pragma unroll 1
  for (int k = 0; k < 125; k++) {
    const int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
    if ((unsigned)x >= (unsigned)d.nx || (unsigned)y >= (unsigned)d.ny || (unsigned)z >= (unsigned)d.nz) continue;
    const int h = ref_hash(x + d.x0, y, z, d.table_size);
    for (int s = b.start[h]; s < b.start[h + 1]; s++) {
      const int g = b.item[s];
      if (g != self) f(g);
    }
  }
}

// HashGrid.neighbor rows (ref: HashGrid.py:88-106) for original particles [i0, i0 + n): counts = neighborCount, lists -1 padded
__global__ void k_bk_rows(Dev d, BucketTab b, int i0, int n, int cap, int *counts, int *lists) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = d.where[i0 + t];
  int *row = lists + (size_t)t * cap;
  int w = 0;
  raw_row(d, b, i, d.pos[i], [&](int g) { if (w < cap) row[w] = g; w++; });
  counts[t] = w;
  for (int k = w; k < cap; k++) row[k] = -1;
}
// HashGrid.grid rows (ref: HashGrid.py:21,76) for buckets [h0, h0 + n)
__global__ void k_bk_grid_rows(Dev d, BucketTab b, int h0, int n, int cap, int *rows) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int lo = b.start[h0 + t], cnt = b.start[h0 + t + 1] - lo;
  for (int k = 0; k < cap; k++) rows[(size_t)t * cap + k] = k < cnt ? b.item[lo + k] : -1;
}

// ---- compute_tension, literally (ref: dfsph.py:264-305) -------------------------------------------------------------
// ref: kernels/CubicKernel.py:20-32
__device__ __forceinline__ float3 bk_gradW(const Dev &d, float3 r) {
  const float r2 = dot3(r, r);
  return r * cubic_gradW_scale(d, r2);
}
// first loop (:266-279): normal_i = sum over the raw row; liquid entries add (m / rho_j) gradW, EVERY entry then scales the
// running value by searchR (Q8): n = (...((t_0) h + t_1) h ...+ t_{K-1}) h = sum_p t_{K-1-p} h^{p+1}.  h = 0.1, so only the
// END of the row survives: the sum is taken from the last entry backwards until h^{p+1} underflows to zero (p = 45 in
// f32) -- the entries before that contribute exactly nothing to the reference's own f32 recurrence either.
// Also leaves the (normal, rho) gather record of the second loop in cg_Ad (free until the viscosity solve).
__global__ void __launch_bounds__(BLOCK) k_tension_normal_literal(Dev d, BucketTab b) {
  const int i = sorted_index(d);   // XCD-aware: every L2 serves one spatial eighth (launched with PASS)
  if (i >= d.nl || !owned(d, i)) return;
  const float4 pi = d.pos[i];
  const float3 xi = xyz(pi);
  float3 n = f3(0, 0, 0);
  float scale = d.h;
  raw_row_s<true, 8>(d, b, i, pi,
    [&](int js) { float4 pj = make_float4(0, 0, 0, 1); if (js >= 0) { pj = nb_ld(d.pos, js); pj.w = nb_ld(d.rho, js); } return pj; },   // (pos_j, rho_j)
    [&](int js, float4 pj) -> bool {
      if (js >= 0) {
        const float3 gw = bk_gradW(d, xi - xyz(pj));
        n = n + gw * (d.mass / pj.w * scale);
      }
      scale *= d.h;
      return scale > 0.0f;
    });
  d.normal[i] = f4(n, 0);
  d.cg_Ad[i] = f4(n, d.rho[i]);
}
// second loop (:281-305): every raw liquid entry adds k_ij * (-gamma (n_i - n_j)) (the cohesion value is overwritten, Q9);
// solid entries: adhesion inside the scene's 0.26 gate
__device__ __forceinline__ float adhesion_W(const Dev &d, float r);
__global__ void __launch_bounds__(BLOCK) k_tension_accel_literal(Dev d, BucketTab b) {
  // plain block order (ELEM): the boundary entries are expensive and sit in one region -- with the XCD-contiguous order of
  // the walks one XCD gets them all (817 -> 979 us at C5)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  const float4 pi = d.pos[i];
  const float3 xi = xyz(pi), ni = xyz(d.normal[i]);
  const float rhoi = d.rho[i];
  float3 dv = xyz(d.d_vel[i]);
  raw_row_s<false, 8>(d, b, i, pi,
    [&](int js) { return js >= 0 ? nb_ld(d.cg_Ad, js) : nb_ld(d.spos, js & 0x7fffffff); },   // (normal_j, rho_j) | boundary position
    [&](int js, float4 nr) -> bool {
    if (js >= 0) {
      const float kij = 2.0f * d.rho_L0 / (rhoi + nr.w);
      dv = dv + (ni - xyz(nr)) * (-d.tension * kij);
    } else {
      const float3 pj = xyz(nr);
      const float3 r = xi - pj, c = pj - f3(0.0f, 0.5f, 0.0f);
      const float len2 = dot3(r, r), dis = sqrtf(dot3(c, c));
      if (len2 > d.eps && dis < 0.26f) {
        const float rl = sqrtf(len2);
        dv = dv + r * (1.0f / rl) * (-d.tension_b * d.rho_S0 * d.VS0 * adhesion_W(d, rl));
      }
    }
    return true;
  });
  d.d_vel[i] = f4(dv, 0);
}

// ---- cal_anistropic_kernel over the raw rows (ref: ParticleData.py:220-289) ----------------------------------------------
// weight_func's support is 2 * mc_grid.searchR = 0.18 > h: a far aliased list entry from a cell three or four away can
// carry weight, so in faithful mode the sums run over the reference's own rows instead of the walked cells.
__global__ void __launch_bounds__(BLOCK) k_surf_pos_avr_raw(Dev d, BucketTab b, SurfDev s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  const float4 pi = d.pos[i];
  const float3 xi = xyz(pi);
  float sw = 0.f;
  float3 sx = f3(0, 0, 0);
  raw_row(d, b, i, pi, [&](int g) {
    if (g >= d.nl) return;
    const float3 xj = xyz(d.pos[d.where[g]]);
    const float3 r = xi - xj;
    const float w = aniso_w(dot3(r, r), s.R2);
    sw += w; sx = sx + xj * w;
  });
  s.pos_avr[i] = (sw > 0.0f) ? f4(sx * (1.0f / sw), 0) : make_float4(pi.x, pi.y, pi.z, 0);
}
__global__ void __launch_bounds__(BLOCK) k_surf_aniso_raw(Dev d, BucketTab b, SurfDev s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  const float kn = 0.5f, ne = 25.0f;
  float4 r0 = make_float4(kn, 0, 0, 0), r1 = make_float4(0, kn, 0, 0), r2v = make_float4(0, 0, kn, 0);
  if ((float)d.ncount[i] > ne) {
    const float4 pi = d.pos[i];
    const float3 xi = xyz(pi), pa = xyz(s.pos_avr[i]);
    float sw = 0.f, c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    raw_row(d, b, i, pi, [&](int g) {
      if (g >= d.nl) return;
      const float3 xj = xyz(d.pos[d.where[g]]);
      const float3 dx = xi - xj;
      const float w = aniso_w(dot3(dx, dx), s.R2);
      const float3 r = xj - pa;
      sw += w;
      c00 += w * r.x * r.x; c01 += w * r.x * r.y; c02 += w * r.x * r.z;
      c11 += w * r.y * r.y; c12 += w * r.y * r.z; c22 += w * r.z * r.z;
    });
    aniso_G(sw, c00, c01, c02, c11, c12, c22, r0, r1, r2v);
  }
  s.g0[i] = r0; s.g1[i] = r1; s.g2[i] = r2v;
}
