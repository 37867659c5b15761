// tile.h -- LDS-staged neighbour passes.
//
// EXPERIMENTAL (off by default, WCSPH_TILES=1; DESIGN.md 4 has the measurements: it does not beat the list path).
// A workgroup owns one TILE of TILE_X x TILE_Y x TILE_Z hash cells.  It stages the tile plus a 2-cell
// halo (the 5x5x5 walk radius, ref: HashGrid.py:82-85) into LDS as float4 records with coalesced loads:
// in the z-fastest cell order every (x,y) row of the staged box is ONE contiguous particle range, so the
// staging is T_NROW = 8x8 = 64 row copies (one wave computes the row table with a single prefix scan).
// The neighbour loop then runs on LDS (`ds_read_b128`, 256 B/clk/CU) instead of 16-byte L1 gathers, and
// the per-step neighbour list shrinks to 16-bit tile-local indices (the staged order is a pure function of
// the step's cell table, so an index stays valid for every pass of the step).
// Solids keep the global compact list (only particles near a wall have any).
#pragma once
#include "device.h"

#define TILE_X 8
#define TILE_Y 4
#define TILE_Z 12
#define T_HALO 2
#define T_RX (TILE_X + 2 * T_HALO)
#define T_RY (TILE_Y + 2 * T_HALO)
#define T_NROW (T_RX * T_RY)          /* 144 staged rows */
#define T_RZ (TILE_Z + 2 * T_HALO)    /* cells per staged row */
#define T_OWN (TILE_X * TILE_Y)       /* 64 rows whose particles this tile updates */
#define T_CAP 2304                    /* staged liquid particles per tile (1.5 per staged cell; rest density is 1.25) */
#define TBLOCK 512                    /* threads per tile workgroup: ~480 own particles at rest density */
static_assert(T_NROW <= TBLOCK && (T_OWN & (T_OWN - 1)) == 0, "row tables are built by one pass of the block");

struct TileTables {
  int row_lo[T_NROW];        // first sorted particle of the staged row
  int row_off[T_NROW + 1];   // LDS offset of the staged row; [T_NROW] = staged total
  int own_lo[T_OWN];         // first sorted particle of the own row
  int own_stage[T_OWN];      // LDS index of that particle
  int own_off[T_OWN + 1];    // ordinal offset of the own row; [T_OWN] = own total
  int qoff;                  // first list slot of the tile
  int tx0, ty0, tz0;
};

__device__ __forceinline__ int wave_excl_scan(int v, int *total) {
  int lane = threadIdx.x & (WAVE - 1);
  int inc = v;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    int t = __shfl_up(inc, o, WAVE);
    if (lane >= o) inc += t;
  }
  *total = __shfl(inc, WAVE - 1, WAVE);
  return inc - v;
}

// number of liquid particles each tile owns (one thread per tile)
__global__ void k_tile_count(Dev d, int *tile_p) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.ntile) return;
  int tz = t % d.ntz, ty = (t / d.ntz) % d.nty, tx = t / (d.ntz * d.nty);
  int z0 = tz * TILE_Z, z1 = min(z0 + TILE_Z, d.nz);
  int p = 0;
  // the cell table is only scanned around the occupied columns (grid.h scan_block_active): tiles elsewhere are empty
  if (d.bbox[3] < 0 || tx * TILE_X > d.bbox[3] + 1 || tx * TILE_X + TILE_X - 1 < d.bbox[0] - 1) { tile_p[t] = 0; return; }
  for (int ox = 0; ox < TILE_X; ox++) {
    int x = tx * TILE_X + ox;
    if (x >= d.nx) break;
    if (x < d.bbox[0] - 2 || x > d.bbox[3] + 2) continue;
    for (int oy = 0; oy < TILE_Y; oy++) {
      int y = ty * TILE_Y + oy;
      if (y >= d.ny) break;
      int base = (x * d.ny + y) * d.nz;
      p += d.cs_l[base + z1] - d.cs_l[base + z0];
    }
  }
  tile_p[t] = p;
}
// one block: compact the non-empty tiles (tile-id order keeps neighbours adjacent) and prefix their sizes
__global__ void __launch_bounds__(1024) k_tile_scan(Dev d, const int *tile_p) {
  __shared__ int s_cnt[1024], s_sum[1024];
  const int tid = threadIdx.x;
  const int per = (d.ntile + 1023) / 1024;
  const int b = tid * per, e = min(b + per, d.ntile);
  int c = 0, s = 0;
  for (int t = b; t < e; t++) { int p = tile_p[t]; c += (p > 0); s += p; }
  s_cnt[tid] = c; s_sum[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of both
    int vc = tid >= o ? s_cnt[tid - o] : 0, vs = tid >= o ? s_sum[tid - o] : 0;
    __syncthreads();
    s_cnt[tid] += vc; s_sum[tid] += vs;
    __syncthreads();
  }
  int oc = s_cnt[tid] - c, os = s_sum[tid] - s;
  for (int t = b; t < e; t++) {
    int p = tile_p[t];
    if (p > 0) { d.tile_list[oc] = t; d.tile_qoff[oc] = os; oc++; os += p; }
  }
  if (tid == 1023) d.tile_meta[0] = s_cnt[1023];
}

// ---- per-tile set-up: row table, own-row table, staging ---------------------------------------------------------
// exclusive scan of one int per thread over the whole workgroup
__device__ __forceinline__ int tblock_excl_scan(int v, int *total, int *wsum /* [TBLOCK/WAVE] LDS */) {
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int wt;
  const int ex = wave_excl_scan(v, &wt);
  if (lane == 0) wsum[w] = wt;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < TBLOCK / WAVE; k++) { const int x = wsum[k]; off += (k < w) ? x : 0; tot += x; }
  __syncthreads();
  *total = tot;
  return off + ex;
}

__device__ __forceinline__ void tile_tables(const Dev &d, TileTables &tt, int slot) {
  __shared__ int wsum[TBLOCK / WAVE];
  const int tid = threadIdx.x;
  const int t = d.tile_list[slot];
  const int tz = t % d.ntz, ty = (t / d.ntz) % d.nty, tx = t / (d.ntz * d.nty);
  const int tx0 = tx * TILE_X, ty0 = ty * TILE_Y, tz0 = tz * TILE_Z;
  // thread r < T_NROW owns staged row r; all its loads are issued before the scans
  const int rx = tid / T_RY, ry = tid % T_RY;
  const int x = tx0 - T_HALO + rx, y = ty0 - T_HALO + ry;
  const int ox = rx - T_HALO, oy = ry - T_HALO;
  // columns further than 2 from every occupied column are empty AND not covered by the cell-table scan (grid.h)
  const bool isrow = tid < T_NROW && (unsigned)x < (unsigned)d.nx && (unsigned)y < (unsigned)d.ny && x >= d.bbox[0] - 2 && x <= d.bbox[3] + 2;
  const bool isown = isrow && (unsigned)ox < TILE_X && (unsigned)oy < TILE_Y;
  int lo = 0, hi = 0, olo = 0, ohi = 0;
  if (isrow) {
    const int base = (x * d.ny + y) * d.nz;
    lo = d.cs_l[base + max(tz0 - T_HALO, 0)];
    hi = d.cs_l[base + min(tz0 + TILE_Z + T_HALO, d.nz)];
    if (isown) { olo = d.cs_l[base + tz0]; ohi = d.cs_l[base + min(tz0 + TILE_Z, d.nz)]; }
  }
  int total, ototal;
  const int off = tblock_excl_scan(hi - lo, &total, wsum);
  const int ooff = tblock_excl_scan(ohi - olo, &ototal, wsum);
  if (tid < T_NROW) { tt.row_lo[tid] = lo; tt.row_off[tid] = off; }
  if (tid < T_NROW && (unsigned)ox < TILE_X && (unsigned)oy < TILE_Y) {
    const int k = ox * TILE_Y + oy;
    tt.own_lo[k] = olo; tt.own_off[k] = ooff; tt.own_stage[k] = off + (olo - lo);
  }
  if (tid == 0) {
    tt.row_off[T_NROW] = total; tt.own_off[T_OWN] = ototal;
    tt.qoff = d.tile_qoff[slot]; tt.tx0 = tx0; tt.ty0 = ty0; tt.tz0 = tz0;
  }
  __syncthreads();
}

// copy the staged rows into LDS: 16 lanes per row, 16 rows per sweep.  A = (pos.xyz, wsrc[j]); B, C optional.
template <int NF>
__device__ __forceinline__ bool tile_stage(const Dev &d, const TileTables &tt, float4 *sA, float4 *sB, float4 *sC,
                                           const float *wsrc, const float4 *bsrc, const float *bw, const float4 *csrc) {
  const int total = tt.row_off[T_NROW];
  if (total > T_CAP) {  // density blow-up: cannot stage (the reference dies at 64 per bucket, HashGrid.py:72)
    if (threadIdx.x == 0) atomicOr(&d.flags[FLAG_ERR], ERRBIT_LISTCAP);
    return false;
  }
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  if (d.tile_ablate == 2) { __syncthreads(); return true; }   // experiment: compute phase only
#pragma unroll
  for (int r = g; r < T_NROW; r += TBLOCK / 16) {   // 64 row groups of 16 lanes: 144 rows in 3 sweeps
    const int lo = tt.row_lo[r], off = tt.row_off[r], len = tt.row_off[r + 1] - off;
    for (int e = l; e < len; e += 16) {
      const int j = lo + e;
      float4 p = d.pos[j];
      p.w = wsrc ? wsrc[j] : 0.f;
      sA[off + e] = p;
      if (NF >= 2) { float4 b = bsrc[j]; if (bw) b.w = bw[j]; sB[off + e] = b; }
      if (NF >= 3) sC[off + e] = csrc[j];
    }
  }
  __syncthreads();
  return true;
}

__device__ __forceinline__ void tblock_accumulate(double *dst, float v) {   // every thread of the workgroup calls this
  __shared__ float sh[TBLOCK / WAVE];
  v = wave_sum(v);
  if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < TBLOCK / WAVE; k++) s += sh[k];
    atomicAdd(dst + (blockIdx.x & (ACC_STRIPES - 1)), (double)s);
  }
}

struct TileParticle { int i, q, stage; };
__device__ __forceinline__ TileParticle tile_particle(const TileTables &tt, int p) {
  int r = 0;
#pragma unroll
  for (int s = T_OWN / 2; s > 0; s >>= 1) r += (tt.own_off[r + s] <= p) ? s : 0;   // largest r with own_off[r] <= p (6 steps)
  const int e = p - tt.own_off[r];
  TileParticle tp;
  tp.i = tt.own_lo[r] + e;
  tp.stage = tt.own_stage[r] + e;
  tp.q = tt.qoff + p;
  return tp;
}

// persistent tile loop: XCD x (= blockIdx % 8) owns one contiguous eighth of the active-tile list
#define TILE_LOOP_BEGIN                                                         \
  const int n_act_ = d.tile_meta[0];                                            \
  const int per_ = (n_act_ + 7) >> 3;                                           \
  const int lo_ = (blockIdx.x & 7) * per_, hi_ = min(lo_ + per_, n_act_);       \
  for (int slot_ = lo_ + (blockIdx.x >> 3); slot_ < hi_; slot_ += (gridDim.x >> 3)) {
#define TILE_LOOP_END \
    __syncthreads();  \
  }

// the neighbour loop on staged data: fl(jl, r, r2, w) for every liquid list entry (jl = LDS index, w = A.w of
// the neighbour); fs(j, r, r2) for the solid entries of the global compact list.  Four entries per trip.
// list layout: entry k of slot q in a 16-byte block of eight u16 ([k/8][q][k%8])
#define NBR16_AT(k, stride, q) ((((size_t)((k) >> 3)) * (stride) + (q)) * 8 + ((k) & 7))
template <class FL, class FS>
__device__ __forceinline__ void walk_tile(const Dev &d, const TileParticle &tp, float3 xi, const float4 *sA, FL &&fl, FS &&fs) {
  const size_t st = (size_t)d.nbr_stride;
  const uint4 *L = reinterpret_cast<const uint4 *>(d.nbr16) + tp.q;
  const int cl = d.cnt16[tp.q];
  if (cl > 0) {
    uint4 e = L[0];
    for (int k = 0; k < cl; k += 8) {
      const uint4 en = (k + 8 < cl) ? L[(size_t)((k >> 3) + 1) * st] : e;   // next block in flight during this one
      const int j[8] = {(int)(e.x & 0xffff), (int)(e.x >> 16), (int)(e.y & 0xffff), (int)(e.y >> 16),
                        (int)(e.z & 0xffff), (int)(e.z >> 16), (int)(e.w & 0xffff), (int)(e.w >> 16)};
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = sA[j[u]];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k + u < cl) {
          const float3 r = xi - xyz(a[u]);
          fl(j[u], r, dot3(r, r), a[u].w);
        }
      e = en;
    }
  }
  const int2 head = d.nhead[tp.i];
  const int cs = (int)((unsigned)head.x >> 16);
  const unsigned short *S = reinterpret_cast<const unsigned short *>(d.nbs + tp.i);
  int j0 = head.y;
  for (int k = 0; k < cs; k++) {   // solid delta stream of the global compact list (device.h "neighbour streams")
    const int dd = (int)(short)S[(size_t)(k >> 3) * st * 8 + (k & 7)];
    j0 += dd;
    if (!NB_IS_ENTRY(dd)) continue;
    const float3 r0 = xi - xyz(d.spos[j0]);
    fs(j0, r0, dot3(r0, r0));
  }
}

// ---- tile-local neighbour lists (once per step) ------------------------------------------------------------------
// The in-support subset of the reference's neighbour row (ref: HashGrid.py:88-106), bucket-multiset duplicates
// included (faithful mode: alias table of the particle's cell), as 16-bit indices into the staged order.
__global__ void __launch_bounds__(TBLOCK) kt_build_lists(Dev d) {
  __shared__ TileTables tt;
  __shared__ float4 sA[T_CAP];
  __shared__ int cst[T_NROW][T_RZ + 1];   // LDS index of the first particle of every staged cell
  const size_t st = (size_t)d.nbr_stride;
  TILE_LOOP_BEGIN
    tile_tables(d, tt, slot_);
    if (tile_stage<1>(d, tt, sA, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
      for (int e = threadIdx.x; e < T_NROW * (T_RZ + 1); e += TBLOCK) {
        const int r = e / (T_RZ + 1), zc = e % (T_RZ + 1);
        const int x = tt.tx0 - T_HALO + r / T_RY, y = tt.ty0 - T_HALO + r % T_RY;
        int v = tt.row_off[r];
        if ((unsigned)x < (unsigned)d.nx && (unsigned)y < (unsigned)d.ny && x >= d.bbox[0] - 2 && x <= d.bbox[3] + 2) {
          const int z = min(max(tt.tz0 - T_HALO + zc, 0), d.nz);
          v += d.cs_l[(x * d.ny + y) * d.nz + z] - tt.row_lo[r];
          v = min(max(v, tt.row_off[r]), tt.row_off[r + 1]);
        }
        cst[r][zc] = v;
      }
      __syncthreads();
      const int P = tt.own_off[T_OWN];
      for (int p = threadIdx.x; p < P; p += TBLOCK) {
        const TileParticle tp = tile_particle(tt, p);
        const float4 pi = sA[tp.stage];
        const float3 xi = xyz(pi);
        const int cw = __float_as_int(d.pos[tp.i].w);
        const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
        const int lx = cx - tt.tx0 + T_HALO, ly = cy - tt.ty0 + T_HALO, lz = cz - tt.tz0 + T_HALO;
        int cnt = 0;
        auto scan = [&](int a, int b, int rep) {
          for (int e = a; e < b; e++) {
            const float3 r = xi - xyz(sA[e]);
            if (dot3(r, r) <= d.list_r2 && e != tp.stage)
              for (int q = 0; q < rep; q++) { if (cnt < d.nbr_cap) d.nbr16[NBR16_AT(cnt, st, tp.q)] = (uint16_t)e; cnt++; }
          }
        };
#pragma unroll 1
        for (int dx = -2; dx <= 2; dx++)
#pragma unroll 1
          for (int dy = -2; dy <= 2; dy++) {
            const int rs = (lx + dx) * T_RY + (ly + dy);
            scan(cst[rs][max(lz - 2, 0)], cst[rs][min(lz + 3, T_RZ)], 1);
          }
        const int info = d.cell_info[(cx * d.ny + cy) * d.nz + cz];
        if (info >> 1) {  // faithful: walked cells that share a bucket are visited again (Q1)
          const uint16_t *al = d.alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE;
          const int n = al[0];
          for (int t = 0; t < n; t++) {
            const int kk = al[2 + t] & 0xff, extra = al[2 + t] >> 8;
            const int rs = (lx + kk / 25 - 2) * T_RY + (ly + (kk / 5) % 5 - 2), zc = lz + kk % 5 - 2;
            scan(cst[rs][zc], cst[rs][zc + 1], extra);
          }
        }
        if (cnt > d.nbr_cap) { atomicOr(&d.flags[FLAG_ERR], ERRBIT_LISTCAP); cnt = d.nbr_cap; }
        for (int k = cnt; (k & 7) && k < d.nbr_cap; k++) d.nbr16[NBR16_AT(k, st, tp.q)] = 0;  // whole blocks are loaded
        d.cnt16[tp.q] = (uint16_t)cnt;
      }
    }
  TILE_LOOP_END
}

// ---- the solver-iteration passes on tiles ----------------------------------------------------------------------------
// ref: dfsph.py:460-471 / 529-541 / 420-439 (see k_vel_from_k): A.w carries k_j
template <int COND>
__global__ void __launch_bounds__(TBLOCK, 4) kt_vel_from_k(Dev d, const float *kb, float dt) {
  __shared__ TileTables tt;
  __shared__ float4 sA[T_CAP];
  TILE_LOOP_BEGIN
    tile_tables(d, tt, slot_);
    if (tile_stage<1>(d, tt, sA, nullptr, nullptr, kb, nullptr, nullptr, nullptr)) {
      const int P = d.tile_ablate == 1 ? 0 : tt.own_off[T_OWN];   // experiment: staging phase only
      for (int p = threadIdx.x; p < P; p += TBLOCK) {
        const TileParticle tp = tile_particle(tt, p);
        if (d.oid[tp.i] >= d.n_owned) continue;
        if (COND == 1 && !(d.adv_rho[tp.i] > 0.0f)) continue;
        const float4 pi = sA[tp.stage];
        const float ki = pi.w;
        const bool ks = fabsf(ki) > d.eps;
        float3 acc = f3(0, 0, 0);
        walk_tile(d, tp, xyz(pi), sA,
                  [&](int, float3 r, float r2, float kj) {
                    const float s = ki + kj;
                    if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
                  },
                  [&](int, float3 r, float r2) {
                    if (ks) acc = acc + r * (ki * (d.VS0 / d.VL0) * cubic_gradW_scale(d, r2));
                  });
        float4 v = d.vel[tp.i];
        const float f = dt * d.VL0;
        v.x += f * acc.x; v.y += f * acc.y; v.z += f * acc.z;
        d.vel[tp.i] = v;
      }
    }
  TILE_LOOP_END
}

// ref: dfsph.py:394-412 (see k_drho_press) and :374-392 (k_drho_div): B = vel
template <int MODE, int DIV>
__global__ void __launch_bounds__(TBLOCK, 4) kt_drho(Dev d, float dt) {
  __shared__ TileTables tt;
  __shared__ float4 sA[T_CAP];
  __shared__ float4 sB[T_CAP];
  float part = 0.f;
  TILE_LOOP_BEGIN
    tile_tables(d, tt, slot_);
    if (tile_stage<2>(d, tt, sA, sB, nullptr, nullptr, d.vel, nullptr, nullptr)) {
      const int P = d.tile_ablate == 1 ? 0 : tt.own_off[T_OWN];
      for (int p = threadIdx.x; p < P; p += TBLOCK) {
        const TileParticle tp = tile_particle(tt, p);
        const int i = tp.i;
        if (d.oid[i] >= d.n_owned) continue;
        if (DIV && MODE == DRHO_WARM) d.kappa_v[i] = 0.5f * fmaxf(d.kappa_v[i] / dt, -0.5f * d.rho_L0 * d.rho_L0);
        const float3 xi = xyz(sA[tp.stage]), vi = xyz(sB[tp.stage]);
        float t = 0.f;
        const float VS = DIV ? d.VS0 : d.VL0;   // update_drho_pressure uses VL0 for solids too (dfsph.py:408)
        walk_tile(d, tp, xi, sA,
                  [&](int jl, float3 r, float r2, float) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(sB[jl]), r); },
                  [&](int, float3 r, float r2) { t += VS * cubic_gradW_scale(d, r2) * dot3(vi, r); });
        float a;
        if (DIV) {
          a = fmaxf(t, 0.0f);
          if (d.ncount[i] < 20) a = 0.0f;
          d.adv_rho[i] = a;
          if (MODE == DRHO_BEGIN) { d.alpha[i] = d.alpha[i] / dt; d.kappa_v[i] = 0.0f; }
          part += a;
        } else {
          a = fmaxf(1.0f, d.rho[i] / d.rho_L0 + dt * t);
          d.adv_rho[i] = a;
          if (MODE == DRHO_BEGIN) { d.alpha[i] = d.alpha[i] / dt / dt; d.kappa[i] = 0.0f; }
          part += a - 1.0f;
        }
      }
    }
  TILE_LOOP_END
  if (MODE == DRHO_ITER) tblock_accumulate(ACC_SLOT(d, ACC_ERR), part);
}
