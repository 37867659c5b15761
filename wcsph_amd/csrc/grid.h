// grid.h -- HashGrid replacement: dense tile-ordered cell table (device.h cell_lin), counting sort of the particles,
// and (faithful mode) the reference hash table's aliasing reproduced as static per-cell
// multiplicity tables + a box-filtered bucket-count field for the raw list length.
// ref: HashGrid.py:44-124.
#pragma once
#include "device.h"

// ref: HashGrid.py:108-114 -- int32 wrap-around products, xor, modulo normalised to [0, T)
__device__ __forceinline__ int ref_hash(int x, int y, int z, int T) {
  int p = (int)(73856093u * (unsigned)x) ^ (int)(19349663u * (unsigned)y) ^ (int)(83492791u * (unsigned)z);
  return ((p % T) + T) % T;
}

// ref: HashGrid.py:68 -- ti.cast((pos - min) * invGridR, i32): f32 sub, f32 mul (no fma), trunc to zero
__device__ __forceinline__ int cell_packed(const Dev &d, float3 p, int *lin, bool *finite = nullptr) {
  float fx = __fmul_rn(__fsub_rn(p.x, d.minx), d.inv_grid);
  float fy = __fmul_rn(__fsub_rn(p.y, d.miny), d.inv_grid);
  float fz = __fmul_rn(__fsub_rn(p.z, d.minz), d.inv_grid);
  bool fin = (fabsf(fx) < 1e9f) && (fabsf(fy) < 1e9f) && (fabsf(fz) < 1e9f);  // NaN/inf => outside
  if (finite) *finite = (fabsf(fx) <= 3.0e38f) && (fabsf(fy) <= 3.0e38f) && (fabsf(fz) <= 3.0e38f);
  int cx = (int)fx - d.x0, cy = (int)fy, cz = (int)fz;  // truncation toward zero BEFORE the window shift (Q4)
  bool in = fin && (unsigned)cx < (unsigned)d.nx && (unsigned)cy < (unsigned)d.ny && (unsigned)cz < (unsigned)d.nz;
  *lin = in ? cell_lin(d, cx, cy, cz) : d.ncell_p;
  return in ? (cx | (cy << 10) | (cz << 20)) : -1;
}

// neighborCount regions (see k_ncount_box below): BC0 x BC1 x BCZ cells (outer x middle x z of the tile order = 4 x 2 x 1 tiles)
#define BC0 8
#define BC1 8
#define BCZ 16
struct Regions { int nb0, nb1, nbz; };
__device__ __forceinline__ Regions regions_of(const Dev &d) {
  Regions r;
  r.nb0 = ((d.perm ? d.ny : d.nx) + BC0 - 1) / BC0; r.nb1 = ((d.perm ? d.nx : d.ny) + BC1 - 1) / BC1; r.nbz = (d.nz + BCZ - 1) / BCZ;
  return r;
}
__device__ __forceinline__ int region_of_cell(const Dev &d, int w) {   // w: packed cell
  const Regions R = regions_of(d);
  const int cx = w & 1023, cy = (w >> 10) & 1023, cz = w >> 20;
  return (((d.perm ? cy : cx) / BC0) * R.nb1 + (d.perm ? cx : cy) / BC1) * R.nbz + cz / BCZ;
}

// phase 1 of the counting sort (ref: HashGrid.py:67-76 "insert pos"): cell of every particle, slot in cell.
// occ != nullptr: the particle's neighborCount region is flagged (k_ncount_box clears the flags it consumes).
// bbox != nullptr: also this block's bounding box of the occupied cells -> bbox[6 * blockIdx.x ..] (no atomics: the
// particles are x-sorted, so a shared max would be raised by nearly every block); k_bbox_final reduces them.
// inc8 != nullptr (one faithful context, between two of its own updates: api_grid.h grid_assign): the byte table of the reference
// bucket occupancy is kept UP TO DATE instead of being rebuilt -- a particle's previous cell travels with it in pos.w (the sort
// writes it there, every position update keeps it), so only a particle that changed its cell touches the table: one count less
// in its old bucket, one more in the new one (32-bit atomics on the containing word, no value returned).  The cap is checked by
// k_cellval_occ, which reads the count of every cell that holds a particle AFTER all of the update's atomics (a bucket that one
// particle leaves and another enters in the same update may pass the cap for a moment, whichever atomic lands first).
// live != nullptr (a slab whose state stays in the library between steps, api_slab.h wcsph_state_apply): the rows are in LAST
// step's sorted order with holes -- row i is original row live[i]; it is DEAD (a row that left, or an old ghost slot nobody
// took) when live[i] >= d.nl, and owned when live[i] < count_limit.  Dead rows get cell word CELL_DEAD and vanish in the sort.
#define CELL_DEAD (-2)
__global__ void __launch_bounds__(BLOCK) k_cell_assign(Dev d, const float4 *pos, int n, int *cnt, int *pw, int *slot, int *bcount, int *bbox, int count_limit, int *occ = nullptr,
                                                      unsigned *inc8 = nullptr, const int *live = nullptr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  bool dead = false;
  if (i < n && live) { dead = live[i] >= d.nl; if (dead) st_once(pw + i, CELL_DEAD); }
  if (i < n && !dead) {
    float4 p = ld_once(pos + i);
    const bool mine = live ? live[i] < count_limit : i < count_limit;
    int lin;
    bool finite;
    int w = cell_packed(d, xyz(p), &lin, &finite);
    // the reference's only failure check is a NaN position (dfsph.py:645-647, one particle); here every particle, every step
    if (!finite) atomicOr(&d.flags[FLAG_ERR], ERRBIT_NONFINITE);
    st_once(pw + i, w);
    st_once(slot + i, atomicAdd(&cnt[lin], 1));
    if (inc8 && mine) {
      const int wo = __float_as_int(p.w);
      if (wo != w) {
        if (wo >= 0) {
          const int h = ref_hash((wo & 1023) + d.x0, (wo >> 10) & 1023, wo >> 20, d.table_size);
          atomicSub(&inc8[h >> 2], 1u << ((h & 3) * 8));
        }
        if (w >= 0) {
          const int h = ref_hash((w & 1023) + d.x0, (w >> 10) & 1023, w >> 20, d.table_size);
          // saturating: a byte at 255 stays there (the add is taken back) instead of carrying into the next bucket's count, and the
          // cap is reported; the host then rebuilds the table at the next update (api_core.h check_flags)
          const unsigned sh = (h & 3) * 8;
          const unsigned old = atomicAdd(&inc8[h >> 2], 1u << sh);
          if (((old >> sh) & 255u) == 255u) { atomicSub(&inc8[h >> 2], 1u << sh); atomicOr(&d.flags[FLAG_ERR], ERRBIT_BUCKET); }
        }
      }
    }
    if (w >= 0) {
      // reference bucket occupancy: every particle is counted ONCE across the ranks of a slab decomposition -- liquid
      // particles by their owner (ghost rows sit behind the owned ones before the sort), boundary particles by the rank
      // whose owned columns hold them (own_lo <= global cell x < own_hi)
      const int gx = (w & 1023) + d.x0;
      if (bcount && mine && gx >= d.own_lo && gx < d.own_hi)
        atomicAdd(&bcount[ref_hash(gx, (w >> 10) & 1023, w >> 20, d.table_size)], 1);
      lo[0] = hi[0] = w & 1023; lo[1] = hi[1] = (w >> 10) & 1023; lo[2] = hi[2] = w >> 20;
      if (occ) occ[region_of_cell(d, w)] = 1;
    }
  }
  if (!bbox) return;
  __shared__ int slo[3][BLOCK / WAVE], shi[3][BLOCK / WAVE];
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_down(lo[a], o, WAVE)); hi[a] = max(hi[a], __shfl_down(hi[a], o, WAVE)); }
    if ((threadIdx.x & (WAVE - 1)) == 0) { slo[a][threadIdx.x / WAVE] = lo[a]; shi[a][threadIdx.x / WAVE] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    int l = slo[a][0], h = shi[a][0];
#pragma unroll
    for (int k = 1; k < BLOCK / WAVE; k++) { l = min(l, slo[a][k]); h = max(h, shi[a][k]); }
    bbox[6 * blockIdx.x + a] = l;
    bbox[6 * blockIdx.x + 3 + a] = h;
  }
}
// the reference bucket histogram on its own (wcsph_grid_update of one faithful context: beside the sort, on the side stream):
// rows [0, n) = the owned liquid rows, cells from k_cell_assign's packed words, the same ownership rule
// live != nullptr (resident slab state, see k_cell_assign): the owned rows are those with live[i] < n_owned, among all n rows
__global__ void __launch_bounds__(BLOCK) k_bucket_hist(Dev d, const int *pw, int n, int *bcount, const int *live = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = pw[i];
  if (w < 0 || (live && live[i] >= d.n_owned)) return;
  const int gx = (w & 1023) + d.x0;
  if (gx >= d.own_lo && gx < d.own_hi) atomicAdd(&bcount[ref_hash(gx, (w >> 10) & 1023, w >> 20, d.table_size)], 1);
}
// ---- the same histogram without one memory-side atomic per particle (profiles/r03w_pmc_bucket_hist.txt: every one of the
// 2 M atomics of k_bucket_hist misses the XCD's L2; 79 us at C3) ------------------------------------------------------------
// k_hash_bin: a workgroup takes HB_ROWS rows per thread, forms their bucket ids, counts them per RANGE of 2^HB_SHIFT buckets in
// LDS, reserves its share of every range's list with ONE device atomic per range and writes the bucket ids there.
// k_bin_hist: one workgroup per range counts its list in an LDS table of 2^HB_SHIFT ints and writes this step's liquid
// histogram of the range, zeros included, coalesced.  A list that does not fit its buffer raises bins[HB_FLAG]: k_bin_hist
// then writes nothing and the gated k_bucket_hist takes the histogram with atomics.  The flag is raised to the update's
// generation number `gen` (a host counter): nothing has to clear it between updates.
#define HB_SHIFT 15
#define HB_MAXBINS 1024   /* tables up to 33.5 M buckets (a 16 M + 4 M particle decomposition has 20 M) */
#define HB_FLAG HB_MAXBINS
#define HB_ROWS 16
__global__ void __launch_bounds__(BLOCK) k_hash_bin(Dev d, const int *pw, int n, int nbins, int cap, int *bins, int *binbuf, int gen, const int *live = nullptr) {
  __shared__ int cnt[HB_MAXBINS], base[HB_MAXBINS];
  for (int b = threadIdx.x; b < nbins; b += BLOCK) cnt[b] = 0;
  __syncthreads();
  const int first = blockIdx.x * (BLOCK * HB_ROWS) + threadIdx.x;
  int h[HB_ROWS], at[HB_ROWS];
#pragma unroll
  for (int r = 0; r < HB_ROWS; r++) {
    const int i = first + r * BLOCK;
    h[r] = -1; at[r] = 0;
    if (i < n) {
      const int w = pw[i];
      if (w >= 0 && !(live && live[i] >= d.n_owned)) {
        const int gx = (w & 1023) + d.x0;
        if (gx >= d.own_lo && gx < d.own_hi) h[r] = ref_hash(gx, (w >> 10) & 1023, w >> 20, d.table_size);
      }
    }
    if (h[r] >= 0) at[r] = atomicAdd(&cnt[h[r] >> HB_SHIFT], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += BLOCK) base[b] = cnt[b] ? atomicAdd(&bins[b], cnt[b]) : 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < HB_ROWS; r++) {
    if (h[r] < 0) continue;
    const int b = h[r] >> HB_SHIFT, o = base[b] + at[r];
    if (o < cap) binbuf[(size_t)b * cap + o] = h[r];
    else bins[HB_FLAG] = gen;
  }
}
__global__ void __launch_bounds__(1024) k_bin_hist(Dev d, int cap, int *bins, const int *binbuf, int *bl, int gen) {
  __shared__ int tab[1 << HB_SHIFT];
  if (bins[HB_FLAG] == gen) return;                // grid-uniform: an overflowing list -> the atomics path
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < (1 << HB_SHIFT); k += 1024) tab[k] = 0;
  __syncthreads();
  const int n = bins[b];
  const int *src = binbuf + (size_t)b * cap;
  for (int k = threadIdx.x; k < n; k += 1024) atomicAdd(&tab[src[k] & ((1 << HB_SHIFT) - 1)], 1);
  __syncthreads();
  if (threadIdx.x == 0) bins[b] = 0;               // consumed: the next update starts from empty lists
  const int b0 = b << HB_SHIFT;
  for (int k = threadIdx.x; k < (1 << HB_SHIFT) && b0 + k < d.table_size; k += 1024) bl[b0 + k] = tab[k];
}
// the atomics path, only when a range list overflowed (also leaves the lists empty for the next update)
// (grid-stride over a SMALL grid: in the usual case every block of this launch only reads the flag and leaves)
__global__ void __launch_bounds__(BLOCK) k_bucket_hist_gated(Dev d, const int *pw, int n, int *bcount, int nbins, int *bins, int gen, const int *live = nullptr) {
  if (bins[HB_FLAG] != gen) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(n, nbins); i += gridDim.x * blockDim.x) {
    if (i < nbins) bins[i] = 0;
    if (i >= n) continue;
    const int w = pw[i];
    if (w < 0 || (live && live[i] >= d.n_owned)) continue;
    const int gx = (w & 1023) + d.x0;
    if (gx >= d.own_lo && gx < d.own_hi) atomicAdd(&bcount[ref_hash(gx, (w >> 10) & 1023, w >> 20, d.table_size)], 1);
  }
}
// one block: reduce the per-block boxes of k_cell_assign into out[6] = {xmin,ymin,zmin,xmax,ymax,zmax} (max < 0: empty)
__global__ void __launch_bounds__(BLOCK) k_bbox_final(const int *part, int nparts, int *out) {
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  for (int b = threadIdx.x; b < nparts; b += BLOCK)
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], part[6 * b + a]); hi[a] = max(hi[a], part[6 * b + 3 + a]); }
  __shared__ int slo[3][BLOCK / WAVE], shi[3][BLOCK / WAVE];
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_down(lo[a], o, WAVE)); hi[a] = max(hi[a], __shfl_down(hi[a], o, WAVE)); }
    if ((threadIdx.x & (WAVE - 1)) == 0) { slo[a][threadIdx.x / WAVE] = lo[a]; shi[a][threadIdx.x / WAVE] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    int l = slo[a][0], h = shi[a][0];
#pragma unroll
    for (int k = 1; k < BLOCK / WAVE; k++) { l = min(l, slo[a][k]); h = max(h, shi[a][k]); }
    out[a] = l; out[3 + a] = h;
  }
}

__device__ __forceinline__ int lin_of(const Dev &d, int w) {
  return w < 0 ? d.ncell_p : cell_lin(d, w & 1023, (w >> 10) & 1023, w >> 20);
}
// plain (x, y, z) index: the faithful box filter's scratch arrays
__device__ __forceinline__ int plain_of(const Dev &d, int w) { return ((w & 1023) * d.ny + ((w >> 10) & 1023)) * d.nz + (w >> 20); }

// phase 3: move every persistent liquid field to its cell-sorted position
struct SortBuffers {
  const float4 *pos, *vel, *vel_guess, *omega;
  const float *kappa, *kappa_v, *pressure;   // pressure: iisph.py:298 carries it into the next step
  const int *oid;
  float4 *pos_o, *vel_o, *vel_guess_o, *omega_o;
  float *kappa_o, *kappa_v_o, *pressure_o;
  int *oid_o, *where;
};
// n: rows to scatter (d.nl; more when the state has holes: rows with cell word CELL_DEAD are dropped, see k_cell_assign)
__global__ void k_scatter_liquid(Dev d, SortBuffers b, const int *cs, const int *pw, const int *slot, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int w = ld_once(pw + i);
  if (w == CELL_DEAD) return;
  int dst = cs[lin_of(d, w)] + ld_once(slot + i);
  float4 p = ld_once(b.pos + i);
  p.w = __int_as_float(w);
  // the sources are read once, in order: non-temporal.  The destinations are scattered (a particle that changed its
  // cell lands elsewhere): plain stores, which the L2 merges into whole lines -- non-temporal ones cost this kernel
  // +60 % in the developed flow (profiles/r03n_ab_once.md)
  b.pos_o[dst] = p;
  b.vel_o[dst] = ld_once(b.vel + i);
  b.vel_guess_o[dst] = ld_once(b.vel_guess + i);
  b.omega_o[dst] = ld_once(b.omega + i);
  b.kappa_o[dst] = ld_once(b.kappa + i);
  b.kappa_v_o[dst] = ld_once(b.kappa_v + i);
  b.pressure_o[dst] = ld_once(b.pressure + i);
  int o = ld_once(b.oid + i);
  b.oid_o[dst] = o;
  b.where[o] = dst;
}
__global__ void k_scatter_solid(Dev d, const float4 *pos, float4 *spos, const int *cs, const int *pw, const int *slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.ns) return;
  int dst = cs[lin_of(d, pw[i])] + slot[i];
  float4 p = pos[i];
  p.w = __int_as_float(i);
  spos[dst] = p;
}

// ---- exclusive scan over the cell counters (two launches, 2048 elements per block) ----------
#define SCAN_PER_THREAD 8
#define SCAN_TILE (BLOCK * SCAN_PER_THREAD)
__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
  __shared__ int wsum[BLOCK / WAVE];
  int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = v;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    int t = __shfl_up(inc, o, WAVE);
    if (lane >= o) inc += t;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < BLOCK / WAVE; k++) {
    if (k < w) off += wsum[k];
    tot += wsum[k];
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}
// Scan blocks none of whose tiles is within the walk radius (2 cells) of the occupied box hold only zeros: their block
// sum is written as 0 without reading, their outputs are left alone (nothing reads the table that far from every
// particle).  A tile is needed if it -- or its predecessor in tile order, whose last cell's END is this tile's first
// START (row_pieces) -- intersects the grown box.  The last block (out-of-box bin, grand total) is always processed.
// bbox == nullptr: everything.  Every thread of the block must call this.
__device__ __forceinline__ bool scan_block_active(const Dev &d, const int *bbox, int n) {
  if (!bbox || (int)blockIdx.x == (n - 1) / SCAN_TILE) return true;
  if (bbox[3] < 0) return false;
  const int lx = max(bbox[0] - 2, 0), ly = max(bbox[1] - 2, 0), lz = max(bbox[2] - 2, 0);
  const int hx = min(bbox[3] + 2, d.nx - 1), hy = min(bbox[4] + 2, d.ny - 1), hz = min(bbox[5] + 2, d.nz - 1);
  const int l0 = (d.perm ? ly : lx) >> T0B, h0 = (d.perm ? hy : hx) >> T0B, l1 = (d.perm ? lx : ly) >> T1B, h1 = (d.perm ? hx : hy) >> T1B;
  const int l2 = lz >> T2B, h2 = hz >> T2B;
  const int tiles_per_block = SCAN_TILE >> TILE_SHIFT;
  int need = 0;
  for (int k = threadIdx.x; k <= tiles_per_block; k += BLOCK) {   // the block's tiles and the one before them
    const int T = blockIdx.x * tiles_per_block - 1 + k;
    if (T >= 0) {
      const int t2 = T % d.nt2, q = T / d.nt2, t1 = q % d.nt1, t0 = q / d.nt1;
      need |= (t0 >= l0 && t0 <= h0 && t1 >= l1 && t1 <= h1 && t2 >= l2 && t2 <= h2);
    }
  }
  return __syncthreads_or(need) != 0;
}
// every thread of the block must call this; returns the block's total to every thread
__device__ __forceinline__ int block_sum(int v) {
  __shared__ int ws[BLOCK / WAVE];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  __syncthreads();   // (the previous user of ws, if any, is done)
  if ((threadIdx.x & (WAVE - 1)) == 0) ws[threadIdx.x / WAVE] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int k = 0; k < BLOCK / WAVE; k++) t += ws[k];
  return t;
}
__global__ void k_scan_reduce(Dev d, const int *in, int n, int *bsum, const int *bbox) {
  if (!scan_block_active(d, bbox, n)) { if (threadIdx.x == 0) bsum[blockIdx.x] = 0; return; }
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  int s = 0;
  if (base + SCAN_PER_THREAD <= n) {
    const int4 a = *reinterpret_cast<const int4 *>(in + base), b = *reinterpret_cast<const int4 *>(in + base + 4);
    s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) s += (base + k < n) ? in[base + k] : 0;
  }
  const int tot = block_sum(s);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// (rounds 1-3 scanned the block sums in a one-block launch of their own, 14 us + two dependent dispatch gaps; since round 4 every
// ACTIVE block of k_scan_final sums the block totals before it itself: <= 32 KB of coalesced, L2-resident reads per block, all
// of a thread's loads in flight at once -- cheaper than the launch it replaces, and a block outside the occupied box skips it.
// The one-launch alternative, decoupled look-back, lost: profiles/r04_experiments.md (3))
// clear != 0: the counters are zeroed as they are consumed, so the next grid update needs no 65 MB memset (only
// cells of the active blocks can be non-zero: every counted particle lies inside the occupied box or in the last bin)
// Grids with many scan blocks (> SCAN_BPRE_MIN: beyond ~16 M cells) would have every active block re-read up to MBs of block
// totals: there the totals are prefix-summed once by a one-block launch (k_scan_bsums) and k_scan_final reads one word (bpre).
#define SCAN_BPRE_MIN 8192
__global__ void __launch_bounds__(1024) k_scan_bsums(const int *bsum, int nb, int *bpre) {
  __shared__ int part[1024];
  const int per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, nb);
  int s = 0;
  for (int k = lo; k < hi; k++) s += bsum[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int k = 0; k < 1024; k++) { const int t = part[k]; part[k] = run; run += t; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int k = lo; k < hi; k++) { bpre[k] = run; run += bsum[k]; }
}
__global__ void k_scan_final(Dev d, int *in, int n, const int *bsum, int *out, const int *bbox, int clear, const int *bpre = nullptr) {
  if (!scan_block_active(d, bbox, n)) return;
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  static_assert(SCAN_PER_THREAD == 8, "two 16-byte accesses per thread");
  const bool whole = base + SCAN_PER_THREAD <= n;   // (only the last block has threads past the end)
  int v[SCAN_PER_THREAD], s = 0;
  if (whole) {
    const int4 a = *reinterpret_cast<const int4 *>(in + base), b = *reinterpret_cast<const int4 *>(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) s += v[k];
    // consumed: one 32-byte clear per thread that counted anything, not one predicated 4-byte store per non-empty cell
    if (clear && s != 0) { *reinterpret_cast<int4 *>(in + base) = make_int4(0, 0, 0, 0); *reinterpret_cast<int4 *>(in + base + 4) = make_int4(0, 0, 0, 0); }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
      v[k] = (base + k < n) ? in[base + k] : 0;
      s += v[k];
      if (clear && v[k] != 0) in[base + k] = 0;
    }
  }
  // the sum of the block totals before this block (bsum: raw totals, k_scan_reduce)
  int before = 0;
  if (bpre) { if (threadIdx.x == 0) before = bpre[blockIdx.x]; }
  else for (int k = threadIdx.x; k < (int)blockIdx.x; k += BLOCK) before += bsum[k];
  int tot;
  int ex = block_exclusive_scan(s, &tot);
  ex += block_sum(before);
  if (whole) {
    int4 a, b;
    a.x = ex; a.y = a.x + v[0]; a.z = a.y + v[1]; a.w = a.z + v[2];
    b.x = a.w + v[3]; b.y = b.x + v[4]; b.z = b.y + v[5]; b.w = b.z + v[6];
    *reinterpret_cast<int4 *>(out + base) = a; *reinterpret_cast<int4 *>(out + base + 4) = b;
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
  }
}

// slab decomposition, faithful mode: the per-step liquid histogram travels as two 16-bit counts per int32 word (an int32
// SUM all-reduce adds both halves independently while every global count stays below 65536; RCCL has no 16-bit integer).
// Each rank's count is clamped to `clamp` = max_in_grid + 1: the summed count still exceeds the cap wherever any rank's
// did (k_bucket_sum raises ERRBIT_BUCKET), and the low half cannot carry into the neighbouring bucket's count.
__global__ void k_hist_pack16(const int *l, int T, int *packed, int clamp) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * w >= T) return;
  const int a = min(l[2 * w], clamp), b = (2 * w + 1 < T) ? min(l[2 * w + 1], clamp) : 0;
  packed[w] = a | (b << 16);
}
__global__ void k_hist_unpack16(const int *packed, int T, int *l) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * w >= T) return;
  const unsigned v = (unsigned)packed[w];
  l[2 * w] = (int)(v & 0xffffu);
  if (2 * w + 1 < T) l[2 * w + 1] = (int)(v >> 16);
}

// ---- static per-cell info ---------------------------------------------------------------
// bit0: any solid particle inside the 5x5x5 walk of this cell (lets bulk particles skip the solid walk)
__global__ void k_cell_solid_near(Dev d, int *cell_info) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;   // plain index
  if (c >= d.ncell) return;
  int cz = c % d.nz, cy = (c / d.nz) % d.ny, cx = c / (d.nz * d.ny);
  int z0 = max(cz - 2, 0), z1 = min(cz + 2, d.nz - 1) + 1;
  int any = 0;
  for (int dx = -2; dx <= 2; dx++) {
    int x = cx + dx;
    if ((unsigned)x >= (unsigned)d.nx) continue;
    for (int dy = -2; dy <= 2; dy++) {
      int y = cy + dy;
      if ((unsigned)y >= (unsigned)d.ny) continue;
      const RowPieces rp = row_pieces(d, d.cs_s, x, y, z0, z1);
      any |= (rp.hi0 > rp.lo0) | (rp.hi1 > rp.lo1);
    }
  }
  const int l = cell_lin(d, cx, cy, cz);
  cell_info[l] = (cell_info[l] & ~1) | (any ? 1 : 0);
}

// Faithful mode (Q1): which cells have two walked cells in one bucket?  64 cells per block, the 125
// bucket ids of each cell in LDS ([k][lane], conflict free), all-pairs compare.
#define ALIAS_TB 64
__device__ __forceinline__ void alias_hashes(const Dev &d, int c, int (*hs)[ALIAS_TB]) {
  int t = threadIdx.x;
  int cz = c % d.nz, cy = (c / d.nz) % d.ny, cx = c / (d.nz * d.ny);
  for (int k = 0; k < 125; k++) {
    int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
    bool in = (unsigned)x < (unsigned)d.nx && (unsigned)y < (unsigned)d.ny && (unsigned)z < (unsigned)d.nz;
    hs[k][t] = in ? ref_hash(x + d.x0, y, z, d.table_size) : (-1 - k);
  }
}
__global__ void __launch_bounds__(ALIAS_TB) k_alias_detect(Dev d, int *cell_info, int *n_alias) {
  __shared__ int hs[125][ALIAS_TB];
  int c = blockIdx.x * ALIAS_TB + threadIdx.x, t = threadIdx.x;
  if (c >= d.ncell) return;
  alias_hashes(d, c, hs);
  int dup = 0;
  for (int a = 1; a < 125; a++) {
    int ha = hs[a][t];
    for (int b = 0; b < a; b++) dup |= (hs[b][t] == ha);
  }
  if (dup) {
    int slot = atomicAdd(n_alias, 1);
    const int l = cell_lin(d, c / (d.nz * d.ny), (c / d.nz) % d.ny, c % d.nz);
    cell_info[l] = (cell_info[l] & 1) | ((slot + 1) << 1);
  }
}
__global__ void __launch_bounds__(ALIAS_TB) k_alias_fill(Dev d, const int *cell_info, uint16_t *alias) {
  __shared__ int hs[125][ALIAS_TB];
  int c = blockIdx.x * ALIAS_TB + threadIdx.x, t = threadIdx.x;
  if (c >= d.ncell) return;
  int info = cell_info[cell_lin(d, c / (d.nz * d.ny), (c / d.nz) % d.ny, c % d.nz)];
  if (!(info >> 1)) return;
  alias_hashes(d, c, hs);
  uint16_t *e = alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE;
  int n = 0;
  for (int a = 0; a < 125; a++) {
    int ha = hs[a][t], m = 0;
    if (ha < 0) continue;
    for (int b = 0; b < 125; b++) m += (hs[b][t] == ha);
    if (a == 62) e[1] = (uint16_t)m;  // multiplicity of the particle's own cell
    if (m > 1) e[2 + n++] = (uint16_t)(a | ((m - 1) << 8));
  }
  e[0] = (uint16_t)n;
}

// ---- raw list length (HashGrid.neighborCount) --------------------------------------------
// faithful: count_i = sum over the in-box walked cells c of gridCount[hash(c)]  -  (visits of the own bucket)
// (ref: HashGrid.py:79-106).  cellval[c] = gridCount[hash(c)], then a separable 5x5x5 box sum in LDS.
// gridCount = static boundary particles + this step's liquid particles (both global sums on a slab decomposition);
// the cap applies to every bucket, not only walked ones
// bcount8: the same counts as bytes (saturated at 255) -- a 3 MB table that stays in the 4 MB L2 of every XCD, where the
// 12 MB of ints do not: k_cellval_occ's gathers are random (used while max_in_grid <= 255: every legal count fits)
// The liquid histogram is zeroed as it is consumed (the next grid update needs no 12 MB memset).
// Four buckets per thread (16-byte loads and stores, one 4-byte store of the four byte counts); the int table is written
// past the caches (read again only on demand: gridCount.to_numpy(), the bucket rows, the slab histogram's scratch).
__global__ void __launch_bounds__(BLOCK) k_bucket_sum(Dev d, const int *bs, int *bl, int *bcount, unsigned char *bcount8, int *flags) {
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int T = d.table_size;
  if (b >= T) return;
  bool over = false;
  if (b + 4 <= T) {
    const int4 l = *reinterpret_cast<const int4 *>(bl + b), s = *reinterpret_cast<const int4 *>(bs + b);
    if (l.x | l.y | l.z | l.w) *reinterpret_cast<int4 *>(bl + b) = make_int4(0, 0, 0, 0);
    const int4 v = make_int4(s.x + l.x, s.y + l.y, s.z + l.z, s.w + l.w);
    __builtin_nontemporal_store(v.x, bcount + b); __builtin_nontemporal_store(v.y, bcount + b + 1);
    __builtin_nontemporal_store(v.z, bcount + b + 2); __builtin_nontemporal_store(v.w, bcount + b + 3);
    *reinterpret_cast<unsigned *>(bcount8 + b) = (unsigned)min(v.x, 255) | ((unsigned)min(v.y, 255) << 8) | ((unsigned)min(v.z, 255) << 16) | ((unsigned)min(v.w, 255) << 24);
    over = max(max(v.x, v.y), max(v.z, v.w)) > d.max_in_grid;
  } else {
    for (int k = b; k < T; k++) {
      const int l = bl[k];
      if (l != 0) bl[k] = 0;
      const int v = bs[k] + l;
      bcount[k] = v;
      bcount8[k] = (unsigned char)min(v, 255);
      over |= v > d.max_in_grid;
    }
  }
  if (over) atomicOr(&flags[FLAG_ERR], ERRBIT_BUCKET);
}
// the int table (HashGrid.gridCount as the reference holds it) from the byte table, on demand, while the latter is the one kept
// up to date (every legal count fits a byte: the incremental path requires max_in_grid <= 200)
__global__ void __launch_bounds__(BLOCK) k_bucket_expand(const unsigned char *b8, int *bcount, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < T) bcount[b] = (int)b8[b];
}
// Restricted to where the liquid is: the grid is cut into REGIONS of BC0 x BC1 x BCZ cells (outer x middle x z of
// the tile order = 4 x 2 x 1 tiles).  k_cell_assign flags the regions that hold particles; k_cellval_occ gathers
// gridCount[hash(cell)] for every region that is, or touches, a flagged one (the hash scatters neighbouring cells all
// over the table: these random gathers are the cost, so each cell is gathered ONCE); k_ncount_box loads a flagged
// region and its 2-cell halo into LDS, runs the three 5-tap sums there and writes neighborCount of the particles in
// its cells.  The cost follows the liquid's volume, not its bounding box (a separable five-launch version over the
// occupied box, round 1: 100 us at C3 at rest, 290 us once the dam has spread over the domain).
// one block per region that is or touches an occupied one: cellval[plain index] = gridCount[hash(cell)]
// A region that only TOUCHES occupied ones gathers just the cells k_ncount_box will read from it: those within two cells of
// an occupied neighbour (a face, edge or corner shell instead of the whole 1024 cells; a third of the launch's regions at C3 at
// rest are of that kind, more once the liquid is a thin layer).
template <class T>
__global__ void __launch_bounds__(BLOCK) k_cellval_occ(Dev d, const T *bcount, const int *occ, int *cellval) {
  const Regions R = regions_of(d);
  const int bz = blockIdx.x % R.nbz, q = blockIdx.x / R.nbz, r1 = q % R.nb1, r0 = q / R.nb1;
  __shared__ unsigned occ27;   // bit (e0 + 1) * 9 + (e1 + 1) * 3 + (ez + 1): the neighbour region (e0, e1, ez) holds particles
  if (threadIdx.x == 0) occ27 = 0;
  __syncthreads();
  if (threadIdx.x < 27) {
    const int e0 = r0 + threadIdx.x / 9 - 1, e1 = r1 + (threadIdx.x / 3) % 3 - 1, ez = bz + threadIdx.x % 3 - 1;
    if ((unsigned)e0 < (unsigned)R.nb0 && (unsigned)e1 < (unsigned)R.nb1 && (unsigned)ez < (unsigned)R.nbz && occ[(e0 * R.nb1 + e1) * R.nbz + ez])
      atomicOr(&occ27, 1u << threadIdx.x);
  }
  __syncthreads();
  const unsigned m = occ27;
  if (!m) return;
  const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
#pragma unroll
  for (int k = 0; k < BC0 * BC1 * BCZ / BLOCK; k++) {
    const int t = threadIdx.x + k * BLOCK;
    const int lz = t % BCZ, u = t / BCZ, l1 = u % BC1, l0 = u / BC1;
    const int z = bz * BCZ + lz, a1 = r1 * BC1 + l1, a0 = r0 * BC0 + l0;
    // which neighbour regions' 2-cell halos this cell lies in: per axis bit 0 = the lower neighbour, 1 = this region, 2 = the upper one
    const unsigned mz = 2u | (lz < 2 ? 1u : 0u) | (lz >= BCZ - 2 ? 4u : 0u), m1 = 2u | (l1 < 2 ? 1u : 0u) | (l1 >= BC1 - 2 ? 4u : 0u),
                   m0 = 2u | (l0 < 2 ? 1u : 0u) | (l0 >= BC0 - 2 ? 4u : 0u);
    const unsigned s1 = ((m1 & 1u) ? mz : 0u) | ((m1 & 2u) ? mz << 3 : 0u) | ((m1 & 4u) ? mz << 6 : 0u);
    const unsigned sel = ((m0 & 1u) ? s1 : 0u) | ((m0 & 2u) ? s1 << 9 : 0u) | ((m0 & 4u) ? s1 << 18 : 0u);
    if ((m & sel) && a0 < n0 && a1 < n1 && z < d.nz) {
      const int x = d.perm ? a1 : a0, y = d.perm ? a0 : a1;
      const int v = (int)bcount[ref_hash(x + d.x0, y, z, d.table_size)];
      cellval[(x * d.ny + y) * d.nz + z] = v;
      // the reference's cap (HashGrid.py:72-74) where the table is updated in place: every cell that holds a particle is read here
      if (v > d.max_in_grid) atomicOr(&d.flags[FLAG_ERR], ERRBIT_BUCKET);
    }
  }
}
// clean neighbour mode: the cell's own occupancy (liquid + boundary particles) instead of gridCount[hash(cell)]; the same
// regions, the same shell rule, the reference's cap on the cell (ref: HashGrid.py:72-74 without the aliasing)
__global__ void __launch_bounds__(BLOCK) k_cellval_clean(Dev d, const int *occ, int *cellval) {
  const Regions R = regions_of(d);
  const int bz = blockIdx.x % R.nbz, q = blockIdx.x / R.nbz, r1 = q % R.nb1, r0 = q / R.nb1;
  __shared__ unsigned occ27;
  if (threadIdx.x == 0) occ27 = 0;
  __syncthreads();
  if (threadIdx.x < 27) {
    const int e0 = r0 + threadIdx.x / 9 - 1, e1 = r1 + (threadIdx.x / 3) % 3 - 1, ez = bz + threadIdx.x % 3 - 1;
    if ((unsigned)e0 < (unsigned)R.nb0 && (unsigned)e1 < (unsigned)R.nb1 && (unsigned)ez < (unsigned)R.nbz && occ[(e0 * R.nb1 + e1) * R.nbz + ez])
      atomicOr(&occ27, 1u << threadIdx.x);
  }
  __syncthreads();
  const unsigned m = occ27;
  if (!m) return;
  const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
#pragma unroll
  for (int k = 0; k < BC0 * BC1 * BCZ / BLOCK; k++) {
    const int t = threadIdx.x + k * BLOCK;
    const int lz = t % BCZ, u = t / BCZ, l1 = u % BC1, l0 = u / BC1;
    const int z = bz * BCZ + lz, a1 = r1 * BC1 + l1, a0 = r0 * BC0 + l0;
    const unsigned mz = 2u | (lz < 2 ? 1u : 0u) | (lz >= BCZ - 2 ? 4u : 0u), m1 = 2u | (l1 < 2 ? 1u : 0u) | (l1 >= BC1 - 2 ? 4u : 0u),
                   m0 = 2u | (l0 < 2 ? 1u : 0u) | (l0 >= BC0 - 2 ? 4u : 0u);
    const unsigned s1 = ((m1 & 1u) ? mz : 0u) | ((m1 & 2u) ? mz << 3 : 0u) | ((m1 & 4u) ? mz << 6 : 0u);
    const unsigned sel = ((m0 & 1u) ? s1 : 0u) | ((m0 & 2u) ? s1 << 9 : 0u) | ((m0 & 4u) ? s1 << 18 : 0u);
    if ((m & sel) && a0 < n0 && a1 < n1 && z < d.nz) {
      const int x = d.perm ? a1 : a0, y = d.perm ? a0 : a1;
      const int c = cell_lin(d, x, y, z);
      // the liquid cell table is current only within two cells of the occupied box (the scan skips the rest: scan_block_active);
      // a cell farther out holds no liquid particle and is read by no box sum of one that does
      const int *bb = d.bbox;
      const bool near = x >= bb[0] - 2 && x <= bb[3] + 2 && y >= bb[1] - 2 && y <= bb[4] + 2 && z >= bb[2] - 2 && z <= bb[5] + 2;
      const int v = (near ? d.cs_l[c + 1] - d.cs_l[c] : 0) + d.cs_s[c + 1] - d.cs_s[c];
      cellval[(x * d.ny + y) * d.nz + z] = v;
      if (v > d.max_in_grid) atomicOr(&d.flags[FLAG_ERR], ERRBIT_BUCKET);
    }
  }
}
__global__ void __launch_bounds__(BLOCK) k_ncount_box(Dev d, const int *cellval, const int *occ) {
  __shared__ int b0[(BC0 + 4) * (BC1 + 4) * (BCZ + 4)];
  __shared__ int b1[(BC0 + 4) * (BC1 + 4) * BCZ];
  const Regions R = regions_of(d);
  const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
  const int bz = blockIdx.x % R.nbz, bq = blockIdx.x / R.nbz;
  const int o0 = (bq / R.nb1) * BC0, o1 = (bq % R.nb1) * BC1, oz = bz * BCZ;   // first cell of the region
  const int *bb = d.bbox;
  // particles outside the grid box (the last bin of the sort) have no list at all
  if (blockIdx.x == 0) for (int j = d.cs_l[d.ncell_p] + threadIdx.x; j < d.nl; j += BLOCK) d.ncount[j] = 0;
  if (!occ[blockIdx.x]) return;   // flagged by k_cell_assign; cleared below, once every thread has passed this test
  const int l0 = d.perm ? bb[1] : bb[0], h0 = d.perm ? bb[4] : bb[3], l1 = d.perm ? bb[0] : bb[1], h1 = d.perm ? bb[3] : bb[4];
  // gridCount of the region's cells + halo (0 outside the grid: the reference walks in-box cells only)
  constexpr int E0 = BC0 + 4, E1 = BC1 + 4, EZ = BCZ + 4;
  {
    constexpr int NG = (E0 * E1 * EZ + BLOCK - 1) / BLOCK;
    int v[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
      const int t = threadIdx.x + k * BLOCK;
      const int z = oz - 2 + t % EZ, q = t / EZ, a1 = o1 - 2 + q % E1, a0 = o0 - 2 + q / E1;
      v[k] = 0;
      if (t < E0 * E1 * EZ && (unsigned)a0 < (unsigned)n0 && (unsigned)a1 < (unsigned)n1 && (unsigned)z < (unsigned)d.nz)
        v[k] = cellval[((d.perm ? a1 : a0) * d.ny + (d.perm ? a0 : a1)) * d.nz + z];
    }
#pragma unroll
    for (int k = 0; k < NG; k++) if (threadIdx.x + k * BLOCK < E0 * E1 * EZ) b0[threadIdx.x + k * BLOCK] = v[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < E0 * E1 * BCZ; t += BLOCK) {   // along z
    const int z = t % BCZ, q = t / BCZ;
    const int *r = b0 + q * EZ + z;
    b1[t] = r[0] + r[1] + r[2] + r[3] + r[4];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < E0 * BC1 * BCZ; t += BLOCK) {   // along the middle axis -> b0 as [E0][BC1][BCZ]
    const int z = t % BCZ, q = t / BCZ, a1 = q % BC1, a0 = q / BC1;
    const int *r = b1 + (a0 * E1 + a1) * BCZ + z;
    b0[t] = r[0] + r[BCZ] + r[2 * BCZ] + r[3 * BCZ] + r[4 * BCZ];
  }
  __syncthreads();
  // along the outer axis, then the particles of the cell (the cell-table and cell_info loads of a thread's cells are
  // issued together)
  constexpr int NC = BC0 * BC1 * BCZ / BLOCK;
  int ja[NC], jb[NC], info[NC];
  const int zmin = bb[2], zmax = bb[5];
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int t = threadIdx.x + k * BLOCK;
    const int zl = t % BCZ, q = t / BCZ, a1l = q % BC1, a0l = q / BC1;
    const int a0 = o0 + a0l, a1 = o1 + a1l, z = oz + zl;
    ja[k] = jb[k] = info[k] = 0;
    if (a0 > h0 || a0 < l0 || a1 > h1 || a1 < l1 || z > zmax || z < zmin) continue;   // outside the occupied box: no particle
    const int lin = cell_lin(d, d.perm ? a1 : a0, d.perm ? a0 : a1, z);
    ja[k] = d.cs_l[lin]; jb[k] = d.cs_l[lin + 1]; info[k] = d.cell_info[lin];
  }
#pragma unroll
  for (int k = 0; k < NC; k++) {
    if (jb[k] <= ja[k]) continue;
    const int t = threadIdx.x + k * BLOCK;
    const int zl = t % BCZ, q = t / BCZ, a1l = q % BC1, a0l = q / BC1;
    const int *r = b0 + (a0l * BC1 + a1l) * BCZ + zl;
    constexpr int S = BC1 * BCZ;
    const int sum = r[0] + r[S] + r[2 * S] + r[3 * S] + r[4 * S];
    const int self = (info[k] >> 1) ? d.alias[(size_t)((info[k] >> 1) - 1) * ALIAS_STRIDE + 1] : 1;
    const int cnt = sum - self;
    if (cnt > d.max_neighbour) atomicOr(&d.flags[FLAG_ERR], ERRBIT_NEIGH);  // ref: HashGrid.py:100-103 (Q3)
    for (int j = ja[k]; j < jb[k]; j++) d.ncount[j] = cnt;
  }
  if (threadIdx.x == 0) const_cast<int *>(occ)[blockIdx.x] = 0;   // k_cellval_occ, the other reader, has finished
}
// Compact per-step neighbour lists: the in-support entries of the reference's neighbour row, with the
// bucket-multiset multiplicity expanded into repeated entries (ref: HashGrid.py:88-106).
// (Measured and rejected, r02: parking the plane's 20 cell-table ranges in LDS and testing candidates four at a time
// from a runtime loop over the pieces -- 375 us against 327 us for this form at C3: the kernel issues ~160 M VALU
// wave-instructions either way and half-empty 4-wide chunks add more than the shorter dependency chains save.)
__global__ void __launch_bounds__(BLOCK) k_build_lists(Dev d, int liquid_too) {
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  const int head = liquid_too ? 0 : d.nhead[i];
  if (!owned(d, i)) { d.nhead[i] = head & 0xffff; return; }  // ghosts never run a neighbour loop
  const float4 pi = d.pos[i];
  const size_t st = (size_t)d.nbr_stride;
  const int cap = d.nbr_cap;
  NbWriter wl, ws;
  wl.init(d.nbl + i, st, i, cap);
  ws.init(d.nbs + i, st, 0, cap);
  int sbase = -1;
  walk_cells(d, i, pi, liquid_too != 0,
             [&](int j, float3, float, float m) { for (int rep = (int)m; rep > 0; rep--) wl.entry(j); },
             [&](int j, float3, float, float m) {
               if (sbase < 0) { sbase = j; ws.run = j; }
               for (int rep = (int)m; rep > 0; rep--) ws.entry(j);
             });
  int cl = liquid_too ? wl.finish() : (head & 0xffff), cs = ws.finish();
  if (cl > cap || cs > cap) { atomicOr(&d.flags[FLAG_ERR], ERRBIT_LISTCAP); cl = min(cl, cap); cs = min(cs, cap); }
  d.nhead[i] = cl | (cs << 16);
  if (cs > 0) d.nsbase[i] = sbase;
}

// ---- slab decomposition: whole-state load / store in original (row) order -----------------------
#define STATE_W 15   /* pos3 vel3 omega3 vel_guess3 kappa kappa_v + the global particle id (int bits) */
// A row is 15 floats = 60 bytes: read or written straight from a lane it is a 60-byte-stride access (1.5 TB/s); the
// 256 rows of a block are one contiguous 15 KB piece, so they go through LDS with fully coalesced global accesses.
__global__ void __launch_bounds__(BLOCK) k_load_state(Dev d, const float *rows, int n, int first) {   // rows[t] -> particle first + t
  __shared__ float sh[BLOCK * STATE_W];
  const int t0 = blockIdx.x * BLOCK, cnt = min(BLOCK, n - t0);
  const float *src = rows + (size_t)t0 * STATE_W;
  for (int k = threadIdx.x; k < cnt * STATE_W; k += BLOCK) sh[k] = src[k];
  __syncthreads();
  if ((int)threadIdx.x >= cnt) return;
  const int i = first + t0 + threadIdx.x;
  const float *r = sh + threadIdx.x * STATE_W;
  d.pos[i] = make_float4(r[0], r[1], r[2], __int_as_float(-1));
  d.vel[i] = make_float4(r[3], r[4], r[5], 0);
  d.omega[i] = make_float4(r[6], r[7], r[8], 0);
  d.vel_guess[i] = make_float4(r[9], r[10], r[11], 0);
  d.kappa[i] = r[12];
  d.kappa_v[i] = r[13];
  if (d.gid) d.gid[i] = __float_as_int(r[14]);
  d.oid[i] = i;
  d.where[i] = i;
}
__global__ void __launch_bounds__(BLOCK) k_store_state(Dev d, float *rows) {   // one thread per ORIGINAL (row) index
  __shared__ float sh[BLOCK * STATE_W];
  const int o0 = blockIdx.x * BLOCK, cnt = min(BLOCK, d.n_owned - o0);
  if (cnt <= 0) return;
  if ((int)threadIdx.x < cnt) {
    const int i = d.where[o0 + threadIdx.x];
    float *r = sh + threadIdx.x * STATE_W;
    const float4 p = d.pos[i], v = d.vel[i], w = d.omega[i], g = d.vel_guess[i];
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = v.x; r[4] = v.y; r[5] = v.z; r[6] = w.x; r[7] = w.y; r[8] = w.z;
    r[9] = g.x; r[10] = g.y; r[11] = g.z; r[12] = d.kappa[i]; r[13] = d.kappa_v[i];
    r[14] = __int_as_float(d.gid ? d.gid[o0 + threadIdx.x] : o0 + (int)threadIdx.x);
  }
  __syncthreads();
  float *dst = rows + (size_t)o0 * STATE_W;
  for (int k = threadIdx.x; k < cnt * STATE_W; k += BLOCK) dst[k] = sh[k];
}

// ---- slab state resident in the library (api_slab.h wcsph_state_*): only the rows that travel leave it --------------------
// rows (STATE_W floats) of the listed ORIGINAL indices
__global__ void __launch_bounds__(BLOCK) k_state_gather_rows(Dev d, const long long *idx, int n, float *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int o = (int)idx[k], i = d.where[o];
  const float4 p = d.pos[i], v = d.vel[i], w = d.omega[i], g = d.vel_guess[i];
  float *r = out + (size_t)k * STATE_W;
  r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = v.x; r[4] = v.y; r[5] = v.z; r[6] = w.x; r[7] = w.y; r[8] = w.z;
  r[9] = g.x; r[10] = g.y; r[11] = g.z; r[12] = d.kappa[i]; r[13] = d.kappa_v[i];
  r[14] = __int_as_float(d.gid ? d.gid[o] : o);
}
// original row src[k] becomes original row dst[k] (a staying row of the shrinking table's tail takes the index of a row that left):
// the two rows swap labels, no data moves.  The sources and the destinations are disjoint sets.
__global__ void __launch_bounds__(BLOCK) k_state_relabel(Dev d, const long long *dst, const long long *src, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int t = (int)src[k], h = (int)dst[k];
  const int st = d.where[t], sh = d.where[h];
  d.where[h] = st; d.oid[st] = h;
  d.where[t] = sh; d.oid[sh] = t;
  if (d.gid) d.gid[h] = d.gid[t];
}
// rows[k] -> original row (idx ? idx[k] : first + k): into the slot that original index has (a row that left, an old ghost) or,
// beyond the nl_old rows the table had, into a new slot of its own
__global__ void __launch_bounds__(BLOCK) k_state_write_rows(Dev d, const long long *idx, int first, const float *rows, int n, int nl_old) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int o = idx ? (int)idx[k] : first + k;
  int i = o;
  if (o < nl_old) i = d.where[o]; else { d.where[o] = o; d.oid[o] = o; }
  const float *r = rows + (size_t)k * STATE_W;
  d.pos[i] = make_float4(r[0], r[1], r[2], __int_as_float(-1));
  d.vel[i] = make_float4(r[3], r[4], r[5], 0);
  d.omega[i] = make_float4(r[6], r[7], r[8], 0);
  d.vel_guess[i] = make_float4(r[9], r[10], r[11], 0);
  d.kappa[i] = r[12];
  d.kappa_v[i] = r[13];
  if (d.gid) d.gid[o] = __float_as_int(r[14]);
}

// ---- original-order <-> sorted-order field transfer ---------------------------------------
__global__ void k_unpermute3(const float4 *f, const int *oid, int n, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 v = f[i];
  int o = oid[i];
  out[o * 3 + 0] = v.x; out[o * 3 + 1] = v.y; out[o * 3 + 2] = v.z;
}
__global__ void k_unpermute1(const float *f, const int *oid, int n, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[oid[i]] = f[i];
}
__global__ void k_permute3(float4 *f, const int *where, int n, const float *in) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  int i = where[o];
  float4 v = f[i];  // keep .w (packed cell of pos)
  v.x = in[o * 3 + 0]; v.y = in[o * 3 + 1]; v.z = in[o * 3 + 2];
  f[i] = v;
}
__global__ void k_permute1(float *f, const int *where, int n, const float *in) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  f[where[o]] = in[o];
}
// slab decomposition, once per step: which owned rows (state rows in the caller's order, x in column 0) sit within `ghost`
// cell columns of a cut with a neighbour (class 0 / 1: left / right -- they are ghosts over there), which have crossed it
// (2 / 3: they change owner), which are further than a ghost layer beyond it (4: CFL broken).  One pass; the row indices
// of classes 0..3 are appended to four lists.  A workgroup takes CLS_ROWS rows per thread, counts its classes in LDS and
// reserves its share of each list with ONE device atomic per class (one per wave was 125 k atomics on five addresses at
// 2 M rows: 1 ms); order inside a list is arbitrary.  Cell column exactly as the grid update forms it (ref: HashGrid.py:68).
constexpr int CLS_ROWS = 16;
// where != nullptr: the rows are the context's own (x of original row i = rows[where[i] * W])
__global__ void __launch_bounds__(BLOCK) k_slab_classify(const float *rows, long long n, int W, float minx, float inv_grid, int lo, int hi, int ghost,
                                                         int has_l, int has_r, long long *idx, int cap, int *counts, const int *where = nullptr,
                                                         const int *pred = nullptr) {   // pred: as Dev::pred (the slab step's speculative tail)
  if (pred && *pred == 0) return;
  __shared__ int cnt[5], base[5];
  if (threadIdx.x < 5) cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long first = (long long)blockIdx.x * (BLOCK * CLS_ROWS) + threadIdx.x;
  unsigned long long cls_of = 0;      // 4 bits per row: class + 1, 0 = interior
  int at[CLS_ROWS];
#pragma unroll
  for (int r = 0; r < CLS_ROWS; r++) {
    const long long i = first + (long long)r * BLOCK;
    int cls = -1;
    if (i < n) {
      const int cx = (int)__fmul_rn(__fsub_rn(rows[(where ? (long long)where[i] : i) * W], minx), inv_grid);
      if (has_l) { if (cx < lo + ghost) cls = 0; if (cx < lo) cls = 2; if (cx < lo - ghost) cls = 4; }
      if (has_r) { if (cx >= hi - ghost) cls = 1; if (cx >= hi) cls = 3; if (cx >= hi + ghost) cls = 4; }
    }
    at[r] = cls >= 0 ? atomicAdd(&cnt[cls], 1) : 0;
    cls_of |= (unsigned long long)(cls + 1) << (4 * r);
  }
  __syncthreads();
  if (threadIdx.x < 5) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]) : 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < CLS_ROWS; r++) {
    const int cls = (int)((cls_of >> (4 * r)) & 15) - 1;
    if (cls >= 0 && cls < 4) {
      const int o = base[cls] + at[r];
      if (o < cap) idx[(size_t)cls * cap + o] = first + (long long)r * BLOCK;
    }
  }
}

// the same classification over the context's OWN rows, walked in SORTED order: pos[i] and oid[i] are read coalesced (the form above
// reaches a row through where[]: a 16-byte gather per row, 23 us at 2 M rows against ~10); a row's label is its original index
// oid[i], ghost rows (label >= n_owned) are skipped.  The lists hold the same sets; their order is arbitrary in both forms.
constexpr int CLS_ROWS_SORTED = 4;   // (16 rows per thread leave 560 workgroups at 2.3 M rows: 2 waves per SIMD, latency-bound; 4: 2240)
__global__ void __launch_bounds__(BLOCK) k_state_classify_sorted(const float4 *pos, const int *oid, int nl, int n_owned, float minx, float inv_grid, int lo, int hi, int ghost,
                                                                 int has_l, int has_r, long long *idx, int cap, int *counts, const int *pred) {
  if (pred && *pred == 0) return;
  __shared__ int cnt[5], base[5];
  if (threadIdx.x < 5) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int first = blockIdx.x * (BLOCK * CLS_ROWS_SORTED) + threadIdx.x;
  unsigned long long cls_of = 0;      // 4 bits per row: class + 1, 0 = interior
  int at[CLS_ROWS_SORTED], lab[CLS_ROWS_SORTED];
  float xs[CLS_ROWS_SORTED];
  // every load of the thread is issued before anything is classified (label and position of a row do not wait for each other)
#pragma unroll
  for (int r = 0; r < CLS_ROWS_SORTED; r++) {
    const int i = first + r * BLOCK;
    lab[r] = i < nl ? oid[i] : 0x7fffffff;
    xs[r] = i < nl ? reinterpret_cast<const float *>(pos + i)[0] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < CLS_ROWS_SORTED; r++) {
    int cls = -1;
    if (lab[r] < n_owned) {
      const int cx = (int)__fmul_rn(__fsub_rn(xs[r], minx), inv_grid);
      if (has_l) { if (cx < lo + ghost) cls = 0; if (cx < lo) cls = 2; if (cx < lo - ghost) cls = 4; }
      if (has_r) { if (cx >= hi - ghost) cls = 1; if (cx >= hi) cls = 3; if (cx >= hi + ghost) cls = 4; }
    }
    // in sorted order the rows next to a cut sit side by side: a wave of them would queue 64 LDS atomics on ONE counter per row
    // (the whole kernel then runs at the pace of its few boundary blocks).  One atomic per wave and class instead.
    at[r] = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == c);
      if (m) {     // wave-uniform
        const int lane = threadIdx.x & (WAVE - 1), leader = __builtin_ctzll(m);
        int b = 0;
        if (lane == leader) b = atomicAdd(&cnt[c], __builtin_popcountll(m));
        b = __shfl(b, leader, WAVE);
        if (cls == c) at[r] = b + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      }
    }
    cls_of |= (unsigned long long)(cls + 1) << (4 * r);
  }
  __syncthreads();
  if (threadIdx.x < 5) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]) : 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < CLS_ROWS_SORTED; r++) {
    const int cls = (int)((cls_of >> (4 * r)) & 15) - 1;
    if (cls >= 0 && cls < 4) {
      const int o = base[cls] + at[r];
      if (o < cap) idx[(size_t)cls * cap + o] = lab[r];
    }
  }
}

// halo pack / unpack of SEVERAL fields in one launch: out[k][W] = the listed fields of original row idx[k], side by side
struct HaloFields { int n; const float *ptr[8]; int width[8], stride[8]; int W; const int *pred; };   // pred: as Dev::pred (speculative slab step)
__global__ void k_halo_pack(HaloFields h, const int *where, const int *idx, int n, float *out) {
  if (h.pred && *h.pred == 0) return;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * h.W) return;
  const int k = t / h.W;
  int c = t % h.W, f = 0;
  while (c >= h.width[f]) { c -= h.width[f]; f++; }
  out[t] = h.ptr[f][(size_t)where[idx[k]] * h.stride[f] + c];
}
__global__ void k_halo_unpack(HaloFields h, const int *where, const int *idx, int n, const float *in) {
  if (h.pred && *h.pred == 0) return;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * h.W) return;
  const int k = t / h.W;
  int c = t % h.W, f = 0;
  while (c >= h.width[f]) { c -= h.width[f]; f++; }
  if (!h.ptr[f]) return;   // (a second set of destinations may leave columns out)
  const_cast<float *>(h.ptr[f])[(size_t)where[idx[k]] * h.stride[f] + c] = in[t];
}
// rows by explicit ORIGINAL index list (single field)
__global__ void k_gather_rows3(const float4 *f, const int *where, const int *idx, int n, float *out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float4 v = f[where[idx[k]]];
  out[k * 3 + 0] = v.x; out[k * 3 + 1] = v.y; out[k * 3 + 2] = v.z;
}
__global__ void k_gather_rows1(const float *f, const int *where, const int *idx, int n, float *out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  out[k] = f[where[idx[k]]];
}
__global__ void k_scatter_rows3(float4 *f, const int *where, const int *idx, int n, const float *in) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int i = where[idx[k]];
  float4 v = f[i];
  v.x = in[k * 3 + 0]; v.y = in[k * 3 + 1]; v.z = in[k * 3 + 2];
  f[i] = v;
}
__global__ void k_scatter_rows1(float *f, const int *where, const int *idx, int n, const float *in) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  f[where[idx[k]]] = in[k];
}
