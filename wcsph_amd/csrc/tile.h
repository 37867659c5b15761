// tile.h -- the two neighbour passes of pressure_iter (ref: dfsph.py:518-547 over HashGrid.py:79-106) with the neighbourhood
// staged in LDS.
//
// A GROUP is TG0 x TG1 x 16 cells (2 x 2 x 1 tiles of the cell order, device.h cell_lin: its owners are TL_RUNS contiguous
// particle ranges); its BOX is the group grown by the walk radius (2 cells) on every side, TL_NBOX cells.  Once per grid update
// k_tile_build numbers the particles of every occupied group's box (box slot = position in (outer, middle, z) cell order, so
// the <= 5 cells of one row of a particle's 5x5x5 walk are ONE slot range), writes the slot -> particle table and builds, from
// the box's positions in LDS, every owner's neighbour list as 16-bit LDS byte offsets in the order of the gather lists
// (grid.h k_build_lists: the same entries, the same order -- the two paths sum bit-identically).  A pass then runs as
// persistent workgroups over the occupied groups: stage the box's records with LDS-DMA (global_load_lds_dwordx4 through the
// slot table), barrier, every owner walks its list with ds_read_b128.  Solid entries keep the gather walk (device.h
// walk_solids: the boundary is static and sorted once; only near-wall particles have any).  A group whose box holds more than
// TL_CAP particles (crowded cells: sesph.py's scenes) and the particles outside the grid box keep the gather walk altogether.
#pragma once
#include "device.h"
#include "grid.h"

#ifndef GT0B
#define GT0B 1   /* tiles per group along the outer axis, log2 */
#endif
#ifndef GT1B
#define GT1B 1   /* ... along the middle axis */
#endif
#define TG0 ((1 << GT0B) << T0B)
#define TG1 ((1 << GT1B) << T1B)
#define TGZ (1 << T2B)
#define TB0 (TG0 + 4)
#define TB1 (TG1 + 4)
#define TBZ (TGZ + 4)
#define TL_NBOX (TB0 * TB1 * TBZ)
#define TL_RUNS (1 << (GT0B + GT1B))
#ifndef TL_CAP
#define TL_CAP 2496   /* box slots, a multiple of 64 (whole-wave DMA chunks): 16 B x TL_CAP x 4 workgroups (velocity pass) / 32 B x TL_CAP x 2 (density-change pass) + their static LDS fill the CU's 160 KiB */
#endif
#ifndef TILE_NT_V
#define TILE_NT_V 640   /* threads per workgroup (a full group owns ~550 particles: one round), workgroups per CU: velocity pass */
#define TILE_WG_V 2
#endif
#ifndef TILE_NT_D
#define TILE_NT_D 640   /* ... density-change pass */
#define TILE_WG_D 2
#endif
#define TL_DESC (4 + 2 * TL_RUNS)   /* ints per group descriptor: packed group, slots (-1: no box), table offset, owners, run starts, cumulative run ends */
static_assert(TL_CAP * 16 < 65536 && TL_CAP % 64 == 0, "list entries are 16-bit LDS byte offsets; the staging DMA writes whole 64-slot chunks");

struct TileGroup { int G0, G1, t2; };
__device__ __forceinline__ TileGroup tile_group(const Dev &d, int g) {
  TileGroup t;
  t.t2 = g % d.nt2;
  const int q = g / d.nt2;
  t.G1 = q % d.tg1n; t.G0 = q / d.tg1n;
  return t;
}
// owners of run r of a group (one tile of the cell order): particle range [*start, *start + n).  Tiles outside the bounding box of
// this step's occupied cells hold nobody, and the cell table that far out is not maintained (grid.h scan_block_active).
__device__ __forceinline__ int tile_run(const Dev &d, const int *bb, TileGroup t, int r, int *start) {
  const int t0 = (t.G0 << GT0B) + (r >> GT1B), t1 = (t.G1 << GT1B) + (r & ((1 << GT1B) - 1));
  *start = 0;
  if (t0 >= d.nt0 || t1 >= d.nt1 || bb[3] < 0) return 0;
  const int lo0 = d.perm ? bb[1] : bb[0], hi0 = d.perm ? bb[4] : bb[3], lo1 = d.perm ? bb[0] : bb[1], hi1 = d.perm ? bb[3] : bb[4];
  if ((t0 << T0B) > hi0 || ((t0 + 1) << T0B) <= lo0 || (t1 << T1B) > hi1 || ((t1 + 1) << T1B) <= lo1 || (t.t2 << T2B) > bb[5] || ((t.t2 + 1) << T2B) <= bb[2]) return 0;
  const int T = (t0 * d.nt1 + t1) * d.nt2 + t.t2;
  const int a = d.cs_l[T << TILE_SHIFT];
  *start = a;
  return d.cs_l[(T + 1) << TILE_SHIFT] - a;
}

// ---- per grid update: which groups hold particles, in group order ----------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_tile_mark(Dev d) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.tl_ngroups) return;
  const TileGroup t = tile_group(d, g);
  int own = 0, st;
#pragma unroll
  for (int r = 0; r < TL_RUNS; r++) own += tile_run(d, d.bbox, t, r, &st);
  d.tl_own[g] = own;
}
__global__ void __launch_bounds__(1024) k_tile_scan(Dev d) {
  __shared__ int part[1024 / WAVE];
  const int n = d.tl_ngroups, per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n);
  int c = 0;
  for (int k = lo; k < hi; k++) c += d.tl_own[k] > 0;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = c;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) { const int t = __shfl_up(inc, o, WAVE); if (lane >= o) inc += t; }
  if (lane == WAVE - 1) part[w] = inc;
  __syncthreads();
  int a = inc - c;
  for (int k = 0; k < w; k++) a += part[k];
  for (int k = lo; k < hi; k++) if (d.tl_own[k] > 0) d.tl_act[a++] = k;
  if (threadIdx.x == 1023) { d.tl_n[0] = a; d.tl_n[1] = 0; d.tl_n[2] = 0; }
}

// persistent workgroups: XCD x (= blockIdx % 8) takes the x-th eighth of the occupied groups, in group order -- neighbouring groups
// share most of their boxes, and an XCD's L2 then holds one spatial region (as sorted_index() does for the gather passes)
#define TILE_GROUP_LOOP(d, a)                                                                              \
  const int na_ = (d).tl_n[0];                                                                             \
  const int xcd_ = blockIdx.x & 7, per_ = gridDim.x >> 3;                                                  \
  const int alo_ = (int)((long)na_ * xcd_ / 8), ahi_ = (int)((long)na_ * (xcd_ + 1) / 8);                   \
  for (int a = alo_ + (blockIdx.x >> 3); a < ahi_; a += per_)

// builder side of a slot list: shift register of eight 16-bit entries, flushed as whole 16-byte blocks (cf. NbWriter)
struct TlWriter {
  uint4 *base;
  size_t stride;
  int slots, cap;
  unsigned w0, w1, w2, w3;
  __device__ __forceinline__ void init(uint4 *b, size_t st, int cap_) { base = b; stride = st; slots = 0; cap = cap_; w0 = w1 = w2 = w3 = 0; }
  __device__ __forceinline__ void entry(unsigned v) {
    w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 = __builtin_amdgcn_alignbit(w2, w1, 16);
    w2 = __builtin_amdgcn_alignbit(w3, w2, 16); w3 = (w3 >> 16) | (v << 16);
    slots++;
    if ((slots & 7) == 0 && slots <= cap) base[(size_t)((slots >> 3) - 1) * stride] = make_uint4(w0, w1, w2, w3);
  }
  __device__ __forceinline__ int finish() {
    const int n = slots;
    if (n <= cap) while (slots & 7) entry(0);
    return n;
  }
};

// the list of owner i from the box in LDS: candidates in the order of device.h walk_cells (outer plane, middle row, z, particle)
__device__ __forceinline__ void tile_list_owner(const Dev &d, int i, const float4 *lds, const unsigned short *bst, TileGroup tg) {
  const float4 pi = d.pos[i];
  const int cw = __float_as_int(pi.w);
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const float3 xi = xyz(pi);
  const int lin = cell_lin(d, cx, cy, cz);
  const int info = d.cell_info[lin];
  const int c0 = d.perm ? cy : cx, c1 = d.perm ? cx : cy, n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
  const int b0 = tg.G0 * TG0 - 2, b1 = tg.G1 * TG1 - 2, bz = (tg.t2 << T2B) - 2;
  const int ec0 = c0 - b0, ec1 = c1 - b1;
  const int self = (int)bst[(ec0 * TB1 + ec1) * TBZ + (cz - bz)] + (i - d.cs_l[lin]);
  // row trimming exactly as walk_cells (a candidate filter: the r2 test decides)
  const float ux = (pi.x - d.minx) * d.inv_grid - (float)(cx + d.x0), uy = (pi.y - d.miny) * d.inv_grid - (float)cy;
  const float fz = (pi.z - d.minz) * d.inv_grid;
  const float rm = __builtin_amdgcn_sqrtf(d.list_r2) * d.inv_grid * 1.0001f + 1e-3f, rm2 = rm * rm;
  auto zrange = [&](int dx, int dy, int &za, int &zb) -> bool {
    const float ex = dx > 0 ? (float)dx - ux : (dx < 0 ? (float)(-dx - 1) + ux : 0.0f);
    const float ey = dy > 0 ? (float)dy - uy : (dy < 0 ? (float)(-dy - 1) + uy : 0.0f);
    const float q = rm2 - ex * ex - ey * ey;
    if (q < 0.0f) return false;
    const float zr = __builtin_amdgcn_sqrtf(q);
    za = max(max((int)floorf(fz - zr), cz - 2), 0);
    zb = min(min((int)floorf(fz + zr), cz + 2), d.nz - 1) + 1;
    return zb > za;
  };
  TlWriter w;
  w.init(d.nbt + i, (size_t)d.nbr_stride, d.nbr_cap);
#pragma unroll 1
  for (int e0 = -2; e0 <= 2; e0++) {
    const int a0 = c0 + e0;
    if ((unsigned)a0 >= (unsigned)n0) continue;
    int sa[5], sb[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int a1 = c1 + t - 2;
      int z0, z1;
      sa[t] = sb[t] = 0;
      if ((unsigned)a1 < (unsigned)n1 && zrange(d.perm ? t - 2 : e0, d.perm ? e0 : t - 2, z0, z1)) {
        const int col = ((ec0 + e0) * TB1 + (ec1 + t - 2)) * TBZ - bz;
        sa[t] = bst[col + z0]; sb[t] = bst[col + z1];
      }
    }
#pragma unroll
    for (int t = 0; t < 5; t++) {
      int s = sa[t];
      const int end = sb[t];
      if (s < end) {
        float4 pj = lds[s];
        for (;;) {
          const bool more = s + 1 < end;
          float4 pn = pj;
          if (more) pn = lds[s + 1];
          const float3 r = xi - xyz(pj);
          const float r2 = dot3(r, r);
          if (r2 <= d.list_r2 && s != self) w.entry((unsigned)s << 4);
          if (!more) break;
          pj = pn; s++;
        }
      }
    }
  }
  if (info >> 1) {   // faithful mode: walked cells that share a hash bucket are visited again (Q1; device.h walk_cells)
    const uint16_t *e = d.alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE;
    const int n = e[0];
#pragma unroll 1
    for (int t = 0; t < n; t++) {
      const int k = e[2 + t] & 0xff, m = e[2 + t] >> 8;
      const int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
      const int bc = (((d.perm ? y : x) - b0) * TB1 + ((d.perm ? x : y) - b1)) * TBZ + (z - bz);
      for (int s = bst[bc]; s < (int)bst[bc + 1]; s++) {
        const float3 r = xi - xyz(lds[s]);
        const float r2 = dot3(r, r);
        if (r2 <= d.list_r2 && s != self) for (int rep = m; rep > 0; rep--) w.entry((unsigned)s << 4);
      }
    }
  }
  int cl = w.finish();
  if (cl > d.nbr_cap) { atomicOr(&d.flags[FLAG_ERR], ERRBIT_LISTCAP); cl = d.nbr_cap; }
  d.ntl[i] = cl;
}

// owner o of a group -> sorted particle index (desc: the group's descriptor in LDS or SGPRs)
__device__ __forceinline__ int tile_owner(const int *desc, int o) {
  int r = 0, base = 0;
#pragma unroll
  for (int q = 1; q < TL_RUNS; q++) if (o >= desc[4 + TL_RUNS + q - 1]) { r = q; base = desc[4 + TL_RUNS + q - 1]; }
  return desc[4 + r] + o - base;
}

// ---- per grid update: boxes, slot tables, slot lists ----------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_tile_build(Dev d) {
  extern __shared__ float4 lds[];                                              // [TL_CAP] box positions, then
  unsigned short *bst = reinterpret_cast<unsigned short *>(lds + TL_CAP);      // [TL_NBOX + 1] first slot of every box cell
  __shared__ int desc[TL_DESC];
  constexpr int PER = (TL_NBOX + BLOCK - 1) / BLOCK;
  const int lo0 = d.perm ? d.bbox[1] : d.bbox[0], hi0 = d.perm ? d.bbox[4] : d.bbox[3], lo1 = d.perm ? d.bbox[0] : d.bbox[1], hi1 = d.perm ? d.bbox[3] : d.bbox[4];
  const int lo2 = d.bbox[2], hi2 = d.bbox[5];
  const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
  TILE_GROUP_LOOP(d, a) {
    const int g = d.tl_act[a];
    const TileGroup tg = tile_group(d, g);
    __syncthreads();   // the previous group's readers of lds / bst / desc are done
    if (threadIdx.x == 0) {
      int cum = 0;
      for (int r = 0; r < TL_RUNS; r++) { int st; cum += tile_run(d, d.bbox, tg, r, &st); desc[4 + r] = st; desc[4 + TL_RUNS + r] = cum; }
      desc[0] = tg.G0 | (tg.G1 << 10) | (tg.t2 << 20);
      desc[3] = cum;
    }
    // occupancy of the box cells, PER consecutive cells (of one or two z-columns) per thread; cells beyond the grown bounding box of
    // this step's occupied cells are empty, and the cell table is not maintained there
    const int b0 = tg.G0 * TG0 - 2, b1 = tg.G1 * TG1 - 2, bz = (tg.t2 << T2B) - 2;
    int cnt[PER], stt[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int b = threadIdx.x * PER + k;
      cnt[k] = 0; stt[k] = 0;
      if (b < TL_NBOX) {
        const int col = b / TBZ, z = bz + b % TBZ, a0 = b0 + col / TB1, a1 = b1 + col % TB1;
        if ((unsigned)a0 < (unsigned)n0 && (unsigned)a1 < (unsigned)n1 && (unsigned)z < (unsigned)d.nz && a0 >= lo0 - 2 && a0 <= hi0 + 2 && a1 >= lo1 - 2 &&
            a1 <= hi1 + 2 && z >= lo2 - 2 && z <= hi2 + 2) {
          const int lin = cell_lin(d, d.perm ? a1 : a0, d.perm ? a0 : a1, z);
          stt[k] = d.cs_l[lin];
          cnt[k] = d.cs_l[lin + 1] - stt[k];
          sum += cnt[k];
        }
      }
    }
    int tot;
    int ex = block_exclusive_scan(sum, &tot);
    if (threadIdx.x == 0) {
      const bool fits = tot <= TL_CAP;
      desc[1] = fits ? tot : -1;
      desc[2] = fits ? atomicAdd(&d.tl_n[1], (tot + 15) & ~15) : 0;
      if (!fits) atomicAdd(&d.tl_n[2], 1);
    }
    __syncthreads();
    if (threadIdx.x < TL_DESC) d.tl_grp[(size_t)a * TL_DESC + threadIdx.x] = desc[threadIdx.x];
    const int owners = desc[3];
    if (desc[1] < 0) {   // no box: its owners keep the gather walk
      for (int o = threadIdx.x; o < owners; o += BLOCK) d.ntl[tile_owner(desc, o)] = 0;
      continue;
    }
    int *tbl = d.tl_tbl + desc[2];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int b = threadIdx.x * PER + k;
      if (b < TL_NBOX) bst[b] = (unsigned short)ex;
      for (int q = 0; q < cnt[k]; q++) { lds[ex + q] = d.pos[stt[k] + q]; tbl[ex + q] = stt[k] + q; }
      ex += cnt[k];
    }
    if (threadIdx.x == 0) bst[TL_NBOX] = (unsigned short)tot;
    __syncthreads();
    for (int o = threadIdx.x; o < owners; o += BLOCK) tile_list_owner(d, tile_owner(desc, o), lds, bst, tg);
  }
}

// ---- the passes --------------------------------------------------------------------------------------------------------------
// Per group: [barrier: the previous group's readers are done] -> the box's records by LDS-DMA (global_load_lds_dwordx4: per-lane
// source address through the slot table, destination = wave base + 16 * lane; the table words were fetched while the previous
// group was computed) + every owner's own data and the first TL_PRE blocks of its list, all in flight together -> [barrier] ->
// the owners walk their lists out of registers and LDS.  One memory round trip per group; the other workgroup(s) of the CU
// compute meanwhile.
#define TL_PRE 6   /* list blocks (of eight entries) an owner loads ahead; longer lists fetch the rest on demand */
template <int NT>
struct TileIdx { int v[(TL_CAP + NT - 1) / NT]; };
template <int NT>
__device__ __forceinline__ void tile_idx_load(const Dev &d, const int *desc, TileIdx<NT> &x) {
  const int nslots = desc[1];
  const int *tbl = d.tl_tbl + desc[2];
#pragma unroll
  for (int r = 0; r < (TL_CAP + NT - 1) / NT; r++) x.v[r] = tbl[max(min(r * NT + (int)threadIdx.x, nslots - 1), 0)];
}
// lanes past the last slot re-read the last record into slots nobody refers to
template <int NT, bool TWO>
__device__ __forceinline__ void tile_stage(const int *desc, const TileIdx<NT> &x, float4 *lds, const float4 *A, const float4 *B) {
  const int nslots = desc[1];
  const int wbase = threadIdx.x & ~(WAVE - 1);
#pragma unroll
  for (int r = 0; r < (TL_CAP + NT - 1) / NT; r++) {
    if (r * NT + wbase < nslots) {   // (wave-uniform)
      __builtin_amdgcn_global_load_lds((const void *)(A + x.v[r]), (__attribute__((address_space(3))) void *)(lds + r * NT + wbase), 16, 0, 0);
      if (TWO) __builtin_amdgcn_global_load_lds((const void *)(B + x.v[r]), (__attribute__((address_space(3))) void *)(lds + TL_CAP + r * NT + wbase), 16, 0, 0);
    }
  }
}
// (the w component is pinned: a pass that uses xyz only would otherwise be narrowed to ds_read_b96, which the LDS serves in eight
// lane groups instead of four -- half the rate of ds_read_b128, MI355X_MICROARCH "LDS"; measured: 7.9 -> 4 array cycles per read)
__device__ __forceinline__ float4 lds_rec(const float4 *lds, unsigned off) {
  float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(lds) + off);
  asm volatile("" : "+v"(v.w));
  return v;
}
// an owner's list: length, and its first TL_PRE blocks whatever the length (blocks past the end hold older lists or zeros --
// valid offsets either way -- and are masked by the length)
struct TileList { int cnt; uint4 blk[TL_PRE]; };
__device__ __forceinline__ void tile_list_load(const Dev &d, int i, TileList &l) {
  l.cnt = d.ntl[i];
  const uint4 *L = d.nbt + i;
  const size_t st = (size_t)d.nbr_stride;
#pragma unroll
  for (int b = 0; b < TL_PRE; b++) l.blk[b] = L[(size_t)b * st];
}
// the liquid entries of owner i: fl(r, r2, record[, second record]); one trip = one block of eight offsets, all eight (or sixteen)
// LDS reads issued before any arithmetic
template <bool TWO, class FL>
__device__ __forceinline__ void tile_walk(const Dev &d, int i, float3 xi, const TileList &l, const float4 *lds, FL &&fl) {
  const int cnt = l.cnt;
#define TW_ONE(u, pp, qq) if (k + u < cnt) { const float3 r = xi - xyz(pp); fl(r, dot3(r, r), pp, qq); }
#define TW_HALF(k, u, oa, ob, oc, od)                                                                                                     \
  {                                                                                                                                      \
    const float4 pa = lds_rec(lds, oa), pb = lds_rec(lds, ob), pc = lds_rec(lds, oc), pd = lds_rec(lds, od);                              \
    float4 qa = pa, qb = pb, qc = pc, qd = pd;                                                                                            \
    if (TWO) { qa = lds_rec(lds + TL_CAP, oa); qb = lds_rec(lds + TL_CAP, ob); qc = lds_rec(lds + TL_CAP, oc); qd = lds_rec(lds + TL_CAP, od); } \
    TW_ONE(u, pa, qa) TW_ONE(u + 1, pb, qb) TW_ONE(u + 2, pc, qc) TW_ONE(u + 3, pd, qd)                                                   \
  }
  // one record per entry: eight reads in flight; two records: four entries (eight reads) at a time -- sixteen 16-byte results
  // would not fit the register budget of five waves per SIMD
#define TW_TRIP(e, k)                                                                                                                    \
  {                                                                                                                                      \
    const unsigned o0 = e.x & 0xffffu, o1 = e.x >> 16, o2 = e.y & 0xffffu, o3 = e.y >> 16, o4 = e.z & 0xffffu, o5 = e.z >> 16, o6 = e.w & 0xffffu, o7 = e.w >> 16; \
    if (TWO) { TW_HALF(k, 0, o0, o1, o2, o3) TW_HALF(k, 4, o4, o5, o6, o7) }                                                              \
    else {                                                                                                                               \
      const float4 p0 = lds_rec(lds, o0), p1 = lds_rec(lds, o1), p2 = lds_rec(lds, o2), p3 = lds_rec(lds, o3);                            \
      const float4 p4 = lds_rec(lds, o4), p5 = lds_rec(lds, o5), p6 = lds_rec(lds, o6), p7 = lds_rec(lds, o7);                            \
      TW_ONE(0, p0, p0) TW_ONE(1, p1, p1) TW_ONE(2, p2, p2) TW_ONE(3, p3, p3)                                                             \
      TW_ONE(4, p4, p4) TW_ONE(5, p5, p5) TW_ONE(6, p6, p6) TW_ONE(7, p7, p7)                                                             \
    }                                                                                                                                    \
  }
#pragma unroll
  for (int b = 0; b < TL_PRE; b++) {
    const int k = b * 8;
    if (k < cnt) TW_TRIP(l.blk[b], k)
  }
  if (cnt > TL_PRE * 8) {
    const size_t st = (size_t)d.nbr_stride;
    const uint4 *L = d.nbt + i;
    for (int k = TL_PRE * 8; k < cnt; k += 8) {
      const uint4 e = *(L + (size_t)(k >> 3) * st);
      TW_TRIP(e, k)
    }
  }
#undef TW_TRIP
#undef TW_HALF
#undef TW_ONE
}
// sum of v over the workgroup into stripe blockIdx % ACC_STRIPES of a reduction slot; every thread must call it
template <int NT>
__device__ __forceinline__ void tile_accumulate(double *slot, float v) {
  __shared__ float sh[NT / WAVE];
  v = wave_sum(v);
  if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NT / WAVE; k++) s += sh[k];
    atomicAdd(slot + (blockIdx.x & (ACC_STRIPES - 1)), (double)s);
  }
}
// the particles outside the grid box (the last bin of the cell table: no cell, no neighbours, Q4) belong to no group
#define TILE_OUTSIDE_LOOP(d, i) for (int i = (d).cs_l[(d).ncell_p] + blockIdx.x * blockDim.x + threadIdx.x; i < (d).nl; i += gridDim.x * blockDim.x)

// velocity pass: k_vel_from_k<0> (dfsph_kernels.h; ref: dfsph.py:529-541) -- one context owning every row
struct TileVelPre { float4 pi, v; float kap; int head; TileList l; };
__device__ __forceinline__ void tile_vel_load(const Dev &d, int i, const float4 *pk, const float *kap, TileVelPre &p) {
  p.pi = pk[i];
  p.v = d.vel[i];
  p.kap = kap ? ld_once(kap + i) : 0.0f;
  p.head = d.nhead[i];
  tile_list_load(d, i, p.l);
}
__device__ __forceinline__ void tile_vel_owner(const Dev &d, int i, const TileVelPre &p, float dt, float *kap, const float4 *lds) {
  const float ki = p.pi.w;
  if (kap) st_once(kap + i, p.kap + ki);
  float3 acc = f3(0, 0, 0);
  const bool ks = fabsf(ki) > d.eps;
  tile_walk<false>(d, i, xyz(p.pi), p.l, lds, [&](float3 r, float r2, float4 q, float4) {
    const float s = ki + q.w;
    if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
  });
  walk_solids(d, i, xyz(p.pi), (int)((unsigned)p.head >> 16), [&](int, float3 r, float r2) { if (ks) acc = acc + r * (ki * (d.VS0 / d.VL0) * cubic_gradW_scale(d, r2)); });
  float4 v = p.v;
  const float f = dt * d.VL0;
  v.x += f * acc.x; v.y += f * acc.y; v.z += f * acc.z;
  d.vel[i] = v;
}
// ... of a particle without a box: the gather walk
__device__ __forceinline__ void tile_vel_gather(const Dev &d, int i, const float4 *pk, float dt, float *kap) {
  const float4 pi = pk[i];
  const float ki = pi.w;
  if (kap) st_once(kap + i, ld_once(kap + i) + ki);
  float3 acc = f3(0, 0, 0);
  const bool ks = fabsf(ki) > d.eps;
  walk<8>(d, i, pi, pk, [](int, float4 q) { return q.w; },
          [&](int, float3 r, float r2, float kj) {
            const float s = ki + kj;
            if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) { if (ks) acc = acc + r * (ki * (d.VS0 / d.VL0) * cubic_gradW_scale(d, r2)); });
  float4 v = d.vel[i];
  const float f = dt * d.VL0;
  v.x += f * acc.x; v.y += f * acc.y; v.z += f * acc.z;
  d.vel[i] = v;
}
// the group loop shared by both passes: LOAD(i, pre) fetches an owner's data, OWNER(i, pre) computes it, GATHER(i) is the pass on the
// gather walk
#define TILE_PASS_BODY(NT, TWO, A, B, PRE_T, LOAD, OWNER, GATHER)                                                           \
  extern __shared__ float4 lds[];                                                                                          \
  const int na_ = d.tl_n[0];                                                                                               \
  const int xcd_ = blockIdx.x & 7, per_ = gridDim.x >> 3;                                                                  \
  const int alo_ = (int)((long)na_ * xcd_ / 8), ahi_ = (int)((long)na_ * (xcd_ + 1) / 8);                                   \
  int a = alo_ + (blockIdx.x >> 3);                                                                                        \
  TileIdx<NT> idx;                                                                                                         \
  if (a < ahi_) tile_idx_load<NT>(d, d.tl_grp + (size_t)a * TL_DESC, idx);                                                 \
  for (; a < ahi_; a += per_) {                                                                                            \
    const int *desc = d.tl_grp + (size_t)a * TL_DESC;   /* uniform: scalar loads */                                        \
    const int owners = desc[3];                                                                                            \
    const bool boxed = desc[1] >= 0;                                                                                       \
    __syncthreads();   /* the previous group's readers are done */                                                         \
    if (!boxed) {                                                                                                          \
      for (int o = threadIdx.x; o < owners; o += NT) { const int i = tile_owner(desc, o); GATHER(i); }                     \
      if (a + per_ < ahi_) tile_idx_load<NT>(d, desc + (size_t)per_ * TL_DESC, idx);                                       \
      continue;                                                                                                            \
    }                                                                                                                      \
    tile_stage<NT, TWO>(desc, idx, lds, A, B);                                                                             \
    int o = threadIdx.x;                                                                                                   \
    int i = tile_owner(desc, o < owners ? o : 0);                                                                          \
    PRE_T pre;                                                                                                             \
    LOAD(i, pre);                                                                                                          \
    if (a + per_ < ahi_) tile_idx_load<NT>(d, desc + (size_t)per_ * TL_DESC, idx);   /* the next group's table words */    \
    __syncthreads();   /* (the compiler drains vmcnt before the barrier: DMA landed, owner data here) */                   \
    for (;;) {                                                                                                             \
      if (o < owners) OWNER(i, pre);                                                                                       \
      o += NT;                                                                                                             \
      if (o - (int)threadIdx.x >= owners) break;   /* (uniform) */                                                         \
      i = tile_owner(desc, o < owners ? o : 0);                                                                            \
      LOAD(i, pre);                                                                                                        \
    }                                                                                                                      \
  }

template <int NT>
__global__ void __launch_bounds__(NT) k_tile_vel(Dev d, const float4 *pk, float dt, float *kap) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  if (kap) acc_clear(ACC_SLOT(d, ACC_ERR));
#define TV_LOAD(i, pre) tile_vel_load(d, i, pk, kap, pre)
#define TV_OWNER(i, pre) tile_vel_owner(d, i, pre, dt, kap, lds)
#define TV_GATHER(i) tile_vel_gather(d, i, pk, dt, kap)
  TILE_PASS_BODY(NT, false, pk, pk, TileVelPre, TV_LOAD, TV_OWNER, TV_GATHER)
  TILE_OUTSIDE_LOOP(d, i) tile_vel_gather(d, i, pk, dt, kap);
#undef TV_LOAD
#undef TV_OWNER
#undef TV_GATHER
}

// density-change pass: k_drho_press<DRHO_ITER, true> (dfsph_kernels.h; ref: dfsph.py:394-412 inlined at :545, next k as in :523-525)
struct TileDrhoPre { float4 pi, v; float rho, alpha; int head; TileList l; };
__device__ __forceinline__ void tile_drho_load(const Dev &d, int i, TileDrhoPre &p) {
  p.pi = d.pk[i];
  p.v = d.vel[i];
  p.rho = d.rho[i];
  p.alpha = d.alpha[i];
  p.head = d.nhead[i];
  tile_list_load(d, i, p.l);
}
__device__ __forceinline__ float tile_drho_finish(const Dev &d, int i, float rho, float alpha, float dt, float t) {
  const float a = fmaxf(1.0f, rho / d.rho_L0 + dt * t);
  d.adv_rho[i] = a;
  const float e = a - 1.0f;
  reinterpret_cast<float *>(d.pk + i)[3] = e * alpha;
  return e;
}
__device__ __forceinline__ float tile_drho_owner(const Dev &d, int i, const TileDrhoPre &p, float dt, const float4 *lds) {
  const float3 vi = xyz(p.v);
  float t = 0.f;
  tile_walk<true>(d, i, xyz(p.pi), p.l, lds, [&](float3 r, float r2, float4, float4 vj) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(vj), r); });
  walk_solids(d, i, xyz(p.pi), (int)((unsigned)p.head >> 16), [&](int, float3 r, float r2) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi, r); });
  return tile_drho_finish(d, i, p.rho, p.alpha, dt, t);
}
__device__ __forceinline__ float tile_drho_gather(const Dev &d, int i, float dt) {
  const float4 pi = d.pk[i];
  const float3 vi = xyz(d.vel[i]);
  float t = 0.f;
  walk<4>(d, i, pi, d.pk, [&](int j, float4) { return nb_ld(d.vel, j); },   // (four entries per trip: the register budget is the tile walk's)
          [&](int, float3 r, float r2, float4 vj) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(vj), r); },
          [&](int, float3 r, float r2) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi, r); });
  return tile_drho_finish(d, i, d.rho[i], d.alpha[i], dt, t);
}
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(TILE_WG_D * NT / WAVE / 4, TILE_WG_D * NT / WAVE / 4))) k_tile_drho(Dev d, float dt) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  float esum = 0.f;
#define TD_LOAD(i, pre) tile_drho_load(d, i, pre)
#define TD_OWNER(i, pre) esum += tile_drho_owner(d, i, pre, dt, lds)
#define TD_GATHER(i) esum += tile_drho_gather(d, i, dt)
  TILE_PASS_BODY(NT, true, d.pk, d.vel, TileDrhoPre, TD_LOAD, TD_OWNER, TD_GATHER)
  TILE_OUTSIDE_LOOP(d, i) esum += tile_drho_gather(d, i, dt);
#undef TD_LOAD
#undef TD_OWNER
#undef TD_GATHER
  tile_accumulate<NT>(ACC_SLOT(d, ACC_ERR), esum);   // ref: dfsph.py:546
}

// ---- development check: an owner's slot list, decoded through its group's slot table, is its gather list entry for entry ------------
// tl_n[3] += owners whose lists differ, tl_n[4] += list entries compared (tools/tile_bench.py)
__global__ void __launch_bounds__(BLOCK) k_tile_check(Dev d) {
  __shared__ int desc[TL_DESC];
  int bad = 0, seen = 0;
  TILE_GROUP_LOOP(d, a) {
    __syncthreads();
    if (threadIdx.x < TL_DESC) desc[threadIdx.x] = d.tl_grp[(size_t)a * TL_DESC + threadIdx.x];
    __syncthreads();
    if (desc[1] < 0) continue;
    const int *tbl = d.tl_tbl + desc[2];
    for (int o = threadIdx.x; o < desc[3]; o += BLOCK) {
      const int i = tile_owner(desc, o);
      const int cl = d.nhead[i] & 0xffff, ct = d.ntl[i];
      const unsigned short *L = reinterpret_cast<const unsigned short *>(d.nbl), *T = reinterpret_cast<const unsigned short *>(d.nbt);
      int run = i, t = 0, ok = 1;
      for (int k = 0; k < cl; k++) {
        const int dd = (int)(short)L[((size_t)(k >> 3) * d.nbr_stride + i) * 8 + (k & 7)];
        run += dd;
        if (!NB_IS_ENTRY(dd)) continue;
        if (t >= ct) { ok = 0; break; }
        const unsigned off = T[((size_t)(t >> 3) * d.nbr_stride + i) * 8 + (t & 7)];
        if ((off & 15u) || (int)(off >> 4) >= desc[1] || tbl[off >> 4] != run) ok = 0;
        t++;
      }
      if (t != ct) ok = 0;
      bad += !ok; seen += t;
    }
  }
  if (bad) atomicAdd(&d.tl_n[3], bad);
  atomicAdd(&d.tl_n[4], seen);
}
