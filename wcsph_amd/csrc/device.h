// device.h -- device-side data view, pair geometry and the neighbour walk shared by every pass.
//
// Layout (DESIGN.md "Data layout in HBM"): liquid particles live in CELL-SORTED order in
// SoA arrays; vec3 fields are float4-padded so that one neighbour gather is a single
// 16-byte load.  pos.w carries the particle's packed cell (cx | cy<<10 | cz<<20, -1 when the
// particle is outside the grid box, ref: HashGrid.py:68-69,117-124).  Static solids are
// sorted once.  Cells are linearised z-fastest so that the 5 cells of one (dx,dy) row of
// the 5x5x5 walk (ref: HashGrid.py:82-85) are ONE contiguous particle range.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define BLOCK 256
#define ALIAS_STRIDE 128 /* uint16 per aliased cell: [0]=n entries, [1]=self multiplicity, [2..] = k | extra<<8 */

struct Dev {
  // counts
  int nl, ns, n_owned;
  int nx, ny, nz, ncell;
  int x0;          // first global cell column of this context's window (slab decomposition); 0 otherwise
  int table_size;  // reference hash table size (HashGrid.py:114)
  int faithful;
  int cfl_prefix;  // Q5: CFL max covers original indices [0, cfl_prefix)
  int max_neighbour, max_in_grid;
  // grid box
  float minx, miny, minz, inv_grid;
  // CubicKernel (ref: kernels/CubicKernel.py:12-17) for the active solver
  float h, h2, inv_h, ml_h3, mk_h3;
  // material constants (ref: ParticleData.py:18-24,61-87)
  float VL0, VS0, rho_L0, rho_S0, mass, eps;
  float gx, gy, gz;
  float dim_coff, visc, visc_b;
  float tension, tension_b, adh_mk;
  float visc_omega, vort_coff, vort_init;
  // sesph private constants (ref: sesph.py:34-62)
  float ss_VL0, ss_VS0, ss_mass, ss_rho0, ss_stiff, ss_dim, ss_visc, ss_visc_b;
  // liquid, cell-sorted
  float4 *pos, *vel, *vel_guess, *omega;  // persistent (re-sorted every grid update)
  float *kappa, *kappa_v;                 // persistent
  int *oid;                               // sorted -> original index
  int *where;                             // original -> sorted index
  float4 *d_vel, *d_omega, *normal, *cg_r, *cg_dir, *cg_Ad, *cg_s;
  float4 *minv0, *minv1, *minv2;          // rows of cg_Minv
  float *rho, *adv_rho, *pressure, *vel_max, *alpha, *kbuf;
  float4 *pk;                             // per-pass packed gather record (pos.xyz, payload)
  // pcisph.py / iisph.py private fields
  float4 *d_vel_pre, *pos_star, *vel_star, *d_ii, *dij_pj;
  float *a_ii, *pressure_pre;
  int *ncount;                            // HashGrid.neighborCount (raw list length)
  // compact per-step neighbour lists (k-major: entry k of particle i at [k * nbr_stride + i])
  int *nbr_l, *nbr_s;                     // sorted liquid index / sorted solid index
  int *ncl, *ncs;                         // entries per particle
  int nbr_stride, nbr_cap;
  float list_r2;                          // squared support radius the lists are built with
  // LDS tiles (tile.h): active-tile list of the step, tile-local 16-bit neighbour lists
  int ntx, nty, ntz, ntile;
  int *tile_list, *tile_qoff, *tile_meta; // [ntile] tile ids / first list slot; meta[0] = active tiles
  int tile_ablate;                        // experiments: 1 = stage only, 2 = compute only (WCSPH_TILE_ABLATE)
  int *bbox;                              // [6] {xmin,ymin,zmin,xmax,ymax,zmax} of the in-box liquid cells of this step
  uint16_t *nbr16, *cnt16;                // [nbr_cap][nbr_stride] staged indices, [nl] entries
  // solids, cell-sorted once
  float4 *spos;
  // cell tables
  int *cs_l;          // [ncell+2] liquid cell start (ncell = out-of-box bin)
  int *cs_s;          // [ncell+1] solid cell start
  int *cell_info;     // [ncell] bit0: solids within the 5^3 walk; bits1..: alias slot + 1
  uint16_t *alias;    // [n_alias][ALIAS_STRIDE]
  // reductions
  double *acc;        // [8]
  int *flags;         // [4] error bits, vmax bits
};

enum { ACC_ERR = 0, ACC_DAD = 1, ACC_DNEW = 2, ACC_DZERO = 3 };
enum { FLAG_ERR = 0, FLAG_VMAX = 1 };
enum { ERRBIT_BUCKET = 1, ERRBIT_NEIGH = 2, ERRBIT_LISTCAP = 4 };

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
  return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float3 xyz(float4 v) { return f3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 f4(float3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// ref: kernels/CubicKernel.py:43-54 (Cubic_W_P) * m_k*h3 -- caller guarantees r2 <= h2
__device__ __forceinline__ float cubic_W(const Dev &d, float r2) {
  float q = sqrtf(r2) * d.inv_h;
  float f = 1.0f - q;
  float w = (q <= 0.5f) ? (6.0f * q * q * q - 6.0f * q * q + 1.0f) : (2.0f * f * f * f);
  return (q <= 1.0f) ? w * d.mk_h3 : 0.0f;
}
// ref: kernels/CubicKernel.py:20-32 (CubicGradW) -- returns g with gradW = g * r
__device__ __forceinline__ float cubic_gradW_scale(const Dev &d, float r2) {
  float inv_rl = rsqrtf(fmaxf(r2, 1e-30f));
  float rl = r2 * inv_rl;
  float q = rl * d.inv_h;
  float f = 1.0f - q;
  float s = (q <= 0.5f) ? q * (3.0f * q - 2.0f) : -(f * f);
  float g = s * d.ml_h3 * inv_rl * d.inv_h;
  return (rl > 1.0e-5f && q <= 1.0f) ? g : 0.0f;
}

// XCD-aware block remap: the dispatcher places block b on XCD b%8 (MI355X_MICROARCH "Workgroup
// dispatch"); give every XCD one CONTIGUOUS eighth of the cell-sorted particle range so that its
// private L2 holds one spatial region instead of a 1/8 sample of all of them.  gridDim.x is a
// multiple of 8 by construction (grid_for()).
__device__ __forceinline__ int sorted_index() {
  int per = gridDim.x >> 3;
  int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return blk * BLOCK + threadIdx.x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
  return v;
}
// every thread of the block must call this
__device__ __forceinline__ void block_accumulate(double *dst, float v) {
  __shared__ float sh[BLOCK / WAVE];
  v = wave_sum(v);
  int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BLOCK / WAVE; k++) s += sh[k];
    atomicAdd(dst, (double)s);
  }
}

// The cell walk (grid update only).  fl(j, r, r2, m) is called for every liquid list entry j != i
// inside the support (r = x_i - x_j, m = multiplicity of the entry in the reference's bucket multiset),
// fs(j, r, r2, m) likewise for solid entries.  Entries outside the support contribute exactly zero
// to every kernel-weighted sum of the reference (compact support) and are skipped here; the raw
// list LENGTH the reference also uses (Q2/Q10) is d.ncount[i], built by the grid update.
template <class FL, class FS>
__device__ __forceinline__ void walk_cells(const Dev &d, int i, float4 pi, bool liquids, FL &&fl, FS &&fs) {
  const int cw = __float_as_int(pi.w);
  if (cw < 0) return;  // outside the grid box: no neighbours (ref: HashGrid.py:80-81, Q4)
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const int z0 = max(cz - 2, 0), z1 = min(cz + 2, d.nz - 1) + 1;
  const float3 xi = xyz(pi);
  const int info = d.cell_info[(cx * d.ny + cy) * d.nz + cz];
  if (liquids)
#pragma unroll 1
  for (int dx = -2; dx <= 2; dx++) {
    const int x = cx + dx;
    if ((unsigned)x >= (unsigned)d.nx) continue;
#pragma unroll 1
    for (int dy = -2; dy <= 2; dy++) {
      const int y = cy + dy;
      if ((unsigned)y >= (unsigned)d.ny) continue;
      const int base = (x * d.ny + y) * d.nz;
      const int lo = d.cs_l[base + z0], hi = d.cs_l[base + z1];
      for (int j = lo; j < hi; j++) {
        const float4 pj = d.pos[j];
        const float3 r = xi - xyz(pj);
        const float r2 = dot3(r, r);
        if (r2 <= d.list_r2 && j != i) fl(j, r, r2, 1.0f);
      }
    }
  }
  if (info & 1) {
#pragma unroll 1
    for (int dx = -2; dx <= 2; dx++) {
      const int x = cx + dx;
      if ((unsigned)x >= (unsigned)d.nx) continue;
#pragma unroll 1
      for (int dy = -2; dy <= 2; dy++) {
        const int y = cy + dy;
        if ((unsigned)y >= (unsigned)d.ny) continue;
        const int base = (x * d.ny + y) * d.nz;
        const int lo = d.cs_s[base + z0], hi = d.cs_s[base + z1];
        for (int j = lo; j < hi; j++) {
          const float4 pj = d.spos[j];
          const float3 r = xi - xyz(pj);
          const float r2 = dot3(r, r);
          if (r2 <= d.list_r2) fs(j, r, r2, 1.0f);
        }
      }
    }
  }
  if (info >> 1) {
    // faithful mode, rare: some of this cell's 125 walked cells share a hash bucket, so the
    // reference visits their particles again (Q1).  Entries: k = walked-cell index, extra visits.
    const uint16_t *e = d.alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE;
    const int n = e[0];
#pragma unroll 1
    for (int t = 0; t < n; t++) {
      const int k = e[2 + t] & 0xff;
      const float m = (float)(e[2 + t] >> 8);
      const int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
      const int c = (x * d.ny + y) * d.nz + z;
      if (liquids)
      for (int j = d.cs_l[c]; j < d.cs_l[c + 1]; j++) {
        const float3 r = xi - xyz(d.pos[j]);
        const float r2 = dot3(r, r);
        if (r2 <= d.list_r2 && j != i) fl(j, r, r2, m);
      }
      for (int j = d.cs_s[c]; j < d.cs_s[c + 1]; j++) {
        const float3 r = xi - xyz(d.spos[j]);
        const float r2 = dot3(r, r);
        if (r2 <= d.list_r2) fs(j, r, r2, m);
      }
    }
  }
}

// The neighbour loop of every pass: iterate the compact list the grid update built for this step
// (the in-support subset of the reference's HashGrid.neighbor row, duplicates included).
// As in the reference the list is frozen until the next update_grid() while r is recomputed from the
// current positions (ref: dfsph.py:254-256 and every other `while k < cur_neighbor` loop).
//
// Latency structure: entry k of particle i sits in a 16-byte block of four ([k/4][i][k%4]); one trip
// fetches U entries with U/4 coalesced 16-byte loads, then issues all U position gathers AND all U payload
// gathers gl(j) (whatever the pass reads from neighbour j: vel_j, k_j, rho_j ...) before any arithmetic, so
// a trip costs two dependent memory round trips instead of 2 per entry.
// fl(j, r, r2, payload) / fs(j, r, r2) are called for the liquid / solid entries.
#define NBR_AT(k, stride, i) ((((size_t)((k) >> 2)) * (stride) + (i)) * 4 + ((k) & 3))
// P: the float4 array positions are gathered from (d.pos, or a per-pass packed copy whose .w already holds
// the neighbour payload so that one 16-byte gather per entry suffices); gl(j, p4) builds the payload.
template <int U, class GL, class FL, class FS>
__device__ __forceinline__ void walk(const Dev &d, int i, float4 pi, const float4 *P, GL &&gl, FL &&fl, FS &&fs) {
  static_assert(U == 4 || U == 8, "U entries per trip");
  const float3 xi = xyz(pi);
  const size_t st = (size_t)d.nbr_stride;
  const int cl = d.ncl[i];
  const int4 *L = reinterpret_cast<const int4 *>(d.nbr_l) + i;
  for (int k = 0; k < cl; k += U) {
    const int4 e0 = L[(size_t)(k >> 2) * st];
    int4 e1 = e0;
    if (U == 8 && k + 4 < cl) e1 = L[(size_t)((k >> 2) + 1) * st];  // tail: e1 = e0 re-uses valid indices
    // scalars, not an indexed array: keeps the indices in VGPRs (an int j[U] lands in scratch here)
    const int j0 = e0.x, j1 = e0.y, j2 = e0.z, j3 = e0.w, j4 = e1.x, j5 = e1.y, j6 = e1.z, j7 = e1.w;
    const float4 p0 = P[j0], p1 = P[j1], p2 = P[j2], p3 = P[j3];
    const auto b0 = gl(j0, p0); const auto b1 = gl(j1, p1); const auto b2 = gl(j2, p2); const auto b3 = gl(j3, p3);
    if (U == 8) {
      const float4 p4 = P[j4], p5 = P[j5], p6 = P[j6], p7 = P[j7];
      const auto b4 = gl(j4, p4); const auto b5 = gl(j5, p5); const auto b6 = gl(j6, p6); const auto b7 = gl(j7, p7);
#define WALK_ONE(u, jj, pp, bb) if (k + u < cl) { const float3 r = xi - xyz(pp); fl(jj, r, dot3(r, r), bb); }
      WALK_ONE(0, j0, p0, b0) WALK_ONE(1, j1, p1, b1) WALK_ONE(2, j2, p2, b2) WALK_ONE(3, j3, p3, b3)
      WALK_ONE(4, j4, p4, b4) WALK_ONE(5, j5, p5, b5) WALK_ONE(6, j6, p6, b6) WALK_ONE(7, j7, p7, b7)
    } else {
      WALK_ONE(0, j0, p0, b0) WALK_ONE(1, j1, p1, b1) WALK_ONE(2, j2, p2, b2) WALK_ONE(3, j3, p3, b3)
    }
  }
  const int cs = d.ncs[i];
  const int4 *S = reinterpret_cast<const int4 *>(d.nbr_s) + i;
  for (int k = 0; k < cs; k += 4) {
    const int4 e = S[(size_t)(k >> 2) * st];
    const int j[4] = {e.x, e.y, e.z, e.w};
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) p[u] = d.spos[j[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k + u < cs) {
        const float3 r = xi - xyz(p[u]);
        fs(j[u], r, dot3(r, r));
      }
  }
}
struct NoPayload {};
