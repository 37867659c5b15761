// device.h -- device-side data view, pair geometry and the neighbour walk shared by every pass.
//
// Layout (DESIGN.md "Data layout in HBM"): liquid particles live in CELL-SORTED order in
// SoA arrays; vec3 fields are float4-padded so that one neighbour gather is a single
// 16-byte load.  pos.w carries the particle's packed cell (cx | cy<<10 | cz<<20, -1 when the
// particle is outside the grid box, ref: HashGrid.py:68-69,117-124).  Static solids are
// sorted once.  Cells are linearised in TILES of 2 x 4 x 16 cells (outer x middle x z, see cell_lin): 64 consecutive
// particles are a compact brick instead of a 64-cell column segment, and the <= 5 cells of one (dx,dy) row of the
// 5x5x5 walk (ref: HashGrid.py:82-85) are at most TWO contiguous particle ranges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define BLOCK 256
#define ALIAS_STRIDE 128 /* uint16 per aliased cell: [0]=n entries, [1]=self multiplicity, [2..] = k | extra<<8 */

struct Dev {
  // counts
  int nl, ns, n_owned;
  int nx, ny, nz, ncell;   // ncell = nx*ny*nz: plain (x,y,z) linearisation, used only by the faithful box filter
  // tile order (cell_lin): perm = 0: x outer, y middle; 1: y outer, x middle (the longer of the two is outer, so that
  // the particles between two neighbouring outer tile planes -- the largest delta of a neighbour stream -- are few)
  int perm, nt0, nt1, nt2, ncell_p;   // tiles per axis, padded cell count (= index of the out-of-box bin)
  int x0;          // first global cell column of this context's window (slab decomposition); 0 otherwise
  int own_lo, own_hi;   // global cell columns this context OWNS (bucket histogram); [0, inf) without a decomposition
  // slab halo/compute overlap: 0 = every owned particle; 1 = only those in global cell columns [part_x0, part_x1)
  // (the interior: no ghost within their 5^3 cells); 2 = only the others (see part_ok)
  int part_mode, part_x0, part_x1;
  // traversal direction of this launch (sorted_index): consecutive neighbour passes alternate, so that a pass starts on
  // the data the previous one touched last -- the part of it that is still in the 256 MB Infinity Cache
  int rev;
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
  int *gid;                               // slab decomposition: GLOBAL particle id of row o (nullptr: the row index is the id)
  float4 *d_vel, *d_omega, *normal, *cg_r, *cg_dir, *cg_Ad, *cg_s;
  // cg_Minv: the inverse of a symmetric 3 x 3 block is symmetric -- six floats per particle (24 B instead of three padded rows, 48 B:
  // the PCG update reads them and the preconditioner pass writes them, once per step each, from / to HBM)
  float4 *minv_a;                         // (m00, m01, m02, m11)
  float2 *minv_b;                         // (m12, m22)
  float *rho, *adv_rho, *pressure, *vel_max, *alpha, *kbuf;
  float4 *pk;                             // per-pass packed gather record (pos.xyz, payload)
  float4 *pr;                             // (pos.xyz, rho): one gather instead of two in the viscosity / vorticity / tension walks
  // pcisph.py / iisph.py private fields
  float4 *d_vel_pre, *pos_star, *vel_star, *d_ii, *dij_pj;
  float *a_ii, *pressure_pre;
  int *ncount;                            // HashGrid.neighborCount (raw list length)
  // compact per-step neighbour lists: 16-bit delta streams, 8 slots per 16-byte block, block b of particle i at
  // [b * nbr_stride + i] (see "neighbour streams" below)
  uint4 *nbl, *nbs;                       // liquid / solid streams
  int *nhead;                             // liquid slots | solid slots << 16
  int *nsbase;                            // sorted solid index the solid stream starts from (read only by particles that have solid entries)
  int nbr_stride, nbr_cap;
  float list_r2;                          // squared support radius the lists are built with
  int *bbox;                              // [6] {xmin,ymin,zmin,xmax,ymax,zmax} of the in-box liquid cells of this step
  // solids, cell-sorted once
  float4 *spos;
  // cell tables
  int *cs_l;          // [ncell_p+2] liquid cell start in tile order (ncell_p = out-of-box bin)
  int *cs_s;          // [ncell_p+2] solid cell start
  int *cell_info;     // [ncell_p] bit0: solids within the 5^3 walk; bits1..: alias slot + 1
  uint16_t *alias;    // [n_alias][ALIAS_STRIDE]
  // speculative launches (wcsph_hip.hip k_publish): a kernel launched with pred != nullptr does nothing when *pred == 0
  const int *pred;
  // ... and takes the time step from *dtp instead of its dt argument when dtp != nullptr: the kernels queued behind the
  // CFL reduction run before the host has seen the new deltaT (k_cfl_store forms it on the device, DtRule)
  const float *dtp;
  // reductions
  double *acc;        // [8 slots][ACC_STRIPES]
  int *flags;         // [4] error bits, vmax bits
};

enum { ACC_ERR = 0, ACC_DAD = 1, ACC_DNEW = 2, ACC_DZERO = 3 };
enum { FLAG_ERR = 0, FLAG_VMAX = 1, FLAG_DT = 2, FLAG_WARM = 4 + 64 };   // [4, 68): striped vmax; FLAG_WARM: generation number of the last step whose warm-start gate opened   // FLAG_DT: bits of the deltaT k_cfl_store formed (speculative steps)
enum { ERRBIT_BUCKET = 1, ERRBIT_NEIGH = 2, ERRBIT_LISTCAP = 4, ERRBIT_NONFINITE = 8, ERRBIT_ROWITEMS = 16 };

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
  // |r| = r2 * rsq(r2): the same rsq cubic_gradW_scale needs (one transcendental for both when a pass calls both);
  // sqrtf() would expand to a ~17-instruction correctly rounded sequence per entry
  float q = r2 * rsqrtf(fmaxf(r2, 1e-30f)) * d.inv_h;
  float f = 1.0f - q;
  float w = (q <= 0.5f) ? (6.0f * q * q * q - 6.0f * q * q + 1.0f) : (2.0f * f * f * f);
  return (q <= 1.0f) ? w * d.mk_h3 : 0.0f;
}
// 1-ulp reciprocal (v_rcp_f32) for the per-entry quotients of the neighbour loops; `/` is the ~10-instruction
// correctly rounded sequence and stays in the per-particle code
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
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

// ---- tile-ordered cell index ----------------------------------------------------------------------------
#define T0B 1   /* outer axis: 2 cells per tile  */
#define T1B 2   /* middle axis: 4 cells per tile */
#define T2B 4   /* z: 16 cells per tile (row_pieces relies on >= 5 cells: a walked row touches at most two tiles) */
#define TILE_SHIFT (T0B + T1B + T2B)
// index of the first cell (z = 0 of z-tile 0) of the z-row through (cx, cy): cell (cx, cy, cz) = row + ((cz >> T2B) << TILE_SHIFT) + (cz & (2^T2B - 1))
__device__ __forceinline__ int cell_row(const Dev &d, int cx, int cy) {
  const int a0 = d.perm ? cy : cx, a1 = d.perm ? cx : cy;
  return ((((a0 >> T0B) * d.nt1 + (a1 >> T1B)) * d.nt2) << TILE_SHIFT) | ((a0 & ((1 << T0B) - 1)) << (T1B + T2B)) | ((a1 & ((1 << T1B) - 1)) << T2B);
}
__device__ __forceinline__ int cell_lin(const Dev &d, int cx, int cy, int cz) {
  return cell_row(d, cx, cy) + ((cz >> T2B) << TILE_SHIFT) + (cz & ((1 << T2B) - 1));
}
// particle ranges of the cells (x, y, z0 .. z1-1), 0 < z1 - z0 <= 2^T2B (= 16; the walk asks for <= 5): such a z-run touches at most two tiles
struct RowPieces { int lo0, hi0, lo1, hi1; };
__device__ __forceinline__ RowPieces row_pieces(const Dev &d, const int *CS, int x, int y, int z0, int z1) {
  const int rb = cell_row(d, x, y);
  const int tA = z0 >> T2B, tB = (z1 - 1) >> T2B;
  const int bA = rb + (tA << TILE_SHIFT);
  RowPieces r;
  r.lo0 = CS[bA + (z0 & ((1 << T2B) - 1))];
  r.hi0 = CS[bA + min(z1 - (tA << T2B), 1 << T2B)];   // [.. + 2^T2B] is the next row of the tile (or the next tile): the end of the tile row's last cell
  // branch free: when the run stays in one tile the second piece reads the same entry twice (an empty range)
  const int bB = rb + (tB << TILE_SHIFT);
  r.lo1 = CS[bB];
  r.hi1 = CS[bB + (tB > tA ? z1 - (tB << T2B) : 0)];
  return r;
}

// ---- "once" accesses: data a launch touches exactly once and nobody gathers afterwards ---------------------------------
// Non-temporal loads and stores do NOT allocate in the 256 MB Infinity Cache (tools/micro/mall_policy.hip,
// profiles/r03n_mall_policy.txt: a 96 MB buffer survives a 900 MB nt sweep and is gone after a 320 MB plain one), so the
// elementwise traffic of a step -- scratch vectors of the PCG, per-particle scalars, the sort's copies -- is kept from
// displacing what the neighbour passes come back to launch after launch: the 16-bit neighbour streams and the gather
// records.  Arrays that a following pass GATHERS (pos, vel, pk, pr, vel_guess, cg_dir, omega) stay on plain accesses.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_once(const float4 *p) {
  const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld_once(const float *p) {
  return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ int ld_once(const int *p) {
  return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ void st_once(float4 *p, float4 v) {
  v4f_nt w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<v4f_nt *>(p));
}
__device__ __forceinline__ float2 ld_once(const float2 *p) {
  typedef float v2f_nt __attribute__((ext_vector_type(2)));
  const v2f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v2f_nt *>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_once(float2 *p, float2 v) {
  typedef float v2f_nt __attribute__((ext_vector_type(2)));
  v2f_nt w = {v.x, v.y};
  __builtin_nontemporal_store(w, reinterpret_cast<v2f_nt *>(p));
}
__device__ __forceinline__ void st_once(float *p, float v) {
  __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ void st_once(int *p, int v) {
  __builtin_nontemporal_store(v, p);
}

// first statement of a kernel that may be launched speculatively (grid-uniform: taken by every thread or by none)
#define PRED_GATE(d) do { if ((d).pred && *(d).pred == 0) return; } while (0)
#define DT_OVERRIDE(d, dt) do { if ((d).dtp) dt = *(d).dtp; } while (0)

// does this context own row i?  Without a decomposition every row is owned (n_owned == nl, wave-uniform): no oid read.
__device__ __forceinline__ bool owned(const Dev &d, int i) { return d.n_owned >= d.nl || d.oid[i] < d.n_owned; }

// which owned particles a neighbour pass computes in this launch (Dev::part_mode); particles outside the grid box
// have no neighbours and count as interior
__device__ __forceinline__ bool part_ok(const Dev &d, float4 pi) {
  if (d.part_mode == 0) return true;
  const int w = __float_as_int(pi.w);
  const bool in = (w < 0) || (unsigned)((w & 1023) + d.x0 - d.part_x0) < (unsigned)(d.part_x1 - d.part_x0);
  return (d.part_mode == 1) == in;
}

// XCD-aware block remap: the dispatcher places block b on XCD b%8 (MI355X_MICROARCH "Workgroup
// dispatch"); give every XCD one CONTIGUOUS eighth of the cell-sorted particle range so that its
// private L2 holds one spatial region instead of a 1/8 sample of all of them.  gridDim.x is a
// multiple of 8 by construction (grid_for()).
__device__ __forceinline__ int sorted_index(const Dev &d) {
  const int per = gridDim.x >> 3;
  int q = blockIdx.x >> 3;
  if (d.rev) q = per - 1 - q;   // every XCD walks its eighth backwards
  const int blk = (blockIdx.x & 7) * per + q;
  return blk * BLOCK + threadIdx.x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
  return v;
}
// Reduction slots are STRIPED: slot S owns ACC_STRIPES doubles and block b adds to stripe b % ACC_STRIPES.  Device
// scope atomics are resolved at the memory side, one per address at a time: 7.8 k blocks adding to ONE double cost a
// pass 27 us at C3 (tools/ablate.py, drho 7 vs 9); over 64 addresses it is 2 us.  Readers sum the stripes
// (acc_total on the device, pull_scalars on the host).
#define ACC_STRIPES 64
#define ACC_SLOT(d, S) ((d).acc + (S) * ACC_STRIPES)
// a kernel that does not otherwise touch slot S clears it for the kernel launched next (saves a memset launch)
__device__ __forceinline__ void acc_clear(double *slot) {
  if (blockIdx.x == 0 && threadIdx.x < ACC_STRIPES) slot[threadIdx.x] = 0.0;
}
// every thread of the block must call this
__device__ __forceinline__ void block_accumulate(double *slot, float v) {
  double *dst = slot + (blockIdx.x & (ACC_STRIPES - 1));
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

// total of a striped slot; every thread of the block must call this (blocks of at least one wave)
__device__ __forceinline__ double acc_total(const double *slot) {
  __shared__ double tot;
  __syncthreads();   // a kernel may call this twice: every wave must have read the previous total before it is overwritten
  if (threadIdx.x < WAVE) {
    double v = slot[threadIdx.x & (ACC_STRIPES - 1)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    if (threadIdx.x == 0) tot = v;
  }
  __syncthreads();
  return tot;
}

// gather of element j of a kernel-argument array with a 32-bit byte offset: lets the compiler use the
// `global_load vdst, voffset, saddr` form (one VGPR per address, one shift) instead of 64-bit pointer arithmetic
template <class T>
__device__ __forceinline__ T nb_ld(const T *base, int j) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (unsigned)j * (unsigned)sizeof(T));
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
  const float3 xi = xyz(pi);
  const int info = d.cell_info[cell_lin(d, cx, cy, cz)];
  // Row trimming: a row (dx, dy) of the 5x5 columns is walked only over the z-cells the support sphere can reach.
  // Distances in cell units from the particle to the nearest face of the column; rm = list radius + a margin
  // for the rounding of the cell assignment.  Purely a candidate filter: the r2 <= list_r2 test below decides.
  const float ux = (pi.x - d.minx) * d.inv_grid - (float)(cx + d.x0), uy = (pi.y - d.miny) * d.inv_grid - (float)cy;
  const float fz = (pi.z - d.minz) * d.inv_grid;
  const float rm = __builtin_amdgcn_sqrtf(d.list_r2) * d.inv_grid * 1.0001f + 1e-3f, rm2 = rm * rm;
  // (squared distance to the column's face along one axis, offset e cells: the five middle-axis values are formed once per
  // particle, the outer-axis one once per plane -- a row costs two subtractions instead of two compare / select chains)
  auto face2 = [](int e, float u) -> float {
    const float v = e > 0 ? (float)e - u : (e < 0 ? (float)(-e - 1) + u : 0.0f);
    return v * v;
  };
  const float u0 = d.perm ? uy : ux, u1 = d.perm ? ux : uy;
  const float m2[5] = {face2(-2, u1), face2(-1, u1), 0.0f, face2(1, u1), face2(2, u1)};
  auto zrange = [&](float o2, float mm2, int &za, int &zb) -> bool {
    const float q = rm2 - o2 - mm2;
    if (q < 0.0f) return false;
    const float zr = __builtin_amdgcn_sqrtf(q);
    za = max(max((int)floorf(fz - zr), cz - 2), 0);
    zb = min(min((int)floorf(fz + zr), cz + 2), d.nz - 1) + 1;
    return zb > za;
  };
  // The walk goes plane by plane along the OUTER axis of the tile order (x, or y when perm = 1), then row by row along
  // the middle axis: indices grow almost monotonically, so the deltas of the neighbour streams stay small.  Per outer
  // plane the cell-table lookups of its five rows (two pieces each, row_pieces) are issued together, then scanned.
  const int c0 = d.perm ? cy : cx, c1 = d.perm ? cx : cy, n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
#define WALK_PLANE(CS, POS, SELF, CALL)                                                            \
  _Pragma("unroll 1") for (int e0 = -2; e0 <= 2; e0++) {                                          \
    const int a0 = c0 + e0;                                                                       \
    if ((unsigned)a0 >= (unsigned)n0) continue;                                                   \
    const float o2 = face2(e0, u0);                                                               \
    RowPieces rp[5];                                                                              \
    _Pragma("unroll") for (int t = 0; t < 5; t++) {                                               \
      const int a1 = c1 + t - 2;                                                                  \
      int z0, z1;                                                                                 \
      rp[t].lo0 = rp[t].hi0 = rp[t].lo1 = rp[t].hi1 = 0;                                          \
      if ((unsigned)a1 < (unsigned)n1 && zrange(o2, m2[t], z0, z1))                               \
        rp[t] = row_pieces(d, CS, d.perm ? a1 : a0, d.perm ? a0 : a1, z0, z1);                    \
    }                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 5; t++) {                                               \
      int j = rp[t].lo0, end = rp[t].hi0, j2 = rp[t].lo1;                                         \
      const int end2 = rp[t].hi1;                                                                 \
      if (j >= end) { j = j2; end = end2; j2 = end2; }                                            \
      if (j < end) {                                                                              \
        /* the next candidate's position is requested before this one is tested */               \
        float4 pj = nb_ld(POS, j);                                                                \
        for (;;) {                                                                                \
          int jn = j + 1, endn = end;                                                             \
          if (jn >= end) { jn = j2; endn = end2; j2 = end2; }                                     \
          const bool more = jn < endn;                                                            \
          float4 pn = pj;                                                                         \
          if (more) pn = nb_ld(POS, jn);                                                          \
          const float3 r = xi - xyz(pj);                                                          \
          const float r2 = dot3(r, r);                                                            \
          if (r2 <= d.list_r2 && (SELF)) CALL(j, r, r2, 1.0f);                                    \
          if (!more) break;                                                                       \
          pj = pn; j = jn; end = endn;                                                            \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
  }
  if (liquids) { WALK_PLANE(d.cs_l, d.pos, j != i, fl) }
  if (info & 1) { WALK_PLANE(d.cs_s, d.spos, true, fs) }
#undef WALK_PLANE
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
      const int c = cell_lin(d, x, y, z);
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

// ---- neighbour streams ---------------------------------------------------------------------------------
// The compact list of particle i (the in-support subset of the reference's HashGrid.neighbor row, duplicates
// included) is stored as a stream of int16 DELTAS: slot k holds j_k - j_{k-1}, the liquid stream starting from
// j_{-1} = i (neighbours are near i in the cell-sorted order), the solid stream from nsbase[i].  A sorted index
// needs 4 bytes, its delta 2: the index stream is what a pass reads most of (tools/ablate.py), so this halves the
// HBM traffic of every neighbour pass.  Deltas that do not fit (the jump to the neighbouring outer tile plane of a
// large grid) are bridged by HOP slots, +-32767, which move the running index but are not entries.  Eight slots form one
// 16-byte block; block b of particle i sits at [b * nbr_stride + i], so a wave loads 1 KB contiguous per trip.
#define NB_HOP 32767
#define NB_IS_ENTRY(dd) ((unsigned)((dd) + (NB_HOP - 1)) <= (unsigned)(2 * NB_HOP - 2))   // |dd| < NB_HOP

// builder side: shift register of eight slots, flushed as whole blocks
struct NbWriter {
  uint4 *base;        // &stream[i]
  size_t stride;
  int run, slots, cap;
  unsigned w0, w1, w2, w3;
  __device__ __forceinline__ void init(uint4 *b, size_t st, int start, int cap_) { base = b; stride = st; run = start; slots = 0; cap = cap_; w0 = w1 = w2 = w3 = 0; }
  __device__ __forceinline__ void slot(int dd) {
    w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 = __builtin_amdgcn_alignbit(w2, w1, 16);
    w2 = __builtin_amdgcn_alignbit(w3, w2, 16); w3 = (w3 >> 16) | ((unsigned)dd << 16);
    slots++;
    if ((slots & 7) == 0 && slots <= cap) base[(size_t)((slots >> 3) - 1) * stride] = make_uint4(w0, w1, w2, w3);
  }
  __device__ __forceinline__ void entry(int j) {
    int dd = j - run;
    if (!NB_IS_ENTRY(dd)) {   // one test for both directions: the hop loops are rare (the jump to another outer tile plane of a large grid)
      while (dd >= NB_HOP) { slot(NB_HOP); dd -= NB_HOP; }
      while (dd <= -NB_HOP) { slot(-NB_HOP); dd += NB_HOP; }
    }
    slot(dd);
    run = j;
  }
  // pads the last block with zero deltas (valid indices, masked by the slot count); returns the slot count
  __device__ __forceinline__ int finish() {
    const int n = slots;
    if (n <= cap) while (slots & 7) slot(0);
    return n;
  }
};

// The neighbour loop of every pass: iterate the compact list the grid update built for this step.
// As in the reference the list is frozen until the next update_grid() while r is recomputed from the
// current positions (ref: dfsph.py:254-256 and every other `while k < cur_neighbor` loop).
//
// Latency structure: one trip decodes U slots from one coalesced 16-byte (U = 8) or 8-byte (U = 4) load, then
// issues all U position gathers AND all U payload gathers gl(j) (whatever the pass reads from neighbour j: vel_j,
// k_j, rho_j ...) before any arithmetic, so a trip costs two dependent memory round trips instead of 2 per entry.
// fl(j, r, r2, payload) / fs(j, r, r2) are called for the liquid / solid entries.
// P: the float4 array positions are gathered from (d.pos, or a per-pass packed copy whose .w already holds
// the neighbour payload so that one 16-byte gather per entry suffices); gl(j, p4) builds the payload.
struct NoPayload {};
// The first slot of a trip can be a hop, so its body is conditional -- and LLVM then sinks that slot's gathers into
// the conditional block, behind the branch, which serialises a third memory round trip per trip.  Pinning the
// loaded values right after the trip's last gather is issued keeps all gathers in front of the first branch.
template <class T>
__device__ __forceinline__ void nb_pin(T &v) {
  if constexpr (sizeof(T) >= 4) {   // one dword of every 16-byte member is enough to hold its load in place
    float *w = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(T) / 4); q += 4) asm volatile("" : "+v"(w[q]));
  }
}
#define NB_LO(w) ((int)(short)((w) & 0xffffu))
#define NB_HI(w) ((int)(w) >> 16)
// the solid stream of particle i (cs slots; see walk()): fs(j, r, r2) for every entry
template <class FS>
__device__ __forceinline__ void walk_solids(const Dev &d, int i, float3 xi, int cs, FS &&fs) {
  if (cs == 0) return;
  const size_t st = (size_t)d.nbr_stride;
  int run;
  const uint4 *S = d.nbs + i;
  run = d.nsbase[i];
#define WALK_SOL(u, dd, jj, pp) if (k + u < cs && NB_IS_ENTRY(dd)) { const float3 r = xi - xyz(pp); fs(jj, r, dot3(r, r)); }
  for (int k = 0; k < cs; k += 8) {
    const uint4 e = *(S + (size_t)(k >> 3) * st);
    const int d0 = NB_LO(e.x), d1 = NB_HI(e.x), d2 = NB_LO(e.y), d3 = NB_HI(e.y), d4 = NB_LO(e.z), d5 = NB_HI(e.z), d6 = NB_LO(e.w), d7 = NB_HI(e.w);
    const int j0 = run + d0, j1 = j0 + d1, j2 = j1 + d2, j3 = j2 + d3, j4 = j3 + d4, j5 = j4 + d5, j6 = j5 + d6, j7 = j6 + d7;
    run = j7;
    float4 p0 = nb_ld(d.spos, j0);
    const float4 p1 = nb_ld(d.spos, j1), p2 = nb_ld(d.spos, j2), p3 = nb_ld(d.spos, j3), p4 = nb_ld(d.spos, j4), p5 = nb_ld(d.spos, j5), p6 = nb_ld(d.spos, j6), p7 = nb_ld(d.spos, j7);
    nb_pin(p0);
    WALK_SOL(0, d0, j0, p0) WALK_SOL(1, d1, j1, p1) WALK_SOL(2, d2, j2, p2) WALK_SOL(3, d3, j3, p3)
    WALK_SOL(4, d4, j4, p4) WALK_SOL(5, d5, j5, p5) WALK_SOL(6, d6, j6, p6) WALK_SOL(7, d7, j7, p7)
  }
#undef WALK_SOL
}

// The stream is loaded plainly by every pass (rounds 1-2 loaded it non-temporally in the once-per-step passes, to keep it from
// displacing the gather records; since round 3 the elementwise traffic is what stays out of the Infinity Cache -- ld_once /
// st_once -- and every walk after the list build finds the stream there).
template <int U, class GL, class FL, class FS>
__device__ __forceinline__ void walk(const Dev &d, int i, float4 pi, const float4 *P, GL &&gl, FL &&fl, FS &&fs) {
  static_assert(U == 4 || U == 8, "U slots per trip");
  const float3 xi = xyz(pi);
  const size_t st = (size_t)d.nbr_stride;
  const int head = d.nhead[i];
  const int cl = head & 0xffff, cs = (int)((unsigned)head >> 16);
  const uint4 *L = d.nbl + i;
  int run = i;
  // scalars, not indexed arrays: keeps the indices in VGPRs (an int j[U] lands in scratch here)
#define WALK_ONE(u, dd, jj, pp, bb) if (dd) { const float3 r = xi - xyz(pp); fl(jj, r, dot3(r, r), bb); }
#define WALK_OK(u, dd) const bool ok##u = (k + u < cl) & NB_IS_ENTRY(dd);   /* a lane mask in SGPRs: the deltas die here */
  // the stream block of the NEXT trip is requested before this trip's gathers go out: its (HBM) latency hides behind
  // the gathers and the arithmetic instead of heading the next trip's dependency chain
  uint4 en = cl > 0 ? *(L) : make_uint4(0, 0, 0, 0);
  for (int k = 0; k < cl; k += U) {
    if (U == 8 && __builtin_amdgcn_ballot_w64(k + 4 < cl) == 0) {
      // last trip of the wave and no lane's list reaches the block's second half: four slots instead of eight (a gather
      // instruction costs its 16+ cycles whether or not a lane takes part, profiles/r03l_gather_rate.txt)
      const uint4 e = en;
      const int d0 = NB_LO(e.x), d1 = NB_HI(e.x), d2 = NB_LO(e.y), d3 = NB_HI(e.y);
      const int j0 = run + d0, j1 = j0 + d1, j2 = j1 + d2, j3 = j2 + d3;
      WALK_OK(0, d0) WALK_OK(1, d1) WALK_OK(2, d2) WALK_OK(3, d3)
      float4 p0 = nb_ld(P, j0);
      const float4 p1 = nb_ld(P, j1), p2 = nb_ld(P, j2), p3 = nb_ld(P, j3);
      auto b0 = gl(j0, p0); const auto b1 = gl(j1, p1); const auto b2 = gl(j2, p2); const auto b3 = gl(j3, p3);
      nb_pin(p0); nb_pin(b0);
      WALK_ONE(0, ok0, j0, p0, b0) WALK_ONE(1, ok1, j1, p1, b1) WALK_ONE(2, ok2, j2, p2, b2) WALK_ONE(3, ok3, j3, p3, b3)
    } else if (U == 8) {
      const uint4 e = en;
      if (k + 8 < cl) en = *(L + (size_t)((k >> 3) + 1) * st);
      const int d0 = NB_LO(e.x), d1 = NB_HI(e.x), d2 = NB_LO(e.y), d3 = NB_HI(e.y), d4 = NB_LO(e.z), d5 = NB_HI(e.z), d6 = NB_LO(e.w), d7 = NB_HI(e.w);
      const int j0 = run + d0, j1 = j0 + d1, j2 = j1 + d2, j3 = j2 + d3, j4 = j3 + d4, j5 = j4 + d5, j6 = j5 + d6, j7 = j6 + d7;
      run = j7;
      WALK_OK(0, d0) WALK_OK(1, d1) WALK_OK(2, d2) WALK_OK(3, d3) WALK_OK(4, d4) WALK_OK(5, d5) WALK_OK(6, d6) WALK_OK(7, d7)
      float4 p0 = nb_ld(P, j0);
      const float4 p1 = nb_ld(P, j1), p2 = nb_ld(P, j2), p3 = nb_ld(P, j3);
      auto b0 = gl(j0, p0); const auto b1 = gl(j1, p1); const auto b2 = gl(j2, p2); const auto b3 = gl(j3, p3);
      const float4 p4 = nb_ld(P, j4), p5 = nb_ld(P, j5), p6 = nb_ld(P, j6), p7 = nb_ld(P, j7);
      const auto b4 = gl(j4, p4); const auto b5 = gl(j5, p5); const auto b6 = gl(j6, p6); const auto b7 = gl(j7, p7);
      nb_pin(p0); nb_pin(b0);
      WALK_ONE(0, ok0, j0, p0, b0) WALK_ONE(1, ok1, j1, p1, b1) WALK_ONE(2, ok2, j2, p2, b2) WALK_ONE(3, ok3, j3, p3, b3)
      WALK_ONE(4, ok4, j4, p4, b4) WALK_ONE(5, ok5, j5, p5, b5) WALK_ONE(6, ok6, j6, p6, b6) WALK_ONE(7, ok7, j7, p7, b7)
    } else {
      const uint2 e = (k & 4) ? make_uint2(en.z, en.w) : make_uint2(en.x, en.y);   // k is wave-uniform
      if ((k & 4) && k + 4 < cl) en = *(L + (size_t)((k >> 3) + 1) * st);
      const int d0 = NB_LO(e.x), d1 = NB_HI(e.x), d2 = NB_LO(e.y), d3 = NB_HI(e.y);
      const int j0 = run + d0, j1 = j0 + d1, j2 = j1 + d2, j3 = j2 + d3;
      run = j3;
      WALK_OK(0, d0) WALK_OK(1, d1) WALK_OK(2, d2) WALK_OK(3, d3)
      float4 p0 = nb_ld(P, j0);
      const float4 p1 = nb_ld(P, j1), p2 = nb_ld(P, j2), p3 = nb_ld(P, j3);
      auto b0 = gl(j0, p0); const auto b1 = gl(j1, p1); const auto b2 = gl(j2, p2); const auto b3 = gl(j3, p3);
      nb_pin(p0); nb_pin(b0);
      WALK_ONE(0, ok0, j0, p0, b0) WALK_ONE(1, ok1, j1, p1, b1) WALK_ONE(2, ok2, j2, p2, b2) WALK_ONE(3, ok3, j3, p3, b3)
    }
  }
#undef WALK_ONE
#undef WALK_OK
  walk_solids(d, i, xi, cs, fs);
}
