// sampler.h -- parallel Poisson-disk sampling of a triangle mesh into boundary particles
// (ref: boundry.py:100-460, SURVEY 8(f) rank 2: the producer of the hot path's solid particles).
//
// Same algorithm as the reference tool: 40 x (area / pi R^2) uniformly random surface points (area-weighted triangle
// pick by rejection, sqrt barycentrics, boundry.py:222-246), cells of edge R / sqrt(3), candidates ordered by cell,
// `trials` rounds over the 27 phase groups (cell coordinates mod 3, boundry.py:265-271) in which every occupied
// cell of the group tests its trial-th candidate against the samples already accepted in the 5x5x5 cells around it
// with the normal-corrected distance of boundry.py:340-370 and accepts it when none is closer than R.
// Differences, on purpose: a dense exact cell table instead of the reference's aliasing hash map of cells, a counting
// sort instead of the bitonic sort, a counter-based RNG (reproducible; ti.random is not), candidates of one cell
// ordered by their index (the reference's order inside a cell is whatever the bitonic network leaves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PS_MAX_PER_CELL 5   /* boundry.py:62 hash_sample_size */

struct PsDev {
  int n_tri, n_cand, nx, ny, nz, ncell;
  float minx, miny, minz, inv_grid, radius, max_area;
  uint32_t seed;
  const float *tri;      // [n_tri][9]
  const float *nrm;      // [n_tri][3]
  const float *area;     // [n_tri]
  float *pos;            // [n_cand][3] candidates
  int *tri_id, *cell;    // [n_cand]
  int *cnt, *start;      // [ncell+1]
  int *order;            // [n_cand] candidate ids sorted by (cell, id)
  int *samp_cnt, *samp;  // [ncell], [ncell][PS_MAX_PER_CELL] accepted candidate ids
  int *accepted;         // [n_cand]
  int *occ, *occ_start;  // occupied cells grouped by phase, [28]
};

__host__ __device__ inline uint32_t ps_hash(uint32_t seed, uint32_t i, uint32_t k) {
  uint32_t x = seed + i * 747796405u + k * 2891336453u + 0x9E3779B9u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ inline float ps_u01(uint32_t seed, uint32_t i, uint32_t k) { return (float)(ps_hash(seed, i, k) >> 8) * (1.0f / 16777216.0f); }

// candidate i: position, triangle, cell -- shared by the device kernel and the CPU restatement, written with explicit
// single roundings (no fma contraction) so both produce the same bits
__host__ __device__ inline void ps_candidate(uint32_t seed, int i, int n_tri, const float *tri, const float *area, float max_area,
                                             float minx, float miny, float minz, float inv_grid, float *p, int *tid, int *cxyz) {
#pragma clang fp contract(off)   // single roundings: keep the compiler from fusing mul+add
#ifdef __HIP_DEVICE_COMPILE__
#define PS_MUL(a, b) ((a) * (b))
#define PS_ADD(a, b) ((a) + (b))
#define PS_SQRT(a) ((float)sqrt((double)(a)))   /* v_sqrt_f32 alone is 1 ulp; the f64 routine rounds to the exact float */
#else
#define PS_MUL(a, b) ((float)((volatile float)(a) * (volatile float)(b)))
#define PS_ADD(a, b) ((float)((volatile float)(a) + (volatile float)(b)))
#define PS_SQRT(a) sqrtf(a)
#endif
  const float rn1 = PS_SQRT(ps_u01(seed, i, 0));
  const float bc1 = PS_ADD(1.0f, -rn1);
  const float bc2 = PS_MUL(ps_u01(seed, i, 1), rn1);
  const float bc3 = PS_ADD(PS_ADD(1.0f, -bc1), -bc2);
  int t = 0;
  for (uint32_t k = 2;; k += 2) {   // boundry.py:233-236 rejection on the triangle area
    t = (int)PS_MUL((float)n_tri, ps_u01(seed, i, k));
    if (t >= n_tri) t = n_tri - 1;
    if (ps_u01(seed, i, k + 1) < area[t] / max_area || k > 4000) break;
  }
  const float *a = tri + 9 * t;
  for (int c = 0; c < 3; c++) p[c] = PS_ADD(PS_ADD(PS_MUL(bc1, a[c]), PS_MUL(bc2, a[3 + c])), PS_MUL(bc3, a[6 + c]));
  *tid = t;
  cxyz[0] = (int)PS_MUL(PS_ADD(p[0], -minx), inv_grid) + 1;   // boundry.py:245 (+1 cell of margin)
  cxyz[1] = (int)PS_MUL(PS_ADD(p[1], -miny), inv_grid) + 1;
  cxyz[2] = (int)PS_MUL(PS_ADD(p[2], -minz), inv_grid) + 1;
}

// boundry.py:340-370: distance between two surface points, stretched by the normals when they lie on different faces
__host__ __device__ inline float ps_distance(const float *p, const float *q, int tp, int tq, const float *nrm) {
#pragma clang fp contract(off)
  float v[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
  float dist = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (tp != tq && dist > 0.0f) {
    const float inv = 1.0f / dist;
    const float c1 = (nrm[3 * tp] * v[0] + nrm[3 * tp + 1] * v[1] + nrm[3 * tp + 2] * v[2]) * inv;
    const float c2 = (nrm[3 * tq] * v[0] + nrm[3 * tq + 1] * v[1] + nrm[3 * tq + 2] * v[2]) * inv;
    if (fabsf(c1 - c2) > 0.00001f) dist *= (asinf(fminf(fmaxf(c1, -1.f), 1.f)) - asinf(fminf(fmaxf(c2, -1.f), 1.f))) / (c1 - c2);
    else dist /= sqrtf(fmaxf(1.0f - c1 * c1, 1e-12f));
  }
  return dist;
}

__global__ void k_ps_init(PsDev s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cand) return;
  float p[3]; int t, c[3];
  ps_candidate(s.seed, i, s.n_tri, s.tri, s.area, s.max_area, s.minx, s.miny, s.minz, s.inv_grid, p, &t, c);
  s.pos[3 * i] = p[0]; s.pos[3 * i + 1] = p[1]; s.pos[3 * i + 2] = p[2];
  s.tri_id[i] = t;
  const int lin = (c[0] * s.ny + c[1]) * s.nz + c[2];
  s.cell[i] = lin;
  atomicAdd(&s.cnt[lin], 1);
}
// rank inside the cell = number of same-cell candidates with a smaller id (deterministic order)
__global__ void k_ps_place(PsDev s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cand) return;
  const int lin = s.cell[i];
  s.order[s.start[lin] + atomicAdd(&s.cnt[lin], 1)] = i;   // cnt was re-zeroed: provisional slot
}
__global__ void k_ps_sort_cells(PsDev s) {   // one thread per cell: order its (few) candidates by id
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= s.ncell) return;
  const int b = s.start[c], e = s.start[c + 1];
  for (int x = b + 1; x < e; x++) {
    const int v = s.order[x];
    int y = x - 1;
    while (y >= b && s.order[y] > v) { s.order[y + 1] = s.order[y]; y--; }
    s.order[y + 1] = v;
  }
}
__global__ void k_ps_trial(PsDev s, int pg, int trial) {
  int k = s.occ_start[pg] + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= s.occ_start[pg + 1]) return;
  const int c = s.occ[k];
  const int slot = s.start[c] + trial;
  if (slot >= s.start[c + 1]) return;
  const int cand = s.order[slot];
  const int cz = c % s.nz, cy = (c / s.nz) % s.ny, cx = c / (s.nz * s.ny);
  const float *p = s.pos + 3 * cand;
  const int tp = s.tri_id[cand];
  bool ok = true;
  for (int dx = -2; dx <= 2 && ok; dx++)
    for (int dy = -2; dy <= 2 && ok; dy++)
      for (int dz = -2; dz <= 2 && ok; dz++) {
        const int x = cx + dx, y = cy + dy, z = cz + dz;
        if ((unsigned)x >= (unsigned)s.nx || (unsigned)y >= (unsigned)s.ny || (unsigned)z >= (unsigned)s.nz) continue;
        const int n = (x * s.ny + y) * s.nz + z;
        const int m = s.samp_cnt[n];
        for (int q = 0; q < m; q++) {
          const int o = s.samp[n * PS_MAX_PER_CELL + q];
          if (ps_distance(p, s.pos + 3 * o, tp, s.tri_id[o], s.nrm) < s.radius) { ok = false; break; }
        }
      }
  if (ok) {
    const int m = s.samp_cnt[c];   // only this thread touches cell c in this phase group
    if (m < PS_MAX_PER_CELL) { s.samp[c * PS_MAX_PER_CELL + m] = cand; s.samp_cnt[c] = m + 1; s.accepted[cand] = 1; }
  }
}
