/*
 * oracle/wcsph_ref.c -- CPU restatement of the lyd405121/wcsph per-particle hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wcsph_amd/ may import, link or call this
 * file; it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the CHECKER (and as the reported CPU baseline), never as the product path.
 *
 * PARITY UNPINNED by the real reference: it is Python+Taichi, ships no tests / golden vectors and cannot be imported in
 * the build container (no taichi wheel).  This file follows the reference sources line by line (citations
 * "ref: file:line" are into /root/reference) and is pinned by
 *   - the analytic known-answers in tests/test_oracle_known_answers.py (kernel normalisation, lattice density 799.978,
 *     alpha, gradW samples, int32-wrap hash values, default-scene blockSize);
 *   - round 3: golden vectors produced by the reference's UNMODIFIED text (dfsph.py, sesph.py, iisph.py imported from
 *     /root/reference on a serial Taichi stand-in, tests/golden/make_reference_fixtures.py -> tests/golden/ref_*.npz;
 *     tests/test_reference_fixtures.py).  That pins formulas, constants, list order and host control flow; it does not
 *     pin Taichi's execution semantics, so the label above stays.
 *
 * Semantics encoded: f32 / i32 arithmetic, float->int truncation
 * toward zero, int32 wrap-around multiply in the hash, every top-level Taichi `for` is
 * one parallel loop (here: an OpenMP loop or a serial loop), scalar `+=` inside a
 * parallel loop is a reduction (here: accumulated in double, stored as f32).
 *
 * neighbour modes:
 *   mode 0  "faithful": hash table with `count` buckets, ref hash, bucket multiset
 *           (aliasing cells => duplicated / far list entries), HashGrid.py:56-114.
 *   mode 1  "clean":    the same walk with a collision-free hash (bucket == cell), i.e.
 *           every particle of the 5x5x5 in-box cells exactly once.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float nsq(v3 a) { return dot(a, a); }
static inline v3 cross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

typedef struct orc_ctx {
  /* ---- ParticleData (ref: ParticleData.py:12-96) ---- */
  int count, liquid_count, solid_count;
  float rho_L0, rho_S0, VL0, VS0, liqiudMass;
  v3 gravity;
  float dim_coff, viscosity, viscosity_b, viscosity_err;
  float tension_coff, tension_coff_b;
  float viscosity_omega, vorticity_coff, vorticity_init;
  float eps;
  int tension_fixed; /* 0: Q8/Q9-faithful normals (garbage), 1: fixed normals n = h*sum */
  v3 *pos;                                              /* [count] */
  v3 *vel, *vel_guess, *d_vel, *omega, *d_omega, *normal; /* [liquid] */
  float *rho, *adv_rho, *pressure, *vel_max;            /* [liquid] */
  float *cg_Minv;                                       /* [liquid*9] row-major */
  v3 *cg_r, *cg_dir, *cg_Ad, *cg_s;
  float avg_density_err, cg_delta, cg_delta_old, cg_delta_zero;
  double cg_dAd, cg_dnew; /* local reductions of the running PCG iteration */
  /* ---- dfsph.py module globals (ref: dfsph.py:42-48) ---- */
  float *alpha_coff, *kappa, *kappa_v;
  float *kbuf;     /* k_i = b_i * alpha_i of the running iteration (dfsph.py:456, 525) */
  int n_reduce;    /* reductions cover particles [0, n_reduce): all liquid, or the owned rows of a slab test */
  float deltaT;
  int vs_iter, dv_iter, pr_iter;
  double particleRadius, user_max_t, user_min_t; /* python-scope doubles, dfsph.py:27,39-40 */
  /* ---- HashGrid (ref: HashGrid.py:10-54) ---- */
  int maxInGrid, maxNeighbour;
  float gridR, invGridR, searchR;
  double gridR_d; /* python-scope double used by setup_grid_cpu */
  int blockSize[3];
  v3 min_boundary, max_boundary;
  int *gridCount, *grid, *neighborCount, *neighbor;
  int table_size; /* faithful: count ; clean: number of cells */
  int mode;
  int overflow_grid, overflow_neighbor;
  /* ---- CubicKernel constants (ref: kernels/CubicKernel.py:12-17) ---- */
  float ck_h, ck_h3, ck_mk, ck_ml;
  /* cohesion / adhesion (ref: kernels/CohesionKernel.py:12-15, AdhesionKernel.py:12-14) */
  float coh_mk, coh_mc, adh_mk;
  /* ---- surface reconstruction fields (ref: ParticleData.py:37-42) ---- */
  float *color; v3 *color_grad, *pos_avr; float *G; /* [liquid*9] */
  float mc_searchR;
  /* ---- pcisph.py / iisph.py private fields (ref: pcisph.py:50-64, iisph.py:56-66) ---- */
  v3 *d_vel_pre, *pos_star, *vel_star, *d_ii, *dij_pj;
  float *a_ii, *pressure_pre;
  float rho_err, pci_coff;
  float ii_visc, ii_visc_b, ii_omega, pc_visc, pc_visc_b;
  /* ---- sesph.py private state (ref: sesph.py:23-62) ---- */
  float ss_searchR, ss_mk, ss_ml, ss_VL0, ss_VS0, ss_mass, ss_rho0, ss_stiff, ss_dim, ss_visc, ss_visc_b;
} orc_ctx;

/* ------------------------------------------------------------------------------------ */
/* creation                                                                               */

/* ref: ParticleData.py:12-31 (constants), HashGrid.py:10-19, kernels/ ctors.
 * particleR is what the driver script passes to ParticleData(); dfsph passes the radius
 * (dfsph.py:66), sesph passes the DIAMETER (sesph.py:71, quirk Q11). */
orc_ctx *orc_create(int n_liquid, int n_solid, double particleR, int maxInGrid, int maxNeighbour, int mode) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->liquid_count = n_liquid; c->solid_count = n_solid; c->count = n_liquid + n_solid;
  c->rho_L0 = 1000.0f; c->rho_S0 = c->rho_L0;
  c->VL0 = (float)((double)particleR * particleR * particleR * 0.8 * 8.0);
  c->VS0 = c->VL0;
  c->liqiudMass = (float)((double)c->VL0 * (double)c->rho_L0);
  c->gravity = V(0.0f, -9.81f, 0.0f);
  c->dim_coff = 10.0f; c->viscosity = 10.0f; c->viscosity_b = 10.0f; c->viscosity_err = 0.05f;
  c->tension_coff = 0.0f; c->tension_coff_b = 0.0f;
  c->viscosity_omega = 0.1f; c->vorticity_coff = 0.01f; c->vorticity_init = 0.5f;
  c->eps = 1e-5f;
  c->particleRadius = particleR; c->user_max_t = 0.005; c->user_min_t = 0.0001;
  c->maxInGrid = maxInGrid; c->maxNeighbour = maxNeighbour; c->mode = mode;
  double gridR = (double)particleR * 2.0;           /* HashGrid(particleR*2.0, ...) ParticleData.py:27 */
  c->gridR_d = gridR; c->gridR = (float)gridR; c->invGridR = (float)(1.0 / gridR); c->searchR = (float)(gridR * 2.0);
  double h = gridR * 2.0;
  c->ck_h = (float)h; c->ck_h3 = (float)(1.0 / (h * h * h));
  c->ck_mk = (float)(8.0 / M_PI); c->ck_ml = (float)(48.0 / M_PI);
  c->coh_mk = (float)(32.0 / (M_PI * pow(h, 9.0))); c->coh_mc = (float)(pow(h, 6.0) / 64.0);
  c->adh_mk = (float)(0.007 / pow(h, 3.25));
  size_t nl = (size_t)n_liquid, n = (size_t)c->count;
  c->pos = (v3 *)calloc(n, sizeof(v3));
#define AL3(f) c->f = (v3 *)calloc(nl, sizeof(v3))
#define AL1(f) c->f = (float *)calloc(nl, sizeof(float))
  AL3(vel); AL3(vel_guess); AL3(d_vel); AL3(omega); AL3(d_omega); AL3(normal);
  AL1(rho); AL1(adv_rho); AL1(pressure); AL1(vel_max);
  c->cg_Minv = (float *)calloc(nl * 9, sizeof(float));
  AL3(cg_r); AL3(cg_dir); AL3(cg_Ad); AL3(cg_s);
  AL1(alpha_coff); AL1(kappa); AL1(kappa_v); AL1(kbuf);
  AL1(color); AL3(color_grad); AL3(pos_avr); c->G = (float *)calloc(nl * 9, sizeof(float));
  c->mc_searchR = (float)((double)particleR * 0.9 * 4.0);   /* MCGrid(particleR, ...) MarchingCubeGrid.py:22-25 */
  AL3(d_vel_pre); AL3(pos_star); AL3(vel_star); AL3(d_ii); AL3(dij_pj); AL1(a_ii); AL1(pressure_pre);
  c->n_reduce = n_liquid;
  c->neighborCount = (int *)calloc(nl, sizeof(int));
  c->neighbor = (int *)malloc(nl * (size_t)maxNeighbour * sizeof(int));
  c->deltaT = 0.0f;
  return c;
}

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  free(c->pos); free(c->vel); free(c->vel_guess); free(c->d_vel); free(c->omega); free(c->d_omega);
  free(c->normal); free(c->rho); free(c->adv_rho); free(c->pressure); free(c->vel_max); free(c->cg_Minv);
  free(c->cg_r); free(c->cg_dir); free(c->cg_Ad); free(c->cg_s); free(c->alpha_coff); free(c->kappa);
  free(c->color); free(c->color_grad); free(c->pos_avr); free(c->G); free(c->kappa_v); free(c->kbuf); free(c->d_vel_pre); free(c->pos_star); free(c->vel_star); free(c->d_ii); free(c->dij_pj); free(c->a_ii); free(c->pressure_pre); free(c->neighborCount); free(c->neighbor); free(c->gridCount); free(c->grid);
  free(c);
}

/* ref: ParticleData.py:180-185 + HashGrid.py:44-54.  pos is [count][3], liquid first.
 * maxb/minb are the running bbox of every added point (ParticleData.py:100-126). */
int orc_setup(orc_ctx *c, const float *pos, const float *maxb, const float *minb) {
  memcpy(c->pos, pos, (size_t)c->count * sizeof(v3));
  c->max_boundary = V(maxb[0], maxb[1], maxb[2]);
  c->min_boundary = V(minb[0], minb[1], minb[2]);
  for (int i = 0; i < 3; i++) /* int( (max-min)/gridR + 1 ): numpy f32 - f32, / python float => double */
    c->blockSize[i] = (int)((double)(float)(maxb[i] - minb[i]) / c->gridR_d + 1.0);
  long ncell = (long)c->blockSize[0] * c->blockSize[1] * c->blockSize[2];
  c->table_size = (c->mode == 0) ? c->count : (int)ncell;
  free(c->gridCount); free(c->grid);
  c->gridCount = (int *)calloc((size_t)c->table_size, sizeof(int));
  c->grid = (int *)malloc((size_t)c->table_size * (size_t)c->maxInGrid * sizeof(int));
  return (c->gridCount && c->grid) ? 0 : -1;
}

/* ------------------------------------------------------------------------------------ */
/* HashGrid                                                                               */

/* ref: HashGrid.py:108-114 -- int32 wrap-around products, xor, normalised modulo. */
static inline int cell_hash(const orc_ctx *c, int x, int y, int z) {
  if (c->mode == 1) return (x * c->blockSize[1] + y) * c->blockSize[2] + z; /* clean: bucket == cell */
  int32_t p1 = (int32_t)(73856093u * (uint32_t)x);
  int32_t p2 = (int32_t)(19349663u * (uint32_t)y);
  int32_t p3 = (int32_t)(83492791u * (uint32_t)z);
  int32_t n = c->count;
  return (int)((((p1 ^ p2 ^ p3) % n) + n) % n);
}
int orc_cell_hash(const orc_ctx *c, int x, int y, int z) { return cell_hash(c, x, y, z); }

/* ref: HashGrid.py:117-124 */
static inline int in_box(const orc_ctx *c, int x, int y, int z) {
  return !(x < 0 || x >= c->blockSize[0] || y < 0 || y >= c->blockSize[1] || z < 0 || z >= c->blockSize[2]);
}
/* ref: HashGrid.py:68 -- ti.cast((pos - min) * invGridR, i32): f32 ops, truncation toward zero */
static inline void cell_of(const orc_ctx *c, v3 p, int *x, int *y, int *z) {
  *x = (int)((p.x - c->min_boundary.x) * c->invGridR);
  *y = (int)((p.y - c->min_boundary.y) * c->invGridR);
  *z = (int)((p.z - c->min_boundary.z) * c->invGridR);
}
void orc_cell_of(const orc_ctx *c, int i, int *xyz) { cell_of(c, c->pos[i], xyz, xyz + 1, xyz + 2); }

/* ref: HashGrid.py:56-106.  Bucket slots end up in index order (the reference's slot order is
 * atomic-schedule dependent, Q16; any order is "a" reference run). */
void orc_update_grid(orc_ctx *c) {
  const int MIG = c->maxInGrid, MN = c->maxNeighbour;
  /* :58-64 full-table clears */
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)c->table_size; i++) {
    c->gridCount[i] = 0;
    for (int j = 0; j < MIG; j++) c->grid[i * MIG + j] = -1;
  }
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)c->liquid_count; i++) {
    c->neighborCount[i] = 0;
    for (int j = 0; j < MN; j++) c->neighbor[i * MN + j] = -1;
  }
  /* :67-76 insert.  Parallel: slots are claimed with an atomic add (as the reference's ti.atomic_add does), then every
   * bucket is sorted by particle id, which is exactly what a serial insert in index order leaves behind.  Only when a
   * bucket overflows does the order of arrival decide who is dropped: that case is redone serially. */
  int ovf_grid = 0;
#pragma omp parallel for schedule(static) reduction(+ : ovf_grid)
  for (int i = 0; i < c->count; i++) {
    int x, y, z; cell_of(c, c->pos[i], &x, &y, &z);
    if (in_box(c, x, y, z)) {
      int h = cell_hash(c, x, y, z);
      int old = __atomic_fetch_add(&c->gridCount[h], 1, __ATOMIC_RELAXED);
      if (old > MIG - 1) ovf_grid++;
      else c->grid[(long)h * MIG + old] = i;
    }
  }
  if (ovf_grid) {   /* serial redo, index order (the first MIG particles of a bucket survive) */
    for (long i = 0; i < (long)c->table_size; i++) { c->gridCount[i] = 0; for (int j = 0; j < MIG; j++) c->grid[i * MIG + j] = -1; }
    for (int i = 0; i < c->count; i++) {
      int x, y, z; cell_of(c, c->pos[i], &x, &y, &z);
      if (in_box(c, x, y, z)) {
        int h = cell_hash(c, x, y, z);
        int old = c->gridCount[h]++;
        if (old > MIG - 1) { c->gridCount[h] = MIG; c->overflow_grid++; }
        else c->grid[(long)h * MIG + old] = i;
      }
    }
  } else {
#pragma omp parallel for schedule(static)
    for (long h = 0; h < (long)c->table_size; h++) {
      int n = c->gridCount[h];
      int *g = c->grid + h * MIG;
      for (int a = 1; a < n; a++) { int v = g[a], k = a - 1; while (k >= 0 && g[k] > v) { g[k + 1] = g[k]; k--; } g[k + 1] = v; }
    }
  }
  /* :79-106 neighbour lists, liquid particles only */
  int ovf = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : ovf)
  for (int i = 0; i < c->liquid_count; i++) {
    int x, y, z; cell_of(c, c->pos[i], &x, &y, &z);
    if (!in_box(c, x, y, z)) continue;
    int cnt = 0;
    for (int m = -2; m < 3; m++) for (int n = -2; n < 3; n++) for (int q = -2; q < 3; q++) {
      int cx = x + m, cy = y + n, cz = z + q;
      if (!in_box(c, cx, cy, cz)) continue; /* :90-92 */
      int h = cell_hash(c, cx, cy, cz);
      int gc = c->gridCount[h];
      for (int k = 0; k < gc; k++) {
        int j = c->grid[(long)h * MIG + k];
        if (j >= 0 && i != j) {
          int old = cnt++;
          if (old > MN - 1) ovf++;          /* :100-103 count keeps growing, entry dropped (Q3) */
          else c->neighbor[(long)i * MN + old] = j;
        }
      }
    }
    c->neighborCount[i] = cnt;
  }
  c->overflow_neighbor += ovf;
}

/* ------------------------------------------------------------------------------------ */
/* kernels (ref: kernels/CubicKernel.py:20-54, CohesionKernel.py:17-29, AdhesionKernel.py:20-29) */

static inline v3 CubicGradW(const orc_ctx *c, v3 r) {
  v3 res = V(0, 0, 0);
  float rl = sqrtf(nsq(r));
  float q = rl / c->ck_h;
  if (rl > 1.0e-5f && q <= 1.0f) {
    v3 gradq = mul(r, 1.0f / (rl * c->ck_h));
    if (q <= 0.5f) res = mul(gradq, c->ck_ml * c->ck_h3 * q * (3.0f * q - 2.0f));
    else { float f = 1.0f - q; res = mul(gradq, -c->ck_ml * c->ck_h3 * (f * f)); }
  }
  return res;
}
static inline float Cubic_W_P(float q) {
  float res = 0.0f;
  if (q <= 1.0f) {
    if (q <= 0.5f) { float qq = q * q, qqq = qq * q; res = 6.0f * qqq - 6.0f * qq + 1.0f; }
    else { float f = 1.0f - q; res = 2.0f * f * f * f; }
  }
  return res;
}
static inline float Cubic_W_norm(const orc_ctx *c, float r) { return Cubic_W_P(r / c->ck_h) * c->ck_mk * c->ck_h3; }
static inline float Cubic_W(const orc_ctx *c, v3 r) { return Cubic_W_norm(c, sqrtf(nsq(r))); }

static inline float Coh_W_norm(const orc_ctx *c, float r) {
  float res = 0.0f, h = c->ck_h, radius2 = h * h, r2 = r * r;
  if (r2 <= radius2) {
    float r3 = r2 * r;
    if (r > 0.5f * h) res = c->coh_mk * powf(h - r, 3.0f) * r3;
    else res = c->coh_mk * 2.0f * powf(h - r, 3.0f) * r3 - c->coh_mc;
  }
  return res;
}
static inline float Adh_W_norm(const orc_ctx *c, float r) {
  float res = 0.0f, h = c->ck_h, radius2 = h * h, r2 = r * r;
  if (r2 <= radius2)
    if (r > 0.5f * h) /* fmaxf: the polynomial cancels to -1e-9 at r == h in f32 and pow() would give NaN */
      res = c->adh_mk * powf(fmaxf(-4.0f * r2 / h + 6.0f * r - 2.0f * h, 0.0f), 0.25f);
  return res;
}
/* exported scalar probes for the known-answer tests */
float orc_W_norm(const orc_ctx *c, float r) { return Cubic_W_norm(c, r); }
void orc_gradW(const orc_ctx *c, const float *r, float *out) {
  v3 g = CubicGradW(c, V(r[0], r[1], r[2])); out[0] = g.x; out[1] = g.y; out[2] = g.z;
}
float orc_coh_W(const orc_ctx *c, float r) { return Coh_W_norm(c, r); }
float orc_adh_W(const orc_ctx *c, float r) { return Adh_W_norm(c, r); }

#define NB_BEGIN(i)                                                        \
  int cur_neighbor = c->neighborCount[i];                                  \
  const int *nb_ = c->neighbor + (long)(i) * c->maxNeighbour;              \
  for (int k = 0; k < cur_neighbor; k++) {                                 \
    int j = nb_[k];                                                        \
    v3 r = sub(c->pos[i], c->pos[j]);
#define NB_END }
#define PFOR _Pragma("omp parallel for schedule(static)")
#define NL (c->liquid_count)

/* ------------------------------------------------------------------------------------ */
/* dfsph.py kernels                                                                       */

/* ref: dfsph.py:167-178 */
void orc_reset_param(orc_ctx *c) {
  for (int i = 0; i < NL; i++) {
    c->vel[i] = V(0, 0, 0); c->omega[i] = V(0, 0, 0);
    c->pressure[i] = 0.0f; c->kappa_v[i] = 0.0f; c->kappa[i] = 0.0f;
  }
  c->deltaT = 0.001f;
}

/* ref: dfsph.py:248-262 */
void orc_compute_density(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float rho = c->VL0 * Cubic_W_norm(c, 0.0f) * c->rho_L0;
    NB_BEGIN(i)
      if (j < NL) rho += c->VL0 * Cubic_W(c, r) * c->rho_L0;
      else        rho += c->VS0 * Cubic_W(c, r) * c->rho_S0;
    NB_END
    c->rho[i] = rho;
  }
}

/* ref: dfsph.py:345-372 */
void orc_compute_dfsph_coff(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    v3 sum_grad = V(0, 0, 0); float sum_grad_square = 0.0f;
    NB_BEGIN(i)
      v3 gradV = CubicGradW(c, r);
      if (j < NL) { v3 t = mul(gradV, c->VL0); sum_grad_square += nsq(t); sum_grad = add(sum_grad, t); }
      else sum_grad = add(sum_grad, mul(gradV, c->VS0));
    NB_END
    sum_grad_square += nsq(sum_grad);
    c->alpha_coff[i] = (sum_grad_square > c->eps) ? -1.0f / sum_grad_square : 0.0f;
  }
}

/* ref: dfsph.py:374-392 */
static inline void update_drho_divergence(orc_ctx *c, int i) {
  float a = 0.0f;
  NB_BEGIN(i)
    v3 gradV = CubicGradW(c, r);
    if (j < NL) a += c->VL0 * dot(sub(c->vel[i], c->vel[j]), gradV);
    else        a += c->VS0 * dot(c->vel[i], gradV);
  NB_END
  a = fmaxf(a, 0.0f);
  if (cur_neighbor < 20) a = 0.0f;
  c->adv_rho[i] = a;
}
/* ref: dfsph.py:394-412 (solid branch uses VL0, as in the reference) */
static inline void update_drho_pressure(orc_ctx *c, int i) {
  float temp = 0.0f;
  NB_BEGIN(i)
    v3 gradV = CubicGradW(c, r);
    if (j < NL) temp += c->VL0 * dot(sub(c->vel[i], c->vel[j]), gradV);
    else        temp += c->VL0 * dot(c->vel[i], gradV);
  NB_END
  float a = c->rho[i] / c->rho_L0 + c->deltaT * temp;
  c->adv_rho[i] = fmaxf(1.0f, a);
}
void orc_update_drho_divergence(orc_ctx *c) { PFOR for (int i = 0; i < NL; i++) update_drho_divergence(c, i); }
/* Test diagnostic (no reference counterpart): how close the sum of update_drho_divergence (dfsph.py:374-392) is to the
 * discontinuity that follows it -- `if adv_rho[i] > 0` (dfsph.py:421) gives a particle the whole warm-start kick or none.  On the
 * CURRENT lists, positions and velocities, per liquid particle: S = the un-clamped sum (double), A = sum |term| (what a change of
 * the summation order can move S by, times an ulp), G = sum V |gradW| (what a velocity error dv can move S by, times 2 dv). */
void orc_diag_drho_divergence(orc_ctx *c, double *S, double *A, double *G) {
  PFOR for (int i = 0; i < NL; i++) {
    double s = 0.0, a = 0.0, g = 0.0;
    NB_BEGIN(i)
      v3 gradV = CubicGradW(c, r);
      const float t = (j < NL) ? c->VL0 * dot(sub(c->vel[i], c->vel[j]), gradV) : c->VS0 * dot(c->vel[i], gradV);
      s += (double)t; a += fabs((double)t);
      g += (double)((j < NL) ? c->VL0 : c->VS0) * sqrt((double)nsq(gradV));
    NB_END
    S[i] = s; A[i] = a; G[i] = g;
  }
}
void orc_update_drho_pressure(orc_ctx *c) { PFOR for (int i = 0; i < NL; i++) update_drho_pressure(c, i); }

/* ref: dfsph.py:415-439, first parallel loop (:417-418) */
void orc_warmstart_divergence_drho(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->kappa_v[i] = 0.5f * fmaxf(c->kappa_v[i] / c->deltaT, -0.5f * c->rho_L0 * c->rho_L0);
    update_drho_divergence(c, i);
  }
}
/* neighbour loop shared by warmstart_divergence_vel (:420-439, kb = kappa_v, only where adv_rho > 0),
 * divergence_iter (:460-471) and pressure_iter (:529-541) (kb = k of the iteration):
 * vel[i] written, kb[j] read (never vel[j]) => order free */
static void vel_from_k(orc_ctx *c, const float *kb, int only_positive_adv) {
  PFOR for (int i = 0; i < NL; i++) {
    if (only_positive_adv && !(c->adv_rho[i] > 0.0f)) continue;
    v3 v = c->vel[i];
    float ki = kb[i];
    NB_BEGIN(i)
      v3 gradV = CubicGradW(c, r);
      if (j < NL) {
        float sum = ki + kb[j];
        if (fabsf(sum) > c->eps) v = add(v, mul(gradV, c->deltaT * sum * c->VL0));
      } else if (fabsf(ki) > c->eps) v = add(v, mul(gradV, c->deltaT * ki * c->VS0));
    NB_END
    c->vel[i] = v;
  }
}
void orc_warmstart_divergence_velloop(orc_ctx *c) { vel_from_k(c, c->kappa_v, 1); }
void orc_warmstart_divergence_vel(orc_ctx *c) {
  orc_warmstart_divergence_drho(c);
  orc_warmstart_divergence_velloop(c);
}

/* ref: dfsph.py:441-446 */
void orc_begin_divergence_iter(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    update_drho_divergence(c, i);
    c->alpha_coff[i] = c->alpha_coff[i] / c->deltaT;
    c->kappa_v[i] = 0.0f;
  }
}

/* ref: dfsph.py:449-477, split at the loop boundaries */
void orc_make_k(orc_ctx *c, float *kap, float bias) { /* :454-458 / :523-527: k_i = b_i alpha_i ; kappa_i += k_i */
  PFOR for (int i = 0; i < NL; i++) {
    float k = (c->adv_rho[i] - bias) * c->alpha_coff[i];
    c->kbuf[i] = k;
    kap[i] += k;
  }
}
void orc_make_k_div(orc_ctx *c) { orc_make_k(c, c->kappa_v, 0.0f); }
void orc_make_k_press(orc_ctx *c) { orc_make_k(c, c->kappa, 1.0f); }
void orc_vel_iter(orc_ctx *c) { vel_from_k(c, c->kbuf, 0); } /* :460-471 / :529-541 */
void orc_drho_div_iter(orc_ctx *c) { /* :474-477 */
  double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err)
  for (int i = 0; i < NL; i++) { update_drho_divergence(c, i); if (i < c->n_reduce) err += (double)c->adv_rho[i]; }
  c->avg_density_err = (float)err;
}
void orc_divergence_iter(orc_ctx *c) {
  orc_make_k_div(c);
  orc_vel_iter(c);
  orc_drho_div_iter(c);
}

/* ref: dfsph.py:480-484 */
void orc_end_divergence_iter(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) { c->kappa_v[i] *= c->deltaT; c->alpha_coff[i] *= c->deltaT; }
}

/* ref: dfsph.py:487-508 (kept literally, including the kappa_v typo in the solid branch, Q7) */
void orc_warmstart_pressure_scale(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++)
    c->kappa[i] = fmaxf(c->kappa[i] / c->deltaT / c->deltaT, -0.5f * c->rho_L0 * c->rho_L0);
}
void orc_warmstart_pressure_vel(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    if (c->adv_rho[i] > c->rho_L0) {
      v3 v = c->vel[i];
      NB_BEGIN(i)
        v3 gradV = CubicGradW(c, r);
        if (j < NL) {
          float sum = c->kappa[i] + c->kappa[j];
          if (fabsf(sum) > c->eps) v = add(v, mul(gradV, c->deltaT * sum * c->VL0));
        } else if (fabsf(c->kappa[i]) > c->eps) v = add(v, mul(gradV, c->deltaT * c->kappa_v[i] * c->VS0));
      NB_END
      c->vel[i] = v;
    }
  }
}
void orc_warmstart_pressure(orc_ctx *c) { orc_warmstart_pressure_scale(c); orc_warmstart_pressure_vel(c); }

/* ref: dfsph.py:511-516 */
void orc_begin_pressure_iter(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    update_drho_pressure(c, i);
    c->alpha_coff[i] = c->alpha_coff[i] / c->deltaT / c->deltaT;
    c->kappa[i] = 0.0f;
  }
}

/* ref: dfsph.py:518-547, split at the loop boundaries (first loop: orc_make_k_press + orc_vel_iter) */
void orc_drho_press_iter(orc_ctx *c) { /* :544-547 */
  double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err)
  for (int i = 0; i < NL; i++) { update_drho_pressure(c, i); if (i < c->n_reduce) err += (double)(c->adv_rho[i] - 1.0f); }
  c->avg_density_err = (float)err;
}
void orc_pressure_iter(orc_ctx *c) {
  orc_make_k_press(c);
  orc_vel_iter(c);
  orc_drho_press_iter(c);
}

/* ref: dfsph.py:549-552 */
void orc_end_pressure_iter(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) c->kappa[i] *= c->deltaT * c->deltaT;
}

/* ref: dfsph.py:333-336 */
void orc_clear_nonpressure(orc_ctx *c) { PFOR for (int i = 0; i < NL; i++) c->d_vel[i] = c->gravity; }

/* ref: dfsph.py:264-305.
 * tension_fixed == 0 reproduces the loop literally (normal *= h once per entry, Q8; the
 * curvature term over EVERY liquid list entry, cohesion overwritten, Q9).
 * tension_fixed == 1 is the documented repair: n_i = h * sum (m/rho_j) gradW, curvature
 * term only for entries inside the support. */
void orc_tension_normal(orc_ctx *c) {
  const float h = c->ck_h;
  PFOR for (int i = 0; i < NL; i++) {
    v3 n = V(0, 0, 0);
    NB_BEGIN(i)
      v3 g = CubicGradW(c, r);
      if (j < NL) n = add(n, mul(g, c->liqiudMass / c->rho[j]));
      if (!c->tension_fixed) n = mul(n, h);
    NB_END
    if (c->tension_fixed) n = mul(n, h);
    c->normal[i] = n;
  }
}
void orc_tension_accel(orc_ctx *c) {
  const float h = c->ck_h;
  PFOR for (int i = 0; i < NL; i++) {
    v3 dv = c->d_vel[i];
    NB_BEGIN(i)
      float len2 = nsq(r);
      if (j < NL) {
        float k_ij = 2.0f * c->rho_L0 / (c->rho[i] + c->rho[j]);
        /* cohesion value is computed and discarded in the reference (Q9) */
        v3 accel = mul(sub(c->normal[i], c->normal[j]), -c->tension_coff);
        if (!c->tension_fixed || len2 <= h * h) dv = add(dv, mul(accel, k_ij));
      } else {
        v3 d = sub(c->pos[j], V(0.0f, 0.5f, 0.0f));
        float dis = sqrtf(nsq(d));
        if (len2 > c->eps && dis < 0.26f) {
          v3 xixj = mul(r, 1.0f / sqrtf(len2));
          dv = add(dv, mul(xixj, -c->tension_coff_b * c->rho_S0 * c->VS0 * Adh_W_norm(c, sqrtf(len2))));
        }
      }
    NB_END
    c->d_vel[i] = dv;
  }
}
void orc_compute_tension(orc_ctx *c) { orc_tension_normal(c); orc_tension_accel(c); }

/* ref: dfsph.py:181-195.  x is vel_guess or cg_dir. */
static inline v3 viscosity_Ax(const orc_ctx *c, const v3 *x, int i) {
  v3 ret = V(0, 0, 0);
  const float e = 0.01f * c->searchR * c->searchR;
  NB_BEGIN(i)
    if (j < NL) {
      float s = c->dim_coff * c->viscosity * c->liqiudMass / c->rho[j] * dot(sub(x[i], x[j]), r) / (nsq(r) + e);
      ret = add(ret, mul(CubicGradW(c, r), s / c->rho[i] * c->deltaT));
    } else {
      float s = c->dim_coff * c->viscosity_b * c->rho_S0 / c->rho[i] * c->VS0 * dot(x[i], r) / (nsq(r) + e);
      ret = add(ret, mul(CubicGradW(c, r), s / c->rho[i] * c->deltaT));
    }
  NB_END
  return sub(x[i], ret);
}

static void inv3(const float *m, float *o) { /* adjugate / determinant (Taichi Matrix.inverse 3x3) */
  float a = m[0], b = m[1], cc = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  float det = a * A + b * B + cc * C, id = 1.0f / det;
  o[0] = A * id; o[1] = -(b * i - cc * h) * id; o[2] = (b * f - cc * e) * id;
  o[3] = B * id; o[4] = (a * i - cc * g) * id;  o[5] = -(a * f - cc * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id;  o[8] = (a * e - b * d) * id;
}
static inline v3 matvec(const float *m, v3 v) {
  return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

/* ref: dfsph.py:197-223 */
void orc_visc_guess(orc_ctx *c) { /* :199-200 */
  PFOR for (int i = 0; i < NL; i++) c->vel_guess[i] = add(c->vel_guess[i], c->vel[i]);
}
void orc_visc_minv(orc_ctx *c) { /* :202-215 */
  const float e = 0.01f * c->searchR * c->searchR;
  PFOR for (int i = 0; i < NL; i++) {
    float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    NB_BEGIN(i)
      v3 g = CubicGradW(c, r);
      float s;
      if (j < NL) s = c->dim_coff * c->viscosity * c->liqiudMass / c->rho[j] / (nsq(r) + e);
      else        s = c->dim_coff * c->viscosity_b * c->rho_S0 / c->rho[i] * c->VS0 / (nsq(r) + e);
      /* outer_product(gradW, r)[a][b] = g_a * r_b */
      m[0] += s * g.x * r.x; m[1] += s * g.x * r.y; m[2] += s * g.x * r.z;
      m[3] += s * g.y * r.x; m[4] += s * g.y * r.y; m[5] += s * g.y * r.z;
      m[6] += s * g.z * r.x; m[7] += s * g.z * r.y; m[8] += s * g.z * r.z;
    NB_END
    float f = c->deltaT / c->rho[i], t[9];
    for (int a = 0; a < 9; a++) t[a] = ((a % 4 == 0) ? 1.0f : 0.0f) - m[a] * f;
    inv3(t, c->cg_Minv + (long)i * 9);
  }
}
void orc_visc_residual(orc_ctx *c) { /* :217-223 */
  double dz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dz)
  for (int i = 0; i < NL; i++) {
    c->cg_r[i] = sub(c->vel[i], viscosity_Ax(c, c->vel_guess, i));
    c->cg_dir[i] = matvec(c->cg_Minv + (long)i * 9, c->cg_r[i]);
    if (i < c->n_reduce) dz += (double)dot(c->cg_r[i], c->cg_dir[i]);
  }
  c->cg_delta_zero = (float)dz;
  c->cg_delta = c->cg_delta_zero;
}
void orc_init_viscosity_para(orc_ctx *c) { orc_visc_guess(c); orc_visc_minv(c); orc_visc_residual(c); }

/* ref: dfsph.py:225-246, split at the loop boundaries.  cg_dAd is the kernel-local scalar of the reference. */
void orc_visc_Ad(orc_ctx *c) { /* :227-230 */
  double dAd = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dAd)
  for (int i = 0; i < NL; i++) {
    c->cg_Ad[i] = viscosity_Ax(c, c->cg_dir, i);
    if (i < c->n_reduce) dAd += (double)dot(c->cg_dir[i], c->cg_Ad[i]);
  }
  c->cg_dAd = dAd;   /* without the eps start value; added where it is used */
}
void orc_visc_update(orc_ctx *c) { /* :233-240 */
  float alpha = c->cg_delta / (float)((double)c->eps + c->cg_dAd);
  double dl = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dl)
  for (int i = 0; i < NL; i++) {
    c->vel_guess[i] = add(c->vel_guess[i], mul(c->cg_dir[i], alpha));
    c->cg_r[i] = sub(c->cg_r[i], mul(c->cg_Ad[i], alpha));
    c->cg_s[i] = matvec(c->cg_Minv + (long)i * 9, c->cg_r[i]);
    if (i < c->n_reduce) dl += (double)dot(c->cg_r[i], c->cg_s[i]);
  }
  c->cg_dnew = dl;
}
void orc_visc_dir(orc_ctx *c) { /* :234-235, 243-246 */
  c->cg_delta_old = c->cg_delta;
  c->cg_delta = (float)c->cg_dnew;
  float beta = c->cg_delta / c->cg_delta_old;
  PFOR for (int i = 0; i < NL; i++) c->cg_dir[i] = add(c->cg_s[i], mul(c->cg_dir[i], beta));
}
void orc_compute_viscosity_force(orc_ctx *c) { orc_visc_Ad(c); orc_visc_update(c); orc_visc_dir(c); }

/* ref: dfsph.py:339-343 */
void orc_end_viscosity(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->d_vel[i] = add(c->d_vel[i], mul(sub(c->vel_guess[i], c->vel[i]), 1.0f / c->deltaT));
    c->vel_guess[i] = sub(c->vel_guess[i], c->vel[i]);
  }
}

/* ref: dfsph.py:307-330.  Q18: for a solid j the reference indexes omega[j] / vel[j] out of
 * bounds; restated with omega_j = vel_j = 0 (static wall) -- unreproducible otherwise.
 * Q10: the damping term is added once per list entry. */
void orc_vorticity_accumulate(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    v3 dom = V(0, 0, 0), dv = c->d_vel[i];
    NB_BEGIN(i)
      v3 gradV = CubicGradW(c, r);
      if (j < NL) {
        v3 dw = sub(c->omega[i], c->omega[j]);
        dom = add(dom, mul(dw, -1.0f / c->deltaT * c->vorticity_init * c->viscosity_omega * (c->liqiudMass / c->rho[j]) * Cubic_W(c, r)));
        dv = add(dv, mul(cross(dw, gradV), c->vorticity_coff / c->rho[i] * c->liqiudMass));
        dom = add(dom, mul(cross(sub(c->vel[i], c->vel[j]), gradV), c->vorticity_coff / c->rho[i] * c->vorticity_init * c->liqiudMass));
      } else {
        dv = add(dv, mul(cross(c->omega[i], gradV), c->vorticity_coff / c->rho[i] * c->rho_L0 * c->VS0));
        dom = add(dom, mul(cross(c->vel[i], gradV), c->vorticity_coff / c->rho[i] * c->vorticity_init * c->rho_L0 * c->VL0));
      }
      dom = add(dom, mul(c->omega[i], -2.0f * c->vorticity_init * c->vorticity_coff));
    NB_END
    c->d_omega[i] = dom; c->d_vel[i] = dv;
  }
}
void orc_vorticity_integrate(orc_ctx *c) { /* :329-330 */
  PFOR for (int i = 0; i < NL; i++) c->omega[i] = add(c->omega[i], mul(c->d_omega[i], c->deltaT));
}
void orc_compute_vorticity(orc_ctx *c) { orc_vorticity_accumulate(c); orc_vorticity_integrate(c); }

/* ref: dfsph.py:555-568 + host loop :107-111.  Literal strided tree (covers the first
 * 2^floor(log2(N-1)) particles, Q5); out-of-range reads return the element itself. */
void orc_cfl_reduce(orc_ctx *c) {
  int N = NL;
  if (c->mode == 1) { /* clean: the true maximum (over the reduction range) lands in slot 0 */
    float m = 0.0f;
    for (int i = 0; i < N; i++) {
      c->vel_max[i] = fmaxf(nsq(add(c->vel[i], mul(c->d_vel[i], c->deltaT))), 0.1f);
      if (i < c->n_reduce && c->vel_max[i] > m) m = c->vel_max[i];
    }
    if (N > 0) c->vel_max[0] = m;
    return;
  }
  for (int size = 1; size < N; size *= 2) {
    if (size == 1) {
      PFOR for (int i = 0; i < N; i++)
        c->vel_max[i] = fmaxf(nsq(add(c->vel[i], mul(c->d_vel[i], c->deltaT))), 0.1f);
    } else {
      int off = size / 2;
      PFOR for (int i = 0; i < N; i += size) {
        float vi = c->vel_max[i];
        float vj = (i + off < N) ? c->vel_max[i + off] : vi;
        c->vel_max[i] = (vi > vj) ? vi : vj;
      }
    }
  }
}

/* ref: dfsph.py:107-129 (host side in Python doubles on f32 values) */
void orc_optimize_time_step(orc_ctx *c) {
  orc_cfl_reduce(c);
  float vmax0 = c->vel_max[0];
  if ((double)vmax0 > (double)c->eps) {
    double time_step = 0.5 * 0.4 * c->particleRadius * 2.0 / sqrt((double)vmax0);
    if (time_step > c->user_max_t) time_step = c->user_max_t;
    if (time_step < c->user_min_t) time_step = c->user_min_t;
    int it = c->vs_iter > c->pr_iter ? c->vs_iter : c->pr_iter; /* max(vs, max(pr, vs)) Q6 */
    float dt = c->deltaT;
    if (it > 10) dt = (float)((double)dt * 0.9);
    else if (it < 5) dt = (float)((double)dt * 1.1);
    if ((double)dt > time_step) dt = (float)time_step;
    c->deltaT = dt;
  }
}

/* ref: dfsph.py:572-580 */
void orc_update_vel(orc_ctx *c) { PFOR for (int i = 0; i < NL; i++) c->vel[i] = add(c->vel[i], mul(c->d_vel[i], c->deltaT)); }
void orc_update_pos(orc_ctx *c) { PFOR for (int i = 0; i < NL; i++) c->pos[i] = add(c->pos[i], mul(c->vel[i], c->deltaT)); }

/* ref: dfsph.py:84-103 */
void orc_compute_nonpressure_force(orc_ctx *c) {
  orc_clear_nonpressure(c);
  orc_compute_tension(c);
  orc_init_viscosity_para(c);
  c->vs_iter = 0;
  while (c->vs_iter < 100) {
    orc_compute_viscosity_force(c);
    c->vs_iter++;
    if ((double)c->cg_delta <= (double)c->viscosity_err * (double)c->cg_delta_zero || (double)c->cg_delta_zero < (double)c->eps) break;
  }
  orc_end_viscosity(c);
  orc_compute_vorticity(c);
}

/* ref: dfsph.py:131-146 */
void orc_solve_vel_divergence(orc_ctx *c) {
  c->dv_iter = 0;
  orc_warmstart_divergence_vel(c);
  double err = -0.1;
  orc_begin_divergence_iter(c);
  float dt = c->deltaT;
  while ((double)c->avg_density_err > err && c->dv_iter < 10) {
    orc_divergence_iter(c);
    err = 0.001 * (double)NL / (double)dt;
    c->dv_iter++;
  }
  orc_end_divergence_iter(c);
}

/* ref: dfsph.py:150-164 */
void orc_solve_pressure(orc_ctx *c) {
  orc_warmstart_pressure(c);
  c->pr_iter = 0;
  double err = 0.0;
  orc_begin_pressure_iter(c);
  while ((err > 0.001 || c->pr_iter < 2) && c->pr_iter < 100) {
    orc_pressure_iter(c);
    err = (double)c->avg_density_err / (double)NL;
    c->pr_iter++;
  }
  orc_end_pressure_iter(c);
}

/* ref: dfsph.py:606-617 main-loop body */
void orc_dfsph_step(orc_ctx *c) {
  orc_update_grid(c);
  orc_compute_density(c);
  orc_compute_dfsph_coff(c);
  orc_solve_vel_divergence(c);
  orc_compute_nonpressure_force(c);
  orc_optimize_time_step(c);
  orc_update_vel(c);
  orc_solve_pressure(c);
  orc_update_pos(c);
}

/* ------------------------------------------------------------------------------------ */
/* sesph.py (private constants with pi = 3.1415926, private vel/d_vel/rho/pressure fields;
 * here they alias the ParticleData arrays of the same names)                             */

/* ref: sesph.py:23-62.  searchR = gridR*2 where gridR = 2*particleRadius (0.05). */
void orc_sesph_init(orc_ctx *c, double particleRadius) {
  double gridR = (double)particleRadius * 2.0, sr = gridR * 2.0, pi = 3.1415926;
  double h3 = sr * sr * sr;
  c->ss_searchR = (float)sr; c->ss_mk = (float)(8.0 / (pi * h3)); c->ss_ml = (float)(48.0 / (pi * h3));
  double VL0 = (double)particleRadius * particleRadius * particleRadius * 0.8 * 8.0;
  c->ss_VL0 = (float)VL0; c->ss_VS0 = (float)(VL0 * 2.0); c->ss_rho0 = 1000.0f; c->ss_mass = (float)(VL0 * 1000.0);
  c->ss_stiff = 50000.0f; c->ss_dim = 10.0f; c->ss_visc = 0.1f; c->ss_visc_b = 0.0f;
}
static inline v3 ss_gradW(const orc_ctx *c, v3 r) { /* ref: sesph.py:96-107 */
  v3 res = V(0, 0, 0);
  float rl = sqrtf(nsq(r)), q = rl / c->ss_searchR;
  if (rl > 1.0e-5f && q <= 1.0f) {
    v3 gradq = mul(r, 1.0f / (rl * c->ss_searchR));
    if (q <= 0.5f) res = mul(gradq, c->ss_ml * q * (3.0f * q - 2.0f));
    else { float f = 1.0f - q; res = mul(gradq, -c->ss_ml * (f * f)); }
  }
  return res;
}
static inline float ss_W_norm(const orc_ctx *c, float v) { /* ref: sesph.py:110-124 */
  float res = 0.0f, q = v / c->ss_searchR;
  if (q <= 1.0f) {
    if (q <= 0.5f) { float qq = q * q, qqq = qq * q; res = c->ss_mk * (6.0f * qqq - 6.0f * qq + 1.0f); }
    else { float f = 1.0f - q; res = c->ss_mk * 2.0f * f * f * f; }
  }
  return res;
}
/* ref: sesph.py:130-135 */
void orc_sesph_reset_param(orc_ctx *c) {
  for (int i = 0; i < NL; i++) { c->vel[i] = V(0, 0, 0); c->pressure[i] = 0.0f; }
  c->deltaT = 0.001f;
}
/* ref: sesph.py:138-154 */
void orc_sesph_update_advection_density(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float rho = c->ss_VL0 * ss_W_norm(c, 0.0f);
    NB_BEGIN(i)
      float d_den = ss_W_norm(c, sqrtf(nsq(r)));
      if (j < NL) rho += c->ss_VL0 * d_den; else rho += c->ss_VS0 * d_den;
    NB_END
    c->rho[i] = rho * c->ss_rho0;
  }
}
/* ref: sesph.py:158-166 */
void orc_sesph_update_pressure(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->rho[i] = fmaxf(c->rho[i], c->ss_rho0);
    float q = c->rho[i] / c->ss_rho0, qq = q * q, qqqq = qq * qq;
    c->pressure[i] = c->ss_stiff * (qqqq * qq * q - 1.0f);
  }
}
/* ref: sesph.py:168-188 */
void orc_sesph_compute_force(orc_ctx *c) {
  const float e = 0.01f * c->ss_searchR * c->ss_searchR, rho_0 = c->ss_rho0;
  PFOR for (int i = 0; i < NL; i++) {
    v3 a = c->gravity;
    NB_BEGIN(i)
      v3 grad = ss_gradW(c, r);
      if (j < NL) {
        a = add(a, mul(grad, c->ss_dim * c->ss_visc * c->ss_mass / c->rho[j] * dot(sub(c->vel[i], c->vel[j]), r) / (nsq(r) + e)));
        a = add(a, mul(grad, -rho_0 * c->ss_VL0 * (c->pressure[i] / (c->rho[i] * c->rho[i]) + c->pressure[j] / (c->rho[j] * c->rho[j]))));
      } else {
        a = add(a, mul(grad, c->ss_dim * c->ss_visc_b * c->ss_VS0 * (c->rho[i] / rho_0) * dot(c->vel[i], r) / (nsq(r) + e)));
        a = add(a, mul(grad, -rho_0 * c->ss_VS0 * (c->pressure[i] / (c->rho[i] * c->rho[i]) + c->pressure[i] / (rho_0 * rho_0))));
      }
    NB_END
    c->d_vel[i] = a;
  }
}
/* ref: sesph.py:191-196 */
void orc_sesph_integrate(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->vel[i] = add(c->vel[i], mul(c->d_vel[i], c->deltaT));
    c->pos[i] = add(c->pos[i], mul(c->vel[i], c->deltaT));
  }
}
/* ref: sesph.py:217-225 */
void orc_sesph_step(orc_ctx *c) {
  orc_update_grid(c);
  orc_sesph_update_advection_density(c);
  orc_sesph_update_pressure(c);
  orc_sesph_compute_force(c);
  orc_sesph_integrate(c);
}


/* ------------------------------------------------------------------------------------ */
/* pcisph.py (ref: pcisph.py:30-64 constants, 87-115 GetPciCoff, 147-285 kernels).  Uses the sesph-style
 * private kernel constants (ss_*: pi = 3.1415926, VS0 = 2 VL0) set by orc_sesph_init().                 */

/* ref: pcisph.py:75-115 -- host-side, float64 NumPy in the reference */
double orc_pci_coff(double particleRadius) {
  const double gridR = particleRadius * 2.0, sr = gridR * 2.0, pi = 3.1415926, h3 = sr * sr * sr, m_l = 48.0 / (pi * h3);
  const double diam = 2.0 * particleRadius;
  double sg[3] = {0, 0, 0}, sg2 = 0.0;
  const double V00 = particleRadius * particleRadius * particleRadius * 0.8 * 8.0;
  double xj[3] = {-sr, -sr, -sr};
  while (xj[0] <= sr) {
    while (xj[1] <= sr) {
      while (xj[2] <= sr) {
        double r[3] = {-xj[0], -xj[1], -xj[2]};
        double dist = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (dist < sr) {
          double g[3] = {0, 0, 0}, q = dist / sr;
          if (dist > 1.0e-5 && q <= 1.0) {
            double f = (q <= 0.5) ? m_l * q * (3.0 * q - 2.0) : -m_l * (1.0 - q) * (1.0 - q);
            for (int a = 0; a < 3; a++) g[a] = f * r[a] / (dist * sr);
          }
          for (int a = 0; a < 3; a++) sg[a] += g[a];
          double dg = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
          sg2 += dg * dg;
        }
        xj[2] += diam;
      }
      xj[1] += diam; xj[2] = -sr;
    }
    xj[0] += diam; xj[1] = -sr; xj[2] = -sr;
  }
  const double beta = 2.0 * V00 * V00;
  const double ds = sqrt(sg[0] * sg[0] + sg[1] * sg[1] + sg[2] * sg[2]);
  return 1.0 / (beta * (ds * ds + sg2));
}
void orc_pcisph_init(orc_ctx *c, double particleRadius) {
  orc_sesph_init(c, particleRadius);
  c->pci_coff = (float)orc_pci_coff(particleRadius);
  c->pc_visc = 0.05f; c->pc_visc_b = 0.0f;   /* pcisph.py:68-70 */
}
/* ref: pcisph.py:193-197 */
void orc_pcisph_reset_param(orc_ctx *c) {
  for (int i = 0; i < NL; i++) c->vel[i] = V(0, 0, 0);
  c->deltaT = 0.001f;
}
/* ref: pcisph.py:199-218.  The reference reads rho[j] inside the loop that is still writing rho (a data race:
 * any mix of last step's, partial and final values).  Restated two-phase: densities first, then the viscous term
 * with the finished rho_j of THIS step.  rho[i] in the solid term is the running partial sum of
 * the reference; restated with the finished rho_i (the term is multiplied by viscosity_b = 0). */
void orc_pcisph_compute_nonpressure_force(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float rho = c->ss_VL0 * ss_W_norm(c, 0.0f) * c->ss_rho0;
    NB_BEGIN(i)
      float w = ss_W_norm(c, sqrtf(nsq(r)));
      rho += ((j < NL) ? c->ss_VL0 : c->ss_VS0) * w * c->ss_rho0;
    NB_END
    c->rho[i] = rho;
  }
  const float e = 0.01f * c->ss_searchR * c->ss_searchR;
  PFOR for (int i = 0; i < NL; i++) {
    v3 a = c->gravity;
    NB_BEGIN(i)
      v3 g = ss_gradW(c, r);
      if (j < NL) a = add(a, mul(g, c->ss_dim * c->pc_visc * c->ss_mass / c->rho[j] * dot(sub(c->vel[i], c->vel[j]), r) / (nsq(r) + e)));
      else        a = add(a, mul(g, c->ss_dim * c->pc_visc_b * c->ss_VS0 * (c->rho[i] / c->ss_rho0) * dot(c->vel[i], r) / (nsq(r) + e)));
    NB_END
    c->d_vel[i] = a;
  }
}
/* ref: pcisph.py:220-226 */
void orc_pcisph_init_iter_info(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->vel_star[i] = c->vel[i]; c->pos_star[i] = c->pos[i]; c->pressure[i] = 0.0f; c->d_vel_pre[i] = V(0, 0, 0);
  }
}
/* ref: pcisph.py:228-235 (rho_err[i] = 0 on a shape-(1) field: only element 0 exists, Q12) */
void orc_pcisph_update_iter_info(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->vel_star[i] = add(c->vel[i], mul(add(c->d_vel[i], c->d_vel_pre[i]), c->deltaT));
    c->pos_star[i] = add(c->pos[i], mul(c->vel_star[i], c->deltaT));
    c->pressure[i] = 0.0f;
  }
  c->rho_err = 0.0f;
}
/* ref: pcisph.py:237-278.  The density prediction uses the CURRENT positions (Q13), the pressure force
 * mixes pos[i] with pos_star[j] -- both kept literally. */
void orc_pcisph_predict_density(orc_ctx *c) {
  double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err)
  for (int i = 0; i < NL; i++) {
    float a = c->ss_VL0 * ss_W_norm(c, 0.0f);
    NB_BEGIN(i)
      float w = ss_W_norm(c, sqrtf(nsq(r)));
      a += ((j < NL) ? c->ss_VL0 : c->ss_VS0) * w;
    NB_END
    a = fmaxf(a, 1.0f);
    c->adv_rho[i] = a;
    c->pressure[i] += c->pci_coff * (a - 1.0f) / (c->deltaT * c->deltaT);
    if (i < c->n_reduce) err += (double)(a - 1.0f);
  }
  c->rho_err = (float)((double)c->rho_err + err);
  PFOR for (int i = 0; i < NL; i++) {
    v3 a = V(0, 0, 0);
    int cur = c->neighborCount[i];
    const int *nb = c->neighbor + (long)i * c->maxNeighbour;
    for (int k = 0; k < cur; k++) {
      int j = nb[k];
      v3 pj = (j < NL) ? c->pos_star[j] : c->pos[j];
      v3 g = ss_gradW(c, sub(c->pos[i], pj));
      float dpi = c->pressure[i];
      if (j < NL) a = add(a, mul(g, -c->ss_VL0 * (dpi + c->pressure[j])));
      else        a = add(a, mul(g, -c->ss_VS0 * dpi));
    }
    c->d_vel_pre[i] = a;
  }
}
/* ref: pcisph.py:147-157 */
void orc_pcisph_solve_pressure(orc_ctx *c) {
  c->pr_iter = 0;
  double err = 0.0;
  orc_pcisph_init_iter_info(c);
  while ((err > 0.01 || c->pr_iter < 3) && c->pr_iter < 50) {
    orc_pcisph_update_iter_info(c);
    orc_pcisph_predict_density(c);
    err = (double)c->rho_err / (double)c->n_reduce;
    c->pr_iter++;
  }
}
/* ref: pcisph.py:281-285 */
void orc_pcisph_update_pos(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->vel[i] = add(c->vel[i], mul(add(c->d_vel[i], c->d_vel_pre[i]), c->deltaT));
    c->pos[i] = add(c->pos[i], mul(c->vel[i], c->deltaT));
  }
}
/* ref: pcisph.py:303-312 main-loop body */
void orc_pcisph_step(orc_ctx *c) {
  orc_update_grid(c);
  orc_pcisph_compute_nonpressure_force(c);
  orc_pcisph_solve_pressure(c);
  orc_pcisph_update_pos(c);
}

/* ------------------------------------------------------------------------------------ */
/* iisph.py (ref: iisph.py:24-96 constants, 114-139 host loops, 184-396 kernels); private kernel constants as
 * sesph (pi = 3.1415926) but VS0 = VL0 (iisph.py:37).                                                    */
void orc_iisph_init(orc_ctx *c, double particleRadius) {
  orc_sesph_init(c, particleRadius);
  c->ss_VS0 = c->ss_VL0;                         /* iisph.py:37 */
  c->ii_visc = 2.0f; c->ii_visc_b = 3.0f; c->ii_omega = 0.5f;   /* iisph.py:80-83 */
}
/* ref: iisph.py:178-182 */
void orc_iisph_reset_param(orc_ctx *c) {
  for (int i = 0; i < NL; i++) { c->vel[i] = V(0, 0, 0); c->pressure[i] = 0.0f; }
  c->deltaT = 0.001f;
}
/* ref: iisph.py:254-268 */
void orc_iisph_compute_density(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float rho = c->ss_VL0 * ss_W_norm(c, 0.0f) * c->ss_rho0;
    NB_BEGIN(i)
      float w = ss_W_norm(c, sqrtf(nsq(r)));
      rho += ((j < NL) ? c->ss_VL0 : c->ss_VS0) * w * c->ss_rho0;
    NB_END
    c->rho[i] = rho;
  }
}
/* ref: iisph.py:184-198 */
static inline v3 ii_viscosity_Ax(const orc_ctx *c, const v3 *x, int i) {
  v3 ret = V(0, 0, 0);
  const float e = 0.01f * c->ss_searchR * c->ss_searchR;
  NB_BEGIN(i)
    if (j < NL) {
      float s = c->ss_dim * c->ii_visc * c->ss_mass / c->rho[j] * dot(sub(x[i], x[j]), r) / (nsq(r) + e);
      ret = add(ret, mul(ss_gradW(c, r), s / c->rho[i] * c->deltaT));
    } else {
      float s = c->ss_dim * c->ii_visc_b * c->ss_rho0 / c->rho[i] * c->ss_VS0 * dot(x[i], r) / (nsq(r) + e);
      ret = add(ret, mul(ss_gradW(c, r), s / c->rho[i] * c->deltaT));
    }
  NB_END
  return sub(x[i], ret);
}
/* ref: iisph.py:200-229 */
void orc_iisph_init_viscosity_para(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) c->vel_guess[i] = add(c->vel_guess[i], c->vel[i]);
  const float e = 0.01f * c->ss_searchR * c->ss_searchR;
  PFOR for (int i = 0; i < NL; i++) {
    float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    NB_BEGIN(i)
      v3 g = ss_gradW(c, r);
      float s;
      if (j < NL) s = c->ss_dim * c->ii_visc * c->ss_mass / c->rho[j] / (nsq(r) + e);
      else        s = c->ss_dim * c->ii_visc_b * c->ss_rho0 / c->rho[i] * c->ss_VS0 / (nsq(r) + e);
      m[0] += s * g.x * r.x; m[1] += s * g.x * r.y; m[2] += s * g.x * r.z;
      m[3] += s * g.y * r.x; m[4] += s * g.y * r.y; m[5] += s * g.y * r.z;
      m[6] += s * g.z * r.x; m[7] += s * g.z * r.y; m[8] += s * g.z * r.z;
    NB_END
    float f = c->deltaT / c->rho[i], t[9];
    for (int a = 0; a < 9; a++) t[a] = ((a % 4 == 0) ? 1.0f : 0.0f) - m[a] * f;
    inv3(t, c->cg_Minv + (long)i * 9);
  }
  double dz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dz)
  for (int i = 0; i < NL; i++) {
    c->cg_r[i] = sub(c->vel[i], ii_viscosity_Ax(c, c->vel_guess, i));
    c->cg_dir[i] = matvec(c->cg_Minv + (long)i * 9, c->cg_r[i]);
    dz += (double)dot(c->cg_r[i], c->cg_dir[i]);
  }
  c->cg_delta_zero = (float)dz;
  c->cg_delta = c->cg_delta_zero;
}
/* ref: iisph.py:231-252 */
void orc_iisph_compute_viscosity_force(orc_ctx *c) {
  double dAd = (double)c->eps;
#pragma omp parallel for schedule(static) reduction(+ : dAd)
  for (int i = 0; i < NL; i++) {
    c->cg_Ad[i] = ii_viscosity_Ax(c, c->cg_dir, i);
    dAd += (double)dot(c->cg_dir[i], c->cg_Ad[i]);
  }
  float alpha = c->cg_delta / (float)dAd;
  c->cg_delta_old = c->cg_delta;
  double dl = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dl)
  for (int i = 0; i < NL; i++) {
    c->vel_guess[i] = add(c->vel_guess[i], mul(c->cg_dir[i], alpha));
    c->cg_r[i] = sub(c->cg_r[i], mul(c->cg_Ad[i], alpha));
    c->cg_s[i] = matvec(c->cg_Minv + (long)i * 9, c->cg_r[i]);
    dl += (double)dot(c->cg_r[i], c->cg_s[i]);
  }
  c->cg_delta = (float)dl;
  float beta = c->cg_delta / c->cg_delta_old;
  PFOR for (int i = 0; i < NL; i++) c->cg_dir[i] = add(c->cg_s[i], mul(c->cg_dir[i], beta));
}
/* ref: iisph.py:270-274 */
void orc_iisph_combine_nonpressure(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    c->d_vel[i] = add(c->gravity, mul(sub(c->vel_guess[i], c->vel[i]), 1.0f / c->deltaT));
    c->vel_guess[i] = sub(c->vel_guess[i], c->vel[i]);
  }
}
/* ref: iisph.py:114-127 */
void orc_iisph_compute_nonpressure_force(orc_ctx *c) {
  orc_iisph_init_viscosity_para(c);
  c->vs_iter = 0;
  while (c->vs_iter < 100) {
    orc_iisph_compute_viscosity_force(c);
    c->vs_iter++;
    if ((double)c->cg_delta <= (double)c->viscosity_err * (double)c->cg_delta_zero || (double)c->cg_delta_zero < (double)c->eps) break;
  }
  orc_iisph_combine_nonpressure(c);
}
/* ref: iisph.py:276-316 */
void orc_iisph_compute_advection(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    v3 dii = V(0, 0, 0);
    c->vel[i] = add(c->vel[i], mul(c->d_vel[i], c->deltaT));
    float inv_den = c->ss_rho0 / c->rho[i];
    NB_BEGIN(i)
      dii = add(dii, mul(ss_gradW(c, r), -c->ss_VL0 * inv_den * inv_den));
    NB_END
    c->d_ii[i] = dii;
  }
  PFOR for (int i = 0; i < NL; i++) {
    float aii = 0.0f, density = c->rho[i] / c->ss_rho0, adv = density;
    c->pressure_pre[i] = 0.5f * c->pressure[i];
    NB_BEGIN(i)
      v3 g = ss_gradW(c, r);
      if (j < NL) adv += c->deltaT * c->ss_VL0 * dot(sub(c->vel[i], c->vel[j]), g);
      else        adv += c->deltaT * c->ss_VS0 * dot(c->vel[i], g);
      v3 d_ji = mul(g, c->ss_VL0 / (density * density));
      aii += c->ss_VL0 * dot(sub(c->d_ii[i], d_ji), g);
    NB_END
    c->a_ii[i] = aii; c->adv_rho[i] = adv;
  }
}
/* ref: iisph.py:318-334 */
void orc_iisph_update_iter_info(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    v3 s = V(0, 0, 0);
    NB_BEGIN(i)
      if (j < NL) {
        float dj = c->rho[j] / c->ss_rho0;
        s = add(s, mul(ss_gradW(c, r), -c->ss_VL0 / (dj * dj) * c->pressure_pre[j]));
      }
    NB_END
    c->dij_pj[i] = s;
  }
  c->avg_density_err = 0.0f;
}
/* ref: iisph.py:336-370 (pressure_pre is never refreshed inside the solver loop: every iteration is the same) */
void orc_iisph_update_pressure_force(orc_ctx *c) {
  double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err)
  for (int i = 0; i < NL; i++) {
    float sum = 0.0f;
    NB_BEGIN(i)
      v3 g = ss_gradW(c, r);
      if (j < NL) {
        float density = c->rho[i] / c->ss_rho0;
        v3 dji = mul(g, c->ss_VL0 / (density * density));
        v3 d_ji_pi = mul(dji, c->pressure_pre[i]);
        v3 t = sub(sub(c->dij_pj[i], mul(c->d_ii[j], c->pressure_pre[j])), sub(c->dij_pj[j], d_ji_pi));
        sum += c->ss_VL0 * dot(t, g);
      } else sum += c->ss_VS0 * dot(c->dij_pj[i], g);
    NB_END
    float b = 1.0f - c->adv_rho[i], h2 = c->deltaT * c->deltaT, denom = c->a_ii[i] * h2, p;
    if (fabsf(denom) > c->eps) p = fmaxf((1.0f - c->ii_omega) * c->pressure_pre[i] + c->ii_omega / denom * (b - h2 * sum), 0.0f);
    else p = 0.0f;
    c->pressure[i] = p;
    if (p != 0.0f && i < c->n_reduce) err += (double)((c->a_ii[i] * p + sum) * h2 - b);
  }
  c->avg_density_err = (float)err;
}
/* ref: iisph.py:130-139 */
void orc_iisph_solve_pressure(orc_ctx *c) {
  c->pr_iter = 0;
  double err = 0.0;
  while ((err > 0.001 || c->pr_iter < 2) && c->pr_iter < 100) {
    orc_iisph_update_iter_info(c);
    orc_iisph_update_pressure_force(c);
    err = (double)c->avg_density_err / (double)c->n_reduce;
    c->pr_iter++;
  }
}
/* ref: iisph.py:372-396 */
void orc_iisph_update_pos(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    v3 a = V(0, 0, 0);
    float di = c->rho[i] / c->ss_rho0, dpi = c->pressure[i] / (di * di);
    NB_BEGIN(i)
      v3 g = ss_gradW(c, r);
      if (j < NL) { float dj = c->rho[j] / c->ss_rho0; a = add(a, mul(g, -c->ss_VL0 * (dpi + c->pressure[j] / (dj * dj)))); }
      else a = add(a, mul(g, -c->ss_VS0 * dpi));
    NB_END
    c->d_vel[i] = a;
  }
  PFOR for (int i = 0; i < NL; i++) {
    c->vel[i] = add(c->vel[i], mul(c->d_vel[i], c->deltaT));
    c->pos[i] = add(c->pos[i], mul(c->vel[i], c->deltaT));
  }
}
/* ref: iisph.py:419-428 main-loop body */
void orc_iisph_step(orc_ctx *c) {
  orc_update_grid(c);
  orc_iisph_compute_density(c);
  orc_iisph_compute_nonpressure_force(c);
  orc_iisph_compute_advection(c);
  orc_iisph_solve_pressure(c);
  orc_iisph_update_pos(c);
}


/* ------------------------------------------------------------------------------------ */
/* ParticleData surface-reconstruction kernels (ref: ParticleData.py:187-298), SURVEY 8(f) rank 3  */

/* ref: ParticleData.py:187-215 */
void orc_compute_color_map(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float col = c->liqiudMass / c->rho[i] * Cubic_W_norm(c, 0.0f);
    NB_BEGIN(i)
      float Wr = Cubic_W(c, r);
      if (j < NL) col += c->liqiudMass / c->rho[j] * Wr;
      else        col += c->VS0 * Wr;
    NB_END
    c->color[i] = col;
  }
  PFOR for (int i = 0; i < NL; i++) {
    v3 g = V(0, 0, 0);
    NB_BEGIN(i)
      if (j < NL) g = add(g, mul(CubicGradW(c, r), c->liqiudMass / c->rho[j] * c->color[j]));
    NB_END
    c->color_grad[i] = mul(g, 1.0f / c->color[i]);
  }
}
/* ref: ParticleData.py:291-298 */
static inline float aniso_weight(const orc_ctx *c, v3 xi, v3 xj) {
  float dis = sqrtf(nsq(sub(xi, xj))), R2 = c->mc_searchR * 2.0f;
  return (dis < R2) ? 1.0f - powf(dis / R2, 3.0f) : 0.0f;
}
/* eigen-decomposition of a symmetric 3x3 matrix by cyclic Jacobi rotations: a = Q diag(w) Q^T, w descending.
 * The reference calls ti.svd on the (symmetric positive semi-definite) covariance: U = V = Q, sigma = w for every
 * direction with w > 0 -- the only ones whose sign convention matters for G = R inv(sigmas) RT^T. */
static void eig3(const float *a, float *w, float *Q) {
  double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[4], a[5]}, {a[2], a[5], a[8]}};
  double Vv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; sweep++) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-40) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (fabs(A[p][q]) < 1e-300) continue;
      double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
      for (int k = 0; k < 3; k++) { double akp = A[k][p], akq = A[k][q]; A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq; }
      for (int k = 0; k < 3; k++) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk; }
      for (int k = 0; k < 3; k++) { double vkp = Vv[k][p], vkq = Vv[k][q]; Vv[k][p] = cs * vkp - sn * vkq; Vv[k][q] = sn * vkp + cs * vkq; }
    }
  }
  int idx[3] = {0, 1, 2};
  for (int x = 0; x < 2; x++) for (int y = x + 1; y < 3; y++) if (A[idx[y]][idx[y]] > A[idx[x]][idx[x]]) { int t = idx[x]; idx[x] = idx[y]; idx[y] = t; }
  for (int k = 0; k < 3; k++) { w[k] = (float)A[idx[k]][idx[k]]; for (int r = 0; r < 3; r++) Q[r * 3 + k] = (float)Vv[r][idx[k]]; }
}
/* ref: ParticleData.py:220-289 */
void orc_cal_anistropic_kernel(orc_ctx *c) {
  PFOR for (int i = 0; i < NL; i++) {
    float sw = 0.0f; v3 sx = V(0, 0, 0);
    NB_BEGIN(i)
      (void)r;
      if (j < NL) { float w = aniso_weight(c, c->pos[i], c->pos[j]); sw += w; sx = add(sx, mul(c->pos[j], w)); }
    NB_END
    c->pos_avr[i] = (sw > 0.0f) ? mul(sx, 1.0f / sw) : c->pos[i];
  }
  PFOR for (int i = 0; i < NL; i++) {
    const float kr = 4.0f, ks = 1400.0f, kn = 0.5f, ne = 25.0f;
    float *G = c->G + (long)i * 9;
    for (int a = 0; a < 9; a++) G[a] = (a % 4 == 0) ? kn : 0.0f;
    int cur = c->neighborCount[i];
    if ((float)cur > ne) {
      float sw = 0.0f, m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int *nb = c->neighbor + (long)i * c->maxNeighbour;
      for (int k = 0; k < cur; k++) {
        int j = nb[k];
        if (j < NL) {
          float w = aniso_weight(c, c->pos[i], c->pos[j]);
          v3 r = sub(c->pos[j], c->pos_avr[i]);
          sw += w;
          m[0] += w * r.x * r.x; m[1] += w * r.x * r.y; m[2] += w * r.x * r.z;
          m[4] += w * r.y * r.y; m[5] += w * r.y * r.z; m[8] += w * r.z * r.z;
        }
      }
      m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
      for (int a = 0; a < 9; a++) m[a] /= sw;
      float w[3], Q[9];
      eig3(m, w, Q);
      if (w[0] > 0.0f) {
        float s0 = ks * w[0], s1 = ks * fmaxf(w[1], w[0] / kr), s2 = ks * fmaxf(w[2], w[0] / kr);
        float inv[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
          G[a * 3 + b] = Q[a * 3 + 0] * inv[0] * Q[b * 3 + 0] + Q[a * 3 + 1] * inv[1] * Q[b * 3 + 1] + Q[a * 3 + 2] * inv[2] * Q[b * 3 + 2];
      }
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* accessors for the ctypes wrapper (oracle/oracle.py)                                    */
#define PTR(name, type) type *orc_ptr_##name(orc_ctx *c) { return (type *)c->name; }
PTR(pos, float) PTR(vel, float) PTR(vel_guess, float) PTR(d_vel, float) PTR(omega, float) PTR(d_omega, float)
PTR(normal, float) PTR(rho, float) PTR(adv_rho, float) PTR(pressure, float) PTR(vel_max, float)
PTR(cg_Minv, float) PTR(cg_r, float) PTR(cg_dir, float) PTR(cg_Ad, float) PTR(cg_s, float)
PTR(alpha_coff, float) PTR(kappa, float) PTR(kappa_v, float) PTR(kbuf, float)
PTR(color, float) PTR(color_grad, float) PTR(pos_avr, float) PTR(G, float)
PTR(d_vel_pre, float) PTR(pos_star, float) PTR(vel_star, float) PTR(d_ii, float) PTR(dij_pj, float) PTR(a_ii, float) PTR(pressure_pre, float)
PTR(gridCount, int) PTR(grid, int) PTR(neighborCount, int) PTR(neighbor, int)

enum { S_deltaT = 0, S_avg_density_err, S_cg_delta, S_cg_delta_old, S_cg_delta_zero, S_viscosity, S_viscosity_b,
       S_tension_coff, S_tension_coff_b, S_VL0, S_VS0, S_liqiudMass, S_searchR, S_gridR, S_invGridR, S_rho_L0,
       S_viscosity_err, S_vorticity_coff, S_viscosity_omega, S_vorticity_init, S_rho_err, S_pci_coff };
static float *sptr(orc_ctx *c, int id) {
  switch (id) {
    case S_deltaT: return &c->deltaT; case S_avg_density_err: return &c->avg_density_err;
    case S_cg_delta: return &c->cg_delta; case S_cg_delta_old: return &c->cg_delta_old;
    case S_cg_delta_zero: return &c->cg_delta_zero; case S_viscosity: return &c->viscosity;
    case S_viscosity_b: return &c->viscosity_b; case S_tension_coff: return &c->tension_coff;
    case S_tension_coff_b: return &c->tension_coff_b; case S_VL0: return &c->VL0; case S_VS0: return &c->VS0;
    case S_liqiudMass: return &c->liqiudMass; case S_searchR: return &c->searchR; case S_gridR: return &c->gridR;
    case S_invGridR: return &c->invGridR; case S_rho_L0: return &c->rho_L0; case S_viscosity_err: return &c->viscosity_err;
    case S_vorticity_coff: return &c->vorticity_coff; case S_viscosity_omega: return &c->viscosity_omega;
    case S_vorticity_init: return &c->vorticity_init; case S_rho_err: return &c->rho_err; case S_pci_coff: return &c->pci_coff;
  }
  return NULL;
}
float orc_get_scalar(orc_ctx *c, int id) { float *p = sptr(c, id); return p ? *p : NAN; }
void orc_set_scalar(orc_ctx *c, int id, float v) { float *p = sptr(c, id); if (p) *p = v; }
enum { I_vs_iter = 0, I_dv_iter, I_pr_iter, I_overflow_grid, I_overflow_neighbor, I_table_size, I_bs0, I_bs1, I_bs2,
       I_tension_fixed, I_mode, I_n_reduce };
int orc_get_int(orc_ctx *c, int id) {
  switch (id) {
    case I_vs_iter: return c->vs_iter; case I_dv_iter: return c->dv_iter; case I_pr_iter: return c->pr_iter;
    case I_overflow_grid: return c->overflow_grid; case I_overflow_neighbor: return c->overflow_neighbor;
    case I_table_size: return c->table_size; case I_bs0: return c->blockSize[0]; case I_bs1: return c->blockSize[1];
    case I_bs2: return c->blockSize[2]; case I_tension_fixed: return c->tension_fixed; case I_mode: return c->mode; case I_n_reduce: return c->n_reduce;
  }
  return -1;
}
void orc_set_int(orc_ctx *c, int id, int v) {
  switch (id) {
    case I_vs_iter: c->vs_iter = v; break; case I_dv_iter: c->dv_iter = v; break; case I_pr_iter: c->pr_iter = v; break;
    case I_tension_fixed: c->tension_fixed = v; break; case I_n_reduce: c->n_reduce = v; break;
  }
}

double orc_get_dAd(orc_ctx *c) { return c->cg_dAd; }
void orc_set_dAd(orc_ctx *c, double v) { c->cg_dAd = v; }
double orc_get_dnew(orc_ctx *c) { return c->cg_dnew; }
void orc_set_dnew(orc_ctx *c, double v) { c->cg_dnew = v; }

/* thread count of the OpenMP loops (bench.py reports an all-cores and a 1-thread figure) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
