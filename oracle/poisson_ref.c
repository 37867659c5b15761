/*
 * oracle/poisson_ref.c -- CPU restatement of the reference's boundary sampler (ref: boundry.py:100-460).
 * TEST INFRASTRUCTURE ONLY (see wcsph_ref.c header); PARITY UNPINNED: the reference tool draws from ti.random and its
 * result differs from run to run.  What is restated is the ALGORITHM -- candidate generation, cell ordering, phase
 * groups, trial rounds, normal-corrected distance test -- on a specified counter-based random stream, so that the GPU
 * sampler can be compared point by point:
 *   u(i,k) = (mix(seed + i*747796405 + k*2891336453 + 0x9E3779B9) >> 8) / 2^24,  mix = the lowbias32 finaliser.
 * Candidates of one cell are tried in index order.  Serial, deterministic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static float u01(uint32_t seed, uint32_t i, uint32_t k) { return (float)(mix(seed + i * 747796405u + k * 2891336453u + 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f); }

static float dist_metric(const float *p, const float *q, int tp, int tq, const float *nrm) { /* boundry.py:340-370 */
  float v[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
  float dist = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (tp != tq && dist > 0.0f) {
    float inv = 1.0f / dist;
    float c1 = (nrm[3 * tp] * v[0] + nrm[3 * tp + 1] * v[1] + nrm[3 * tp + 2] * v[2]) * inv;
    float c2 = (nrm[3 * tq] * v[0] + nrm[3 * tq + 1] * v[1] + nrm[3 * tq + 2] * v[2]) * inv;
    if (fabsf(c1 - c2) > 0.00001f) dist *= (asinf(fminf(fmaxf(c1, -1.f), 1.f)) - asinf(fminf(fmaxf(c2, -1.f), 1.f))) / (c1 - c2);
    else dist /= sqrtf(fmaxf(1.0f - c1 * c1, 1e-12f));
  }
  return dist;
}
float orc_ps_distance(const float *p, const float *q, int tp, int tq, const float *nrm) { return dist_metric(p, q, tp, tq, nrm); }

/* returns the number of samples (written to out up to cap); tri: [n_tri][9] */
int orc_poisson_sample(const float *tri, int n_tri, double R, uint32_t seed, int trials, float *out, int cap) {
  float *nrm = (float *)malloc((size_t)n_tri * 12), *area = (float *)malloc((size_t)n_tri * 4);
  double total = 0.0, maxa = 0.0;
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int t = 0; t < n_tri; t++) { /* boundry.py:128-161 */
    const float *a = tri + 9 * (size_t)t, *b = a + 3, *c = a + 6;
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], fminf(a[k], fminf(b[k], c[k]))); mx[k] = fmaxf(mx[k], fmaxf(a[k], fmaxf(b[k], c[k]))); }
    double d1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, d2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
    double n[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
    double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    area[t] = (float)(len * 0.5);
    for (int k = 0; k < 3; k++) nrm[3 * (size_t)t + k] = len > 0 ? (float)(n[k] / len) : 0.f;
    total += (double)area[t];
    if ((double)area[t] > maxa) maxa = (double)area[t];
  }
  int N = (int)(40.0 * (total / (3.1415926 * R * R)));   /* boundry.py:163-164 */
  double grid_r = R / sqrt(3.0);
  float inv = (float)(1.0 / grid_r), max_area = (float)maxa, Rf = (float)R;
  int nx = (int)(((double)mx[0] - mn[0]) / grid_r) + 3, ny = (int)(((double)mx[1] - mn[1]) / grid_r) + 3, nz = (int)(((double)mx[2] - mn[2]) / grid_r) + 3;
  long ncell = (long)nx * ny * nz;
  float *pos = (float *)malloc((size_t)N * 12);
  int *tid = (int *)malloc((size_t)N * 4), *cell = (int *)malloc((size_t)N * 4), *order = (int *)malloc((size_t)N * 4);
  int *start = (int *)calloc((size_t)ncell + 1, 4), *fill = (int *)calloc((size_t)ncell + 1, 4);
  int *scnt = (int *)calloc((size_t)ncell, 4), *samp = (int *)malloc((size_t)ncell * 5 * 4);
  char *acc = (char *)calloc((size_t)N, 1);
  for (int i = 0; i < N; i++) { /* boundry.py:222-246 */
    volatile float rn1 = sqrtf(u01(seed, i, 0));
    volatile float bc1 = 1.0f - rn1;
    volatile float bc2 = u01(seed, i, 1) * rn1;
    volatile float t1 = 1.0f - bc1;
    volatile float bc3 = t1 - bc2;
    int t = 0;
    for (uint32_t k = 2;; k += 2) {
      volatile float f = (float)n_tri * u01(seed, i, k);
      t = (int)f;
      if (t >= n_tri) t = n_tri - 1;
      if (u01(seed, i, k + 1) < area[t] / max_area || k > 4000) break;
    }
    const float *a = tri + 9 * (size_t)t;
    for (int c = 0; c < 3; c++) {
      volatile float x = bc1 * a[c], y = bc2 * a[3 + c], z = bc3 * a[6 + c];
      volatile float s = x + y;
      pos[3 * (size_t)i + c] = s + z;
    }
    tid[i] = t;
    volatile float fx = (pos[3 * (size_t)i] - mn[0]), fy = (pos[3 * (size_t)i + 1] - mn[1]), fz = (pos[3 * (size_t)i + 2] - mn[2]);
    volatile float gx = fx * inv, gy = fy * inv, gz = fz * inv;
    int cx = (int)gx + 1, cy = (int)gy + 1, cz = (int)gz + 1;
    cell[i] = (cx * ny + cy) * nz + cz;
    start[cell[i] + 1]++;
  }
  for (long c = 0; c < ncell; c++) start[c + 1] += start[c];
  for (int i = 0; i < N; i++) order[start[cell[i]] + fill[cell[i]]++] = i;   /* index order inside a cell */
  for (int trial = 0; trial < trials; trial++)          /* boundry.py:424-458 */
    for (int pg = 0; pg < 27; pg++)
      for (long c = 0; c < ncell; c++) {
        int cz = (int)(c % nz), cy = (int)((c / nz) % ny), cx = (int)(c / ((long)nz * ny));
        if (cx % 3 + 3 * (cy % 3) + 9 * (cz % 3) != pg) continue;
        int slot = start[c] + trial;
        if (slot >= start[c + 1]) continue;
        int cand = order[slot], ok = 1;
        for (int dx = -2; dx <= 2 && ok; dx++) for (int dy = -2; dy <= 2 && ok; dy++) for (int dz = -2; dz <= 2 && ok; dz++) {
          int x = cx + dx, y = cy + dy, z = cz + dz;
          if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) continue;
          long n = ((long)x * ny + y) * nz + z;
          for (int q = 0; q < scnt[n]; q++) {
            int o = samp[n * 5 + q];
            if (dist_metric(pos + 3 * (size_t)cand, pos + 3 * (size_t)o, tid[cand], tid[o], nrm) < Rf) { ok = 0; break; }
          }
        }
        if (ok && scnt[c] < 5) { samp[c * 5 + scnt[c]++] = cand; acc[cand] = 1; }
      }
  int n = 0;
  for (int i = 0; i < N; i++) if (acc[i]) { if (n < cap) memcpy(out + 3 * (size_t)n, pos + 3 * (size_t)i, 12); n++; }
  free(nrm); free(area); free(pos); free(tid); free(cell); free(order); free(start); free(fill); free(scnt); free(samp); free(acc);
  return n;
}
