/*
 * oracle/mc_ref.c -- CPU restatement of the reference's surface grid (ref: MarchingCubeGrid.py), TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (the Taichi reference cannot run here and ships no golden meshes).
 *   orc_mc_value      cal_surface_point, MarchingCubeGrid.py:159-184, particle-to-node scatter over the +-4 nodes the
 *                     reference's 9^3 walk reaches (a different summation order than the GPU's node gather)
 *   orc_mc_crossings  the vertices marching_cube can emit, table-free: every grid edge inside a processed cube whose
 *                     end values straddle the isolevel, interpolated as vertex_interp does (:373-409)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float cubic_w(float r, float h, float mk_h3) {   /* kernels/CubicKernel.py:36-54 */
  float q = r / h, res = 0.0f;
  if (q <= 1.0f) {
    if (q <= 0.5f) { float qq = q * q; res = 6.0f * qq * q - 6.0f * qq + 1.0f; }
    else { float f = 1.0f - q; res = 2.0f * f * f * f; }
  }
  return res * mk_h3;
}

/* pos[n][3], rho[n] of the LIQUID particles; value[nx*ny*nz] (x slowest) */
void orc_mc_value(const float *pos, const float *rho, int n, float mass, double grid_r, double search_r,
                  const float *minb, const int *dims, float *value) {
  const int nx = dims[0], ny = dims[1], nz = dims[2];
  const float gr = (float)grid_r, h = (float)search_r, inv = (float)(1.0 / grid_r);
  const float mk_h3 = (float)((8.0 / 3.14159265358979323846) / (search_r * search_r * search_r));
  const float rho_min = mass * cubic_w(0.0f, h, mk_h3);
  for (long t = 0; t < (long)nx * ny * nz; t++) value[t] = 0.0f;
  for (int j = 0; j < n; j++) {
    if (!(rho[j] > rho_min)) continue;
    const float *p = pos + 3 * (size_t)j;
    /* MarchingCubeGrid.py:147: cell of the particle on the MC grid (truncation), :148 particles outside are not binned */
    int c[3];
    int inbox = 1;
    for (int a = 0; a < 3; a++) { c[a] = (int)((p[a] - minb[a]) * inv); if (c[a] < 0 || c[a] >= dims[a]) inbox = 0; }
    if (!inbox) continue;
    for (int x = c[0] - 4; x <= c[0] + 4; x++) for (int y = c[1] - 4; y <= c[1] + 4; y++) for (int z = c[2] - 4; z <= c[2] + 4; z++) {
      if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) continue;
      float rx = (minb[0] + (float)x * gr) - p[0], ry = (minb[1] + (float)y * gr) - p[1], rz = (minb[2] + (float)z * gr) - p[2];
      float w = cubic_w(sqrtf(rx * rx + ry * ry + rz * rz), h, mk_h3);
      if (w > 0.0f) value[((long)x * ny + y) * nz + z] += mass / rho[j] * w;
    }
  }
}

/* out[cap][3]; returns the number of crossings (one per straddling grid edge that belongs to at least one cube) */
long orc_mc_crossings(const float *value, const int *dims, const float *minb, double grid_r, float iso, float *out, long cap) {
  const int nx = dims[0], ny = dims[1], nz = dims[2];
  const float gr = (float)grid_r;
  long n = 0;
  for (int x = 0; x < nx; x++) for (int y = 0; y < ny; y++) for (int z = 0; z < nz; z++)
    for (int a = 0; a < 3; a++) {
      int x2 = x + (a == 0), y2 = y + (a == 1), z2 = z + (a == 2);
      if (x2 >= nx || y2 >= ny || z2 >= nz) continue;
      float v1 = value[((long)x * ny + y) * nz + z], v2 = value[((long)x2 * ny + y2) * nz + z2];
      if ((v1 < iso) == (v2 < iso)) continue;
      float p1[3] = {minb[0] + (float)x * gr, minb[1] + (float)y * gr, minb[2] + (float)z * gr};
      float p2[3] = {minb[0] + (float)x2 * gr, minb[1] + (float)y2 * gr, minb[2] + (float)z2 * gr};
      float p[3] = {p1[0], p1[1], p1[2]};   /* p1 is the lower end: no swap needed (check_pos) */
      if (fabsf(v1 - v2) > 0.00001f) for (int q = 0; q < 3; q++) p[q] = p1[q] + (p2[q] - p1[q]) / (v2 - v1) * (iso - v1);
      if (n < cap) { out[3 * n] = p[0]; out[3 * n + 1] = p[1]; out[3 * n + 2] = p[2]; }
      n++;
    }
  return n;
}

/* marching_cube, MarchingCubeGrid.py:252-328, with the case tables passed in (edge_table[256], tri_table[256][16]):
 * out[cap][3] vertices, three per triangle, cubes in index order (the reference appends through an atomic counter, so its
 * triangle ORDER is schedule dependent; the set is not).  Returns the vertex count. */
static void mc_interp(const float *p1, const float *p2, float v1, float v2, float iso, float *p) {   /* :373-409 */
  int swap = 1;   /* check_pos(p2, p1): the LAST differing coordinate decides */
  for (int a = 0; a < 3; a++) { if (p2[a] < p1[a]) swap = 1; else if (p2[a] > p1[a]) swap = 0; }
  if (swap) { const float *tp = p1; p1 = p2; p2 = tp; float tv = v1; v1 = v2; v2 = tv; }
  for (int a = 0; a < 3; a++) p[a] = p1[a];
  if (fabsf(v1 - v2) > 0.00001f) for (int a = 0; a < 3; a++) p[a] = p1[a] + (p2[a] - p1[a]) / (v2 - v1) * (iso - v1);
}
long orc_mc_triangles(const float *value, const int *dims, const float *minb, double grid_r, float iso,
                      const int *edge_table, const int *tri_table, float *out, long cap) {
  static const int CO[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};   /* :257-264 */
  static const int ED[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  const int nx = dims[0], ny = dims[1], nz = dims[2];
  const float gr = (float)grid_r;
  long n = 0;
  for (int x = 0; x + 1 < nx; x++) for (int y = 0; y + 1 < ny; y++) for (int z = 0; z + 1 < nz; z++) {
    float v[8], p[8][3];
    int ci = 0;
    for (int c = 0; c < 8; c++) {
      const int cx = x + CO[c][0], cy = y + CO[c][1], cz = z + CO[c][2];
      v[c] = value[((long)cx * ny + cy) * nz + cz];
      p[c][0] = minb[0] + (float)cx * gr; p[c][1] = minb[1] + (float)cy * gr; p[c][2] = minb[2] + (float)cz * gr;
      if (v[c] < iso) ci |= 1 << c;
    }
    if (edge_table[ci] == 0) continue;
    float vl[12][3];
    for (int e = 0; e < 12; e++) {
      for (int a = 0; a < 3; a++) vl[e][a] = p[0][a];
      if (edge_table[ci] & (1 << e)) mc_interp(p[ED[e][0]], p[ED[e][1]], v[ED[e][0]], v[ED[e][1]], iso, vl[e]);
    }
    for (int k = 0; k + 2 < 16 && tri_table[ci * 16 + k] != -1; k += 3)
      for (int q = 0; q < 3; q++) {
        if (n < cap) for (int a = 0; a < 3; a++) out[3 * n + a] = vl[tri_table[ci * 16 + k + q]][a];
        n++;
      }
  }
  return n;
}
