// mcgrid.h -- surface reconstruction on a dense grid (grid half):
//   scalar field  ref: MarchingCubeGrid.py:159-196 (cal_surface_point, isotropic)
//   marching cubes ref: MarchingCubeGrid.py:232-328 (marching_cube), :373-409 (vertex_interp / check_pos)
// The reference bins the particles a second time on the MC grid (cell 0.9 R, 9^3-cell walk per node) and reads its case
// tables from MCData.txt.  Here the node sum walks the solver's own cell-sorted particles (cell 2 R, 5^3 cells cover
// the 3.6 R support), and the case tables are GENERATED (mc_build_tables): for each of the 256 corner states the
// iso-polygon loops are traced face by face -- on every cube face each run of flagged corners is cut off by one
// directed segment, the segments of the six faces chain into closed loops, each loop is fanned into triangles.  The
// rule depends only on a face's corner states, so two cubes sharing a face agree on it and the mesh is watertight;
// it is a valid marching-cubes table but not the published one, so triangle connectivity inside a cube can differ
// from the reference's while the vertices (edge crossings) are the same.
#pragma once
#include <vector>

#include "device.h"

#define MC_TRI_ROW 16   /* up to 5 triangles + terminator, as the reference's tritable (MarchingCubeGrid.py:43) */

struct McDev {
  int nx, ny, nz;            // nodes per axis (MarchingCubeGrid.py:67-69)
  float minx, miny, minz;    // node (0,0,0)
  float grid_r;              // node pitch
  float h, inv_h, mk_h3;     // Cubic kernel of the MC grid (searchR, MarchingCubeGrid.py:29)
  float rho_min;             // liqiudMass * W_norm(0)  (:178)
  float isolevel;
  float *value;              // [nx*ny*nz]
  float *tri;                // [cap][3]
  int *vertex_count;         // [1]
  int cap;
  const int *edge_table;     // [256]
  const int *tri_table;      // [256][MC_TRI_ROW]
};

// ---- case tables -------------------------------------------------------------------------------------------------
// corner c of a cube: bit0 = +x, bit... in the reference's order (index0..index7, MarchingCubeGrid.py:240-247)
static const int MC_CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge e joins corners MC_EDGE[e][0..1] (the vertlist slots of MarchingCubeGrid.py:276-299)
static const int MC_EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

inline int mc_edge_between(int a, int b) {
  for (int e = 0; e < 12; e++)
    if ((MC_EDGE[e][0] == a && MC_EDGE[e][1] == b) || (MC_EDGE[e][0] == b && MC_EDGE[e][1] == a)) return e;
  return -1;
}

// edge[256], tri[256 * MC_TRI_ROW] (-1 terminated rows).  Returns the largest triangle count of a case.
inline int mc_build_tables(int *edge, int *tri) {
  // the six faces, corners counter-clockwise seen from OUTSIDE the cube (checked below)
  int face[6][4] = {{0, 3, 2, 1}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 7, 6, 2}, {0, 4, 7, 3}, {1, 2, 6, 5}};
  for (int f = 0; f < 6; f++) {   // make the orientation a checked property instead of a typed-in one
    const int *a = MC_CORNER[face[f][0]], *b = MC_CORNER[face[f][1]], *c = MC_CORNER[face[f][2]];
    const int u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, v[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    const int n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    double ctr[3] = {0, 0, 0};
    for (int k = 0; k < 4; k++) for (int q = 0; q < 3; q++) ctr[q] += MC_CORNER[face[f][k]][q] * 0.25;
    const double out = n[0] * (ctr[0] - 0.5) + n[1] * (ctr[1] - 0.5) + n[2] * (ctr[2] - 0.5);
    if (out < 0) { int t = face[f][1]; face[f][1] = face[f][3]; face[f][3] = t; }
  }
  int worst = 0;
  for (int ci = 0; ci < 256; ci++) {
    int em = 0;
    for (int e = 0; e < 12; e++)
      if (((ci >> MC_EDGE[e][0]) & 1) != ((ci >> MC_EDGE[e][1]) & 1)) em |= 1 << e;
    edge[ci] = em;   // MarchingCubeGrid.py:274-299 reads exactly this: the edges whose ends differ
    int next[12];
    for (int e = 0; e < 12; e++) next[e] = -1;
    for (int f = 0; f < 6; f++) {
      const int *c = face[f];
      int flagged = 0;
      for (int k = 0; k < 4; k++) flagged += (ci >> c[k]) & 1;
      if (flagged == 0 || flagged == 4) continue;
      // every maximal run of flagged corners (cyclic) is cut off by one segment, flagged side on its left:
      // from the crossing after the run (flagged -> clear) to the crossing before it (clear -> flagged)
      for (int k = 0; k < 4; k++) {
        const int cur = (ci >> c[k]) & 1, prev = (ci >> c[(k + 3) & 3]) & 1;
        if (!(cur && !prev)) continue;   // k starts a run
        int end = k;
        while ((ci >> c[(end + 1) & 3]) & 1) end = (end + 1) & 3;
        const int e_after = mc_edge_between(c[end], c[(end + 1) & 3]), e_before = mc_edge_between(c[(k + 3) & 3], c[k]);
        next[e_after] = e_before;
      }
    }
    int *row = tri + ci * MC_TRI_ROW, w = 0, ntri = 0;
    bool used[12] = {false};
    for (int e0 = 0; e0 < 12; e0++) {
      if (!((em >> e0) & 1) || used[e0]) continue;
      int loop[12], n = 0;
      for (int e = e0; !used[e]; e = next[e]) { used[e] = true; loop[n++] = e; }
      for (int k = 1; k + 1 < n; k++) {   // fan; the loop direction makes the normal point to the flagged (value < iso) side
        if (w + 3 < MC_TRI_ROW) { row[w] = loop[0]; row[w + 1] = loop[k]; row[w + 2] = loop[k + 1]; }
        w += 3; ntri++;
      }
    }
    for (int k = (w < MC_TRI_ROW ? w : MC_TRI_ROW - 1); k < MC_TRI_ROW; k++) row[k] = -1;
    if (ntri > worst) worst = ntri;
  }
  return worst;
}

// ---- device ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float3 mc_node_pos(const McDev &m, int x, int y, int z) {   // MarchingCubeGrid.py:356-358
  return f3(m.minx + (float)x * m.grid_r, m.miny + (float)y * m.grid_r, m.minz + (float)z * m.grid_r);
}
__device__ __forceinline__ float mc_cubic_W(const McDev &m, float r2) {   // kernels/CubicKernel.py:8-18 with h = searchR
  float q = sqrtf(r2) * m.inv_h;
  float f = 1.0f - q;
  float w = (q <= 0.5f) ? (6.0f * q * q * q - 6.0f * q * q + 1.0f) : (2.0f * f * f * f);
  return (q <= 1.0f) ? w * m.mk_h3 : 0.0f;
}

// ref: MarchingCubeGrid.py:159-184 -- value(node) = sum over liquid j with W > 0 and rho_j > rho_min of m / rho_j * W
__global__ void __launch_bounds__(BLOCK) k_mc_value(Dev d, McDev m) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)m.nx * m.ny * m.nz;
  if (t >= n) return;
  const int z = (int)(t % m.nz), y = (int)((t / m.nz) % m.ny), x = (int)(t / ((long)m.nz * m.ny));
  const float3 p = mc_node_pos(m, x, y, z);
  // cell of the node in the solver's grid (may lie outside it: the MC grid reaches searchR beyond the bounding box)
  const int cx = (int)floorf((p.x - d.minx) * d.inv_grid) - d.x0, cy = (int)floorf((p.y - d.miny) * d.inv_grid),
            cz = (int)floorf((p.z - d.minz) * d.inv_grid);
  // only cells inside the occupied box of this step hold particles -- and only the cell table of the tiles around it
  // is current (the grid update scans nothing else, grid.h scan_block_active)
  const int bx0 = d.bbox[0], by0 = d.bbox[1], bz0 = d.bbox[2], bx1 = d.bbox[3], by1 = d.bbox[4], bz1 = d.bbox[5];
  const int z0 = max(cz - 2, bz0), z1 = min(cz + 2, bz1) + 1;
  float s = 0.f;
  if (z1 > z0)
    for (int xx = max(cx - 2, bx0); xx <= min(cx + 2, bx1); xx++)
      for (int yy = max(cy - 2, by0); yy <= min(cy + 2, by1); yy++) {
        const RowPieces rp = row_pieces(d, d.cs_l, xx, yy, z0, z1);
        for (int piece = 0; piece < 2; piece++)
          for (int j = piece ? rp.lo1 : rp.lo0; j < (piece ? rp.hi1 : rp.hi0); j++) {
            const float3 r = p - xyz(d.pos[j]);
            const float w = mc_cubic_W(m, dot3(r, r));
            if (w > 0.0f) {
              const float rho = d.rho[j];
              if (rho > m.rho_min) s += d.mass / rho * w;
            }
          }
      }
  m.value[t] = s;
}

// ref: MarchingCubeGrid.py:373-409 -- the interpolation always starts from the endpoint with the smaller coordinate
__device__ __forceinline__ float3 mc_vertex_interp(const McDev &m, float3 p1, float3 p2, float v1, float v2) {
  const bool swap = (p2.x < p1.x) || (p2.y < p1.y) || (p2.z < p1.z);   // the ends of a cube edge differ in one coordinate
  if (swap) { float3 tp = p1; p1 = p2; p2 = tp; float tv = v1; v1 = v2; v2 = tv; }
  if (fabsf(v1 - v2) > 0.00001f) {
    const float t = (m.isolevel - v1) / (v2 - v1);
    return f3(p1.x + (p2.x - p1.x) * t, p1.y + (p2.y - p1.y) * t, p1.z + (p2.z - p1.z) * t);
  }
  return p1;
}

// ref: MarchingCubeGrid.py:232-328
__global__ void __launch_bounds__(BLOCK) k_mc_march(McDev m) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)m.nx * m.ny * m.nz;
  if (t >= n) return;
  const int z = (int)(t % m.nz), y = (int)((t / m.nz) % m.ny), x = (int)(t / ((long)m.nz * m.ny));
  if (x + 1 >= m.nx || y + 1 >= m.ny || z + 1 >= m.nz) return;
  float v[8];
  int ci = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const int cx = x + ((0x66 >> c) & 1), cy = y + ((0xCC >> c) & 1), cz = z + ((0xF0 >> c) & 1);   // MC_CORNER as bit masks
    v[c] = m.value[((long)cx * m.ny + cy) * m.nz + cz];
    if (v[c] < m.isolevel) ci |= 1 << c;
  }
  const int em = m.edge_table[ci];
  if (em == 0) return;
  const int *row = m.tri_table + ci * MC_TRI_ROW;
  for (int k = 0; k < MC_TRI_ROW - 1 && row[k] >= 0; k += 3) {
    const int old = atomicAdd(m.vertex_count, 3);
    if (old + 3 > m.cap) continue;   // counted, not stored (the reference prints "exceed max tri")
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int e = row[k + q];
      // corners of edge e (MC_EDGE) as nibble tables
      const int a = (int)((0x321076543210ULL >> (4 * e)) & 7), b = (int)((0x765447650321ULL >> (4 * e)) & 7);
      const float3 pa = mc_node_pos(m, x + ((0x66 >> a) & 1), y + ((0xCC >> a) & 1), z + ((0xF0 >> a) & 1));
      const float3 pb = mc_node_pos(m, x + ((0x66 >> b) & 1), y + ((0xCC >> b) & 1), z + ((0xF0 >> b) & 1));
      float va = 0.f, vb = 0.f;
#pragma unroll
      for (int c = 0; c < 8; c++) { if (c == a) va = v[c]; if (c == b) vb = v[c]; }
      const float3 p = mc_vertex_interp(m, pa, pb, va, vb);
      float *o = m.tri + 3 * (size_t)(old + q);
      o[0] = p.x; o[1] = p.y; o[2] = p.z;
    }
  }
}
