// canvas.h -- headless frame dump: the particle rasteriser of the reference's viewer
//   ref: Canvas.py:138-142 transform, :144-149 fill_pixel, :151-179 draw_sphere (midpoint circle, r = 3),
//        :201-205 draw_point, :207-211 clear_canvas; dfsph.py:585-593 draw_particle (liquid: white circles, every
//        particle: a grey point)
// The reference's depth test is a non-atomic compare-and-write (a data race between particles landing on one pixel);
// here a pixel is one 64-bit word {orderable depth, colour} resolved with atomicMin: the nearest fragment wins.
#pragma once
#include "device.h"

struct CanvasDev {
  float vp[16];     // proj * view, row-major
  int sx, sy;
  unsigned long long *pix;   // [sx][sy]
};

__device__ __forceinline__ unsigned cv_order(float z) {   // monotone float -> uint
  unsigned u = __float_as_uint(z);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned cv_pack(float r, float g, float b) {
  auto q = [](float v) { return (unsigned)(fminf(fmaxf(v, 0.0f), 1.0f) * 1023.0f + 0.5f); };
  return (q(r) << 20) | (q(g) << 10) | q(b);
}
__device__ __forceinline__ void cv_fill(const CanvasDev &c, int x, int y, float z, unsigned col) {   // Canvas.py:144-149
  if (x < 0 || x >= c.sx || y < 0 || y >= c.sy || !(z < 1.0f)) return;   // depth starts at 1.0, `depth > z`
  atomicMin(&c.pix[(size_t)x * c.sy + y], ((unsigned long long)cv_order(z) << 32) | col);
}
__device__ __forceinline__ bool cv_transform(const CanvasDev &c, float3 v, float *px, float *py, float *pz) {   // Canvas.py:138-142
  const float *m = c.vp;
  const float x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
  const float z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11], w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15];
  if (w == 0.0f) return false;
  *px = (x / w + 1.0f) * 0.5f * (float)c.sx; *py = (y / w + 1.0f) * 0.5f * (float)c.sy; *pz = z / w;
  return fabsf(*px) < 1e9f && fabsf(*py) < 1e9f;
}

__global__ void k_canvas_clear(CanvasDev c) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)c.sx * c.sy) c.pix[t] = ((unsigned long long)cv_order(1.0f) << 32);   // black, depth 1.0
}
// liquid particle i (sorted order; the image does not depend on the order): circle outline + centre point
__global__ void k_canvas_liquid(Dev d, CanvasDev c, unsigned col_sphere, unsigned col_point) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float fx, fy, z;
  if (!cv_transform(c, xyz(d.pos[i]), &fx, &fy, &z)) return;
  const int xc = (int)fx, yc = (int)fy;
  int x = 0, y = 3, dd = 3 - 2 * 3;   // Canvas.py:157-179
  while (x <= y) {
    cv_fill(c, xc + x, yc + y, z, col_sphere); cv_fill(c, xc - x, yc + y, z, col_sphere);
    cv_fill(c, xc + x, yc - y, z, col_sphere); cv_fill(c, xc - x, yc - y, z, col_sphere);
    cv_fill(c, xc + y, yc + x, z, col_sphere); cv_fill(c, xc - y, yc + x, z, col_sphere);
    cv_fill(c, xc + y, yc - x, z, col_sphere); cv_fill(c, xc - y, yc - x, z, col_sphere);
    if (dd < 0) dd = dd + 4 * x + 6;
    else { dd = dd + 4 * (x - y) + 10; y--; }
    x++;
  }
  cv_fill(c, xc, yc, z, col_point);
}
__global__ void k_canvas_solid(Dev d, CanvasDev c, unsigned col_point) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.ns) return;
  float fx, fy, z;
  if (cv_transform(c, xyz(d.spos[i]), &fx, &fy, &z)) cv_fill(c, (int)fx, (int)fy, z, col_point);
}
__global__ void k_canvas_resolve(CanvasDev c, float *img) {   // -> [sx][sy][3] float, like Canvas.img
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)c.sx * c.sy) return;
  const unsigned col = (unsigned)(c.pix[t] & 0xffffffffu);
  img[3 * t] = (float)((col >> 20) & 1023) / 1023.0f;
  img[3 * t + 1] = (float)((col >> 10) & 1023) / 1023.0f;
  img[3 * t + 2] = (float)(col & 1023) / 1023.0f;
}
