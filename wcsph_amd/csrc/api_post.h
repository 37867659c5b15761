// api_post.h -- what sits either side of the path: surface reconstruction, frame dump, marching cubes, list diagnostics, the boundary sampler.
#pragma once
extern "C" {

// ---- ParticleData surface reconstruction (post-processing) ------------------------------------------
static int surf_alloc(wcsph_ctx *ctx) {
  if (ctx->surf.color) return 0;
  const size_t n = (size_t)ctx->capacity;
  DALLOC(ctx->surf.color, n); DALLOC(ctx->surf.color_grad, n); DALLOC(ctx->surf.pos_avr, n);
  DALLOC(ctx->surf.g0, n); DALLOC(ctx->surf.g1, n); DALLOC(ctx->surf.g2, n);
  return 0;
}
int wcsph_surface_color_map(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = surf_alloc(ctx);
  if (rc) return rc;
  Dev v = ctx->d;
  v.list_r2 = v.h2;   // Cubic kernel support
  if (v.nl > 0) {
    hipLaunchKernelGGL(k_surf_color, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
    hipLaunchKernelGGL(k_surf_color_grad, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
  }
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_surface_anisotropic_kernel(wcsph_ctx *ctx, float mc_searchR) {
  NEED_GRID();
  if (!(mc_searchR > 0.0f)) return ctx->fail(WCSPH_EARG, "mc_searchR must be positive");
  int rc = surf_alloc(ctx);
  if (rc) return rc;
  Dev v = ctx->d;
  ctx->surf.R2 = mc_searchR * 2.0f;                 // ParticleData.py:295 support of weight_func
  v.list_r2 = ctx->surf.R2 * ctx->surf.R2;          // entries of the walked cells beyond it weigh exactly zero
  if (v.nl > 0 && v.faithful && ctx->win_x1 < 0) {
    // weight_func's support (0.18) exceeds the walk radius: over the reference's raw rows (csrc/bucket.h)
    if ((rc = bucket_build(ctx))) return rc;
    hipLaunchKernelGGL(k_surf_pos_avr_raw, dim3(nblk(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->bk, ctx->surf);
    hipLaunchKernelGGL(k_surf_aniso_raw, dim3(nblk(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->bk, ctx->surf);
  } else if (v.nl > 0) {
    hipLaunchKernelGGL(k_surf_pos_avr, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
    hipLaunchKernelGGL(k_surf_aniso, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
  }
  LAUNCHCHK();
  return WCSPH_OK;
}

// ---- Canvas.py (headless) ---------------------------------------------------------------------------------------
int wcsph_canvas_draw(wcsph_ctx *ctx, const float *proj_view, int32_t size_x, int32_t size_y, const float *color_sphere,
                      const float *color_point, float *img) {
  NEED_SETUP();
  if (!proj_view || size_x < 1 || size_y < 1 || size_x > 16384 || size_y > 16384 || !color_sphere || !color_point || !img) return WCSPH_EARG;
  CanvasDev c;
  memcpy(c.vp, proj_view, sizeof c.vp);
  c.sx = size_x; c.sy = size_y;
  const size_t npix = (size_t)size_x * size_y;
  float *dimg = nullptr;
  c.pix = nullptr;
  if (hipMalloc((void **)&c.pix, npix * 8) != hipSuccess || hipMalloc((void **)&dimg, npix * 12) != hipSuccess) {
    if (c.pix) hipFree(c.pix);
    return ctx->fail(WCSPH_EHIP, "wcsph_canvas_draw: allocation");
  }
  const unsigned nb = (unsigned)((npix + BLOCK - 1) / BLOCK);
  auto pack = [](const float *v) { auto q = [](float x) { x = x < 0 ? 0 : (x > 1 ? 1 : x); return (unsigned)(x * 1023.0f + 0.5f); }; return (q(v[0]) << 20) | (q(v[1]) << 10) | q(v[2]); };
  hipLaunchKernelGGL(k_canvas_clear, dim3(nb), dim3(BLOCK), 0, ctx->stream, c);
  if (ctx->d.nl > 0) hipLaunchKernelGGL(k_canvas_liquid, dim3(nblk(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, c, pack(color_sphere), pack(color_point));
  if (ctx->d.ns > 0) hipLaunchKernelGGL(k_canvas_solid, dim3(nblk(ctx->d.ns)), dim3(BLOCK), 0, ctx->stream, ctx->d, c, pack(color_point));
  hipLaunchKernelGGL(k_canvas_resolve, dim3(nb), dim3(BLOCK), 0, ctx->stream, c, dimg);
  hipError_t e1 = hipMemcpyAsync(img, dimg, npix * 12, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(c.pix); hipFree(dimg);
  if (e1 != hipSuccess || e2 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_canvas_draw");
  return WCSPH_OK;
}

// ---- MarchingCubeGrid.py ----------------------------------------------------------------------------------
int wcsph_mc_tables(int32_t *edge_table, int32_t *tri_table) {
  if (!edge_table || !tri_table) return WCSPH_EARG;
  mc_build_tables(edge_table, tri_table);
  return WCSPH_OK;
}
int wcsph_mc_setup(wcsph_ctx *ctx, double grid_r, double search_r, const float *maxb, const float *minb, int32_t dims[3]) {
  NEED_SETUP();
  if (!(grid_r > 0.0) || !(search_r > 0.0) || !maxb || !minb || !dims) return WCSPH_EARG;
  McDev &m = ctx->mc;
  long n = 1;
  for (int a = 0; a < 3; a++) {   // ref: MarchingCubeGrid.py:67-69 (python doubles, int() truncation)
    dims[a] = (int32_t)(((double)maxb[a] - (double)minb[a]) / grid_r + 1.0);
    if (dims[a] < 2) return ctx->fail(WCSPH_EARG, "wcsph_mc_setup: empty box");
    n *= dims[a];
  }
  if (n > (1L << 31) - 1) return ctx->fail(WCSPH_EARG, "wcsph_mc_setup: more than 2^31 grid nodes");
  m.nx = dims[0]; m.ny = dims[1]; m.nz = dims[2];
  m.minx = minb[0]; m.miny = minb[1]; m.minz = minb[2];
  m.grid_r = (float)grid_r;
  m.h = (float)search_r; m.inv_h = 1.0f / m.h;
  m.mk_h3 = (float)((8.0 / M_PI) / (search_r * search_r * search_r));   // kernels/CubicKernel.py:13-15
  m.rho_min = ctx->p.liqiudMass * m.mk_h3;                               // MarchingCubeGrid.py:178
  if ((size_t)n > ctx->mc_nodes) { DALLOC(m.value, (size_t)n); ctx->mc_nodes = (size_t)n; }
  if (!ctx->mc_tables) {
    std::vector<int> t(256 + 256 * MC_TRI_ROW);
    if (mc_build_tables(t.data(), t.data() + 256) > 5) return ctx->fail(WCSPH_EARG, "marching-cubes table: a case with more than 5 triangles");
    DALLOC(ctx->mc_tables, t.size());
    DALLOC(m.vertex_count, 1);
    HIPCHK(hipMemcpyAsync(ctx->mc_tables, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // t lives on this stack
  }
  m.edge_table = ctx->mc_tables; m.tri_table = ctx->mc_tables + 256;
  return WCSPH_OK;
}
#define NEED_MC() do { NEED_GRID(); if (!ctx->mc.value) return ctx->fail(WCSPH_EARG, "call wcsph_mc_setup() first"); } while (0)
int wcsph_mc_surface_value(wcsph_ctx *ctx) {   // ref: MarchingCubeGrid.py:159-184 (rho must be current: compute_density)
  NEED_MC();
  const long n = (long)ctx->mc.nx * ctx->mc.ny * ctx->mc.nz;
  hipLaunchKernelGGL(k_mc_value, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->mc);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_mc_value_to_host(wcsph_ctx *ctx, float *out, int64_t n) {
  NEED_MC();
  if (!out || n != (int64_t)ctx->mc.nx * ctx->mc.ny * ctx->mc.nz) return ctx->fail(WCSPH_EARG, "wcsph_mc_value_to_host: size mismatch");
  HIPCHK(hipMemcpyAsync(out, ctx->mc.value, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return WCSPH_OK;
}
int wcsph_mc_marching_cube(wcsph_ctx *ctx, float isolevel, float *tri_xyz, int32_t cap_vertices, int32_t *n_vertices) {
  NEED_MC();
  if (cap_vertices < 0 || !n_vertices || (cap_vertices > 0 && !tri_xyz)) return WCSPH_EARG;
  McDev m = ctx->mc;
  m.isolevel = isolevel;
  m.cap = cap_vertices;
  float *dtri = nullptr;
  HIPCHK(hipMalloc((void **)&dtri, (size_t)(cap_vertices > 0 ? cap_vertices : 1) * 3 * sizeof(float)));
  m.tri = dtri;
  hipError_t e0 = hipMemsetAsync(m.vertex_count, 0, sizeof(int), ctx->stream);
  const long n = (long)m.nx * m.ny * m.nz;
  hipLaunchKernelGGL(k_mc_march, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, m);
  int count = 0;
  hipError_t e1 = hipMemcpyAsync(&count, m.vertex_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipError_t e3 = hipSuccess;
  const int stored = count < cap_vertices ? count : (cap_vertices / 3) * 3;
  if (e0 == hipSuccess && e1 == hipSuccess && e2 == hipSuccess && stored > 0)
    e3 = hipMemcpy(tri_xyz, dtri, (size_t)stored * 3 * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(dtri);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_mc_marching_cube");
  *n_vertices = count;   // > cap_vertices: the mesh was truncated (the reference prints "exceed max tri")
  return WCSPH_OK;
}

// ---- explicit neighbour list (debug / API parity) ----------------------------------------------
__global__ void k_neighbor_lists(Dev d, int i0, int n, int cap, int *counts, int *lists) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int o = i0 + t;
  int i = d.where[o];
  int *row = lists + (size_t)t * cap;
  for (int k = 0; k < cap; k++) row[k] = -1;
  counts[t] = d.ncount[i];
  float4 pi = d.pos[i];
  const int cw = __float_as_int(pi.w);
  if (cw < 0) return;
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const int info = d.cell_info[cell_lin(d, cx, cy, cz)];
  const uint16_t *e = (info >> 1) ? d.alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE : nullptr;
  int w = 0;
  for (int k = 0; k < 125; k++) {
    int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
    if ((unsigned)x >= (unsigned)d.nx || (unsigned)y >= (unsigned)d.ny || (unsigned)z >= (unsigned)d.nz) continue;
    int mult = 1;
    if (e) for (int q = 0; q < e[0]; q++) if ((e[2 + q] & 0xff) == k) mult += e[2 + q] >> 8;
    int c = cell_lin(d, x, y, z);
    for (int rep = 0; rep < mult; rep++) {
      for (int j = d.cs_l[c]; j < d.cs_l[c + 1]; j++) if (j != i) { if (w < cap) row[w] = d.oid[j]; w++; }
      for (int j = d.cs_s[c]; j < d.cs_s[c + 1]; j++) { if (w < cap) row[w] = d.nl + __float_as_int(d.spos[j].w); w++; }
    }
  }
}
int wcsph_neighbors_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists) {
  NEED_GRID();
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || cap < 1 || !counts || !lists) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  const bool exact = ctx->d.faithful && ctx->win_x1 < 0;   // the reference row itself, far aliased entries included (bucket.h)
  if (exact) { int rc = bucket_build(ctx); if (rc) return rc; }
  int *dc = nullptr, *dl = nullptr;
  HIPCHK(hipMalloc((void **)&dc, (size_t)n * 4));
  HIPCHK(hipMalloc((void **)&dl, (size_t)n * cap * 4));
  if (exact) hipLaunchKernelGGL(k_bk_rows, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->bk, i0, n, cap, dc, dl);
  else hipLaunchKernelGGL(k_neighbor_lists, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, i0, n, cap, dc, dl);
  hipError_t e1 = hipMemcpyAsync(counts, dc, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(lists, dl, (size_t)n * cap * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipStreamSynchronize(ctx->stream);
  hipFree(dc); hipFree(dl);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "neighbors_to_host copy");
  return WCSPH_OK;
}

// the compact per-step lists the passes iterate, decoded from the 16-bit delta streams (diagnostics / tests)
__global__ void k_compact_lists(Dev d, int i0, int n, int cap, int *counts, int *lists) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = d.where[i0 + t];
  int *row = lists + (size_t)t * cap;
  const int head = d.nhead[i];
  const int cl = head & 0xffff, cs = (int)((unsigned)head >> 16);
  const size_t st = (size_t)d.nbr_stride;
  const unsigned short *L = reinterpret_cast<const unsigned short *>(d.nbl + i), *S = reinterpret_cast<const unsigned short *>(d.nbs + i);
  int w = 0, hops = 0, run = i;
  for (int k = 0; k < cl; k++) {
    const int dd = (int)(short)L[(size_t)(k >> 3) * st * 8 + (k & 7)];
    run += dd;
    if (!NB_IS_ENTRY(dd)) { hops++; continue; }
    if (w < cap) row[w] = d.oid[run];
    w++;
  }
  counts[3 * t] = w;
  run = cs > 0 ? d.nsbase[i] : 0;
  for (int k = 0; k < cs; k++) {
    const int dd = (int)(short)S[(size_t)(k >> 3) * st * 8 + (k & 7)];
    run += dd;
    if (!NB_IS_ENTRY(dd)) { hops++; continue; }
    if (w < cap) row[w] = d.nl + __float_as_int(d.spos[run].w);
    w++;
  }
  counts[3 * t + 1] = w - counts[3 * t];
  counts[3 * t + 2] = hops;
  for (int k = w; k < cap; k++) row[k] = -1;
}
int wcsph_compact_lists_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists) {
  NEED_GRID();
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || cap < 1 || !counts || !lists) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  int *dc = nullptr, *dl = nullptr;
  HIPCHK(hipMalloc((void **)&dc, (size_t)n * 12));
  HIPCHK(hipMalloc((void **)&dl, (size_t)n * cap * 4));
  hipLaunchKernelGGL(k_compact_lists, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, i0, n, cap, dc, dl);
  hipError_t e1 = hipMemcpyAsync(counts, dc, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(lists, dl, (size_t)n * cap * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipStreamSynchronize(ctx->stream);
  hipFree(dc); hipFree(dl);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "compact_lists_to_host copy");
  return WCSPH_OK;
}

__global__ void __launch_bounds__(BLOCK) k_list_stats(Dev d, unsigned long long *hist) {
  const int i = sorted_index(d);
  const bool own = i < d.nl && d.oid[i] < d.n_owned;
  const int head = own ? d.nhead[i] : 0;
  const int nbl = min(((head & 0xffff) + 7) >> 3, 63), nbs = min((int)(((unsigned)head >> 16) + 7) >> 3, 63);
  if (own) { atomicAdd(&hist[nbl], 1ull); atomicAdd(&hist[64 + nbs], 1ull); }
  int ml = nbl, ms = nbs;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ml = max(ml, __shfl_xor(ml, o, WAVE)); ms = max(ms, __shfl_xor(ms, o, WAVE)); }
  int mt = min(nbl + nbs, 63);   // the trips of the merged walk (device.h walk()): the wave's maximum of liquid + solid blocks
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mt = max(mt, __shfl_xor(mt, o, WAVE));
  if ((threadIdx.x & (WAVE - 1)) == 0 && __ballot(own)) { atomicAdd(&hist[128 + ml], 1ull); atomicAdd(&hist[192 + ms], 1ull); atomicAdd(&hist[256 + mt], 1ull); }
}
int wcsph_list_stats(wcsph_ctx *ctx, int64_t hist[5][64]) {
  NEED_GRID();
  if (!hist) return WCSPH_EARG;
  unsigned long long *dh = nullptr;
  HIPCHK(hipMalloc((void **)&dh, 5 * 64 * 8));
  hipError_t e0 = hipMemsetAsync(dh, 0, 5 * 64 * 8, ctx->stream);
  if (ctx->d.nl > 0) hipLaunchKernelGGL(k_list_stats, dim3(nblk8(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, dh);
  hipError_t e1 = hipMemcpyAsync(hist, dh, 5 * 64 * 8, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(dh);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_list_stats");
  return WCSPH_OK;
}

// ---- boundary sampler --------------------------------------------------------------------------------
int wcsph_sample_surface(const float *tri_xyz, int32_t n_tri, double R, uint32_t seed, int32_t trials,
                         float *out_xyz, int32_t cap, int32_t *n_out) {
  if (!tri_xyz || n_tri < 1 || !(R > 0.0) || trials < 1 || !out_xyz || cap < 0 || !n_out) return WCSPH_EARG;
  // ref: boundry.py:128-165 -- face areas / normals (float32 arrays), bounding box, candidate count
  std::vector<float> nrm((size_t)n_tri * 3), area(n_tri);
  double total = 0.0, maxa = 0.0;
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int t = 0; t < n_tri; t++) {
    const float *a = tri_xyz + 9 * (size_t)t, *b = a + 3, *c = a + 6;
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], fminf(a[k], fminf(b[k], c[k]))); mx[k] = fmaxf(mx[k], fmaxf(a[k], fmaxf(b[k], c[k]))); }
    const double d1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, d2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
    const double n[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
    const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    area[t] = (float)(len * 0.5);
    for (int k = 0; k < 3; k++) nrm[3 * (size_t)t + k] = len > 0 ? (float)(n[k] / len) : 0.f;
    total += (double)area[t];
    if ((double)area[t] > maxa) maxa = (double)area[t];
  }
  if (!(maxa > 0.0)) return WCSPH_EARG;
  const double pi_ref = 3.1415926;   // boundry.py:54
  const long ncand_l = (long)(40.0 * (total / (pi_ref * R * R)));
  if (ncand_l < 1 || ncand_l > 400000000L) return WCSPH_EARG;
  PsDev s;
  memset(&s, 0, sizeof s);
  s.n_tri = n_tri; s.n_cand = (int)ncand_l; s.seed = seed; s.radius = (float)R; s.max_area = (float)maxa;
  const double grid_r = R / sqrt(3.0);   // boundry.py:21
  s.inv_grid = (float)(1.0 / grid_r);
  s.minx = mn[0]; s.miny = mn[1]; s.minz = mn[2];
  const long dims[3] = {(long)(((double)mx[0] - mn[0]) / grid_r) + 3, (long)(((double)mx[1] - mn[1]) / grid_r) + 3, (long)(((double)mx[2] - mn[2]) / grid_r) + 3};
  if (dims[0] * dims[1] * dims[2] > 1500000000L) return WCSPH_EARG;
  s.nx = (int)dims[0]; s.ny = (int)dims[1]; s.nz = (int)dims[2]; s.ncell = s.nx * s.ny * s.nz;
  std::vector<void *> mem;
  hipError_t herr = hipSuccess;
  auto A = [&](size_t bytes) -> void * { void *p = nullptr; if (herr == hipSuccess) herr = hipMalloc(&p, bytes ? bytes : 4); if (p) { mem.push_back(p); (void)hipMemset(p, 0, bytes ? bytes : 4); } return p; };
  float *d_tri = (float *)A((size_t)n_tri * 36), *d_nrm = (float *)A((size_t)n_tri * 12), *d_area = (float *)A((size_t)n_tri * 4);
  s.pos = (float *)A((size_t)s.n_cand * 12); s.tri_id = (int *)A((size_t)s.n_cand * 4); s.cell = (int *)A((size_t)s.n_cand * 4);
  s.cnt = (int *)A(((size_t)s.ncell + 1) * 4); s.start = (int *)A(((size_t)s.ncell + 1) * 4); s.order = (int *)A((size_t)s.n_cand * 4);
  s.samp_cnt = (int *)A((size_t)s.ncell * 4); s.samp = (int *)A((size_t)s.ncell * PS_MAX_PER_CELL * 4); s.accepted = (int *)A((size_t)s.n_cand * 4);
  s.occ_start = (int *)A(28 * 4);
  int rc = WCSPH_OK;
  auto done = [&](int code) { for (void *p : mem) (void)hipFree(p); return code; };
  if (herr != hipSuccess) return done(WCSPH_EHIP);
  s.tri = d_tri; s.nrm = d_nrm; s.area = d_area;
  (void)hipMemcpy(d_tri, tri_xyz, (size_t)n_tri * 36, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_nrm, nrm.data(), (size_t)n_tri * 12, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_area, area.data(), (size_t)n_tri * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ps_init, dim3(nblk(s.n_cand)), dim3(BLOCK), 0, 0, s);
  // cell table on the host (an offline tool: one pass over the dense cell array)
  std::vector<int> cnt((size_t)s.ncell + 1), start((size_t)s.ncell + 1);
  if (hipMemcpy(cnt.data(), s.cnt, ((size_t)s.ncell + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return done(WCSPH_EHIP);
  std::vector<int> occ[27];
  int run = 0;
  for (int c = 0; c < s.ncell; c++) {
    start[c] = run; run += cnt[c];
    if (cnt[c] > 0) { const int cz = c % s.nz, cy = (c / s.nz) % s.ny, cx = c / (s.nz * s.ny); occ[cx % 3 + 3 * (cy % 3) + 9 * (cz % 3)].push_back(c); }   // boundry.py:265
  }
  start[s.ncell] = run;
  std::vector<int> occ_all, occ_start(28, 0);
  for (int g = 0; g < 27; g++) { occ_start[g] = (int)occ_all.size(); occ_all.insert(occ_all.end(), occ[g].begin(), occ[g].end()); }
  occ_start[27] = (int)occ_all.size();
  s.occ = (int *)A(occ_all.size() * 4);
  if (herr != hipSuccess) return done(WCSPH_EHIP);
  (void)hipMemcpy(s.start, start.data(), ((size_t)s.ncell + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(s.occ, occ_all.data(), occ_all.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(s.occ_start, occ_start.data(), 28 * 4, hipMemcpyHostToDevice);
  (void)hipMemset(s.cnt, 0, ((size_t)s.ncell + 1) * 4);
  hipLaunchKernelGGL(k_ps_place, dim3(nblk(s.n_cand)), dim3(BLOCK), 0, 0, s);
  hipLaunchKernelGGL(k_ps_sort_cells, dim3(nblk(s.ncell)), dim3(BLOCK), 0, 0, s);
  for (int trial = 0; trial < trials; trial++)          // boundry.py:424-458: trial_total rounds over the 27 phase groups
    for (int pg = 0; pg < 27; pg++) {
      const int n = occ_start[pg + 1] - occ_start[pg];
      if (n > 0) hipLaunchKernelGGL(k_ps_trial, dim3(nblk(n)), dim3(BLOCK), 0, 0, s, pg, trial);
    }
  std::vector<int> acc(s.n_cand);
  std::vector<float> pos((size_t)s.n_cand * 3);
  if (hipMemcpy(acc.data(), s.accepted, (size_t)s.n_cand * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(pos.data(), s.pos, (size_t)s.n_cand * 12, hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess)
    return done(WCSPH_EHIP);
  int n = 0;
  for (int i = 0; i < s.n_cand; i++)
    if (acc[i]) { if (n < cap) { out_xyz[3 * n] = pos[3 * (size_t)i]; out_xyz[3 * n + 1] = pos[3 * (size_t)i + 1]; out_xyz[3 * n + 2] = pos[3 * (size_t)i + 2]; } n++; }
  *n_out = n;
  return done(rc);
}


}  // extern "C"
