// api_setup.h -- wcsph_create / wcsph_destroy / wcsph_setup (ref: HashGrid.py:10-54, ParticleData.py:142-185): allocation, the grid box, the static boundary tables.
#pragma once
extern "C" {


int wcsph_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *wcsph_last_error(wcsph_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int wcsph_create(const wcsph_params *p, wcsph_ctx **out) {
  if (!p || !out || p->n_liquid < 0 || p->n_solid < 0) return WCSPH_EARG;
  wcsph_ctx *ctx = new wcsph_ctx();
  *out = ctx;  // returned even on failure so that wcsph_last_error() works
  ctx->p = *p;
  // gathers use 32-bit byte offsets into 16-byte records (device.h nb_ld): 2^28 particles of one kind is the ceiling
  if (p->n_liquid >= (1 << 28) || p->n_solid >= (1 << 28)) return ctx->fail(WCSPH_EARG, "at most 2^28 - 1 liquid and 2^28 - 1 boundary particles per context");
  if (p->tension_fixed != 1 && p->neighbor_mode != WCSPH_MODE_FAITHFUL)
    return ctx->fail(WCSPH_EARG, "tension_fixed=0 (the literal loop of dfsph.py:264-305) runs over the reference's raw bucket lists: faithful neighbour mode only");
  HIPCHK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;
  Dev &d = ctx->d;
  memset(&d, 0, sizeof d);
  d.nl = p->n_liquid; d.ns = p->n_solid; d.n_owned = p->n_liquid;
  d.own_lo = 0; d.own_hi = 0x7fffffff;
  ctx->capacity = p->n_liquid;
  d.faithful = (p->neighbor_mode == WCSPH_MODE_FAITHFUL);
  d.table_size = p->hash_table_size > 0 ? p->hash_table_size : (p->n_liquid + p->n_solid);
  if (d.table_size < 1) d.table_size = 1;
  d.max_in_grid = p->max_in_grid; d.max_neighbour = p->max_neighbour;
  // ref: HashGrid.py:16-19 gridR = 2*particleR, searchR = 2*gridR ; kernels/CubicKernel.py:12-17
  ctx->grid_r = p->particle_radius * 2.0;
  double h = ctx->grid_r * 2.0;
  d.inv_grid = (float)(1.0 / ctx->grid_r);
  d.h = (float)h; d.h2 = d.h * d.h; d.inv_h = 1.0f / d.h;
  d.ml_h3 = (float)(48.0 / M_PI) * (float)(1.0 / (h * h * h));
  d.mk_h3 = (float)(8.0 / M_PI) * (float)(1.0 / (h * h * h));
  d.adh_mk = (float)(0.007 / pow(h, 3.25));
  { float lr = p->list_radius > 0.0f ? p->list_radius : d.h; d.list_r2 = p->list_radius < 0.0f ? 1e30f : lr * lr * 1.00001f; }
  d.VL0 = p->VL0; d.VS0 = p->VS0; d.rho_L0 = p->rho_L0; d.rho_S0 = p->rho_S0; d.mass = p->liqiudMass; d.eps = p->eps;
  d.gx = p->gravity[0]; d.gy = p->gravity[1]; d.gz = p->gravity[2];
  d.dim_coff = p->dim_coff; d.visc = p->viscosity; d.visc_b = p->viscosity_b;
  d.tension = p->tension_coff; d.tension_b = p->tension_coff_b;
  d.visc_omega = p->viscosity_omega; d.vort_coff = p->vorticity_coff; d.vort_init = p->vorticity_init;
  d.ss_VL0 = p->ss_VL0; d.ss_VS0 = p->ss_VS0; d.ss_mass = p->ss_mass; d.ss_rho0 = p->ss_rho0; d.ss_stiff = p->ss_stiff;
  d.ss_dim = p->ss_dim; d.ss_visc = p->ss_visc; d.ss_visc_b = p->ss_visc_b;
  // Q5: the reference CFL tree covers original indices [0, 2^floor(log2(N-1))); clean mode takes the true max
  d.cfl_prefix = 0x7fffffff;
  if (d.faithful && d.nl > 1) { int s = 1; while (s * 2 < d.nl) s *= 2; d.cfl_prefix = s; }
  const size_t nl = (size_t)d.nl;
  DALLOC(d.pos, nl); DALLOC(d.vel, nl); DALLOC(d.vel_guess, nl); DALLOC(d.omega, nl);
  DALLOC(ctx->pos_b, nl); DALLOC(ctx->vel_b, nl); DALLOC(ctx->vel_guess_b, nl); DALLOC(ctx->omega_b, nl);
  DALLOC(d.kappa, nl); DALLOC(d.kappa_v, nl); DALLOC(ctx->kappa_b, nl); DALLOC(ctx->kappa_v_b, nl); DALLOC(ctx->pressure_b, nl);
  DALLOC(d.oid, nl); DALLOC(ctx->oid_b, nl); DALLOC(d.where, nl);
  DALLOC(d.d_vel, nl); DALLOC(d.d_omega, nl); DALLOC(d.normal, nl);
  DALLOC(d.cg_r, nl); DALLOC(d.cg_dir, nl); DALLOC(d.cg_Ad, nl); DALLOC(d.cg_s, nl);
  DALLOC(d.minv_a, nl); DALLOC(d.minv_b, nl);
  DALLOC(d.rho, nl); DALLOC(d.adv_rho, nl); DALLOC(d.pressure, nl); DALLOC(d.vel_max, nl); DALLOC(d.alpha, nl); DALLOC(d.kbuf, nl);
  DALLOC(d.ncount, nl); DALLOC(d.nhead, nl); DALLOC(d.nsbase, nl); DALLOC(d.pk, nl); DALLOC(d.pr, nl);
  DALLOC(d.d_vel_pre, nl); DALLOC(d.pos_star, nl); DALLOC(d.vel_star, nl); DALLOC(d.d_ii, nl); DALLOC(d.dij_pj, nl);
  DALLOC(d.a_ii, nl); DALLOC(d.pressure_pre, nl);
  d.nbr_stride = (int)((nl + 63) / 64 * 64);
  d.nbr_cap = p->list_cap > 0 ? p->list_cap : 256;
  if (d.nbr_cap > p->max_neighbour) d.nbr_cap = p->max_neighbour;
  d.nbr_cap = (d.nbr_cap + 7) & ~7;
  if (d.nbr_cap < 1) d.nbr_cap = 1;
  DALLOC(d.nbl, (size_t)d.nbr_stride * (d.nbr_cap / 8)); DALLOC(d.nbs, (size_t)d.nbr_stride * (d.nbr_cap / 8));
  DALLOC(d.spos, (size_t)d.ns);
  DALLOC(d.acc, 8 * ACC_STRIPES); DALLOC(d.flags, 4 + ACC_STRIPES + 4);   // [4..]: striped vmax, then FLAG_WARM
  size_t nmax = nl > (size_t)d.ns ? nl : (size_t)d.ns;
  DALLOC(ctx->pw, nmax); DALLOC(ctx->slot, nmax);
  size_t stage_n = (nl + d.ns) * 3 > nl * 9 ? (nl + d.ns) * 3 : nl * 9;
  DALLOC(ctx->stage, stage_n);
  DALLOC(ctx->pred_dev, 4); DALLOC(ctx->dt_dev, 2);
  HIPCHK(hipMemsetAsync(ctx->pred_dev, 0, 4 * sizeof(int), ctx->stream));
  static_assert(MAIL_RING * sizeof(HostMail) <= 8 * ACC_STRIPES * sizeof(double), "mail ring fits the pinned buffer");
  HIPCHK(hipHostMalloc((void **)&ctx->h_acc_raw, 8 * ACC_STRIPES * sizeof(double)));
  HIPCHK(hipHostMalloc((void **)&ctx->h_flags, 4 * sizeof(int)));
  memset(ctx->h_acc_raw, 0, 8 * ACC_STRIPES * sizeof(double)); memset(ctx->h_flags, 0, 4 * sizeof(int));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return WCSPH_OK;
}

int wcsph_destroy(wcsph_ctx *ctx) {
  if (!ctx) return WCSPH_EARG;
  if (ctx->own_stream) hipStreamSynchronize(ctx->own_stream);
  wcsph_comm_destroy(ctx);
  for (void *p : ctx->allocs) hipFree(p);
  if (ctx->h_acc_raw) hipHostFree(ctx->h_acc_raw);
  if (ctx->tc.h_counts) hipHostFree(ctx->tc.h_counts);
  if (ctx->tc.d_nbr) hipFree(ctx->tc.d_nbr);
  { auto &m = ctx->mg; if (m.send) hipFree(m.send); if (m.recv) hipFree(m.recv); if (m.halo_send) hipFree(m.halo_send); if (m.halo_ghost) hipFree(m.halo_ghost); if (m.work) hipFree(m.work); }
  for (hipEvent_t e : ctx->prof_ev) hipEventDestroy(e);
  for (hipEvent_t e : ctx->lap_ev) hipEventDestroy(e);
  if (ctx->h_flags) hipHostFree(ctx->h_flags);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return WCSPH_OK;
}

int wcsph_set_stream(wcsph_ctx *ctx, void *s) {
  if (!ctx) return WCSPH_EARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
  return WCSPH_OK;
}

int wcsph_sync(wcsph_ctx *ctx) {
  if (!ctx) return WCSPH_EARG;
  return pull_scalars(ctx);
}

static int run_scan(wcsph_ctx *ctx, int *in, int n, int *out, const int *bbox, int clear = 0) {
  int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(BLOCK), 0, ctx->stream, ctx->d, (const int *)in, n, ctx->bsum, bbox);
  const int *bpre = nullptr;
  if (nb > SCAN_BPRE_MIN) {   // (bsum holds 2 * (scan_blocks + 1) words: the totals, then their exclusive prefix)
    int *pre = ctx->bsum + ctx->scan_blocks + 1;
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, ctx->stream, (const int *)ctx->bsum, nb, pre);
    bpre = pre;
  }
  hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BLOCK), 0, ctx->stream, ctx->d, in, n, (const int *)ctx->bsum, out, bbox, clear, bpre);
  LAUNCHCHK();
  return 0;
}

int wcsph_setup(wcsph_ctx *ctx, const float *pos_xyz, const float *maxb, const float *minb) {
  if (!ctx || !pos_xyz || !maxb || !minb) return WCSPH_EARG;
  Dev &d = ctx->d;
  if (ctx->is_setup) return ctx->fail(WCSPH_EARG, "wcsph_setup() called twice");
  // ref: HashGrid.py:46-48 blockSize = int((max - min) / gridR + 1): f32 difference, double division
  long ncell = 1;
  for (int a = 0; a < 3; a++) {
    ctx->bs[a] = (int)((double)(float)(maxb[a] - minb[a]) / ctx->grid_r + 1.0);
    if (ctx->bs[a] < 1 || ctx->bs[a] > 1024) return ctx->fail(WCSPH_EARG, "grid blockSize must be in [1,1024] per axis");
    ncell *= ctx->bs[a];
  }
  d.nx = ctx->bs[0]; d.ny = ctx->bs[1]; d.nz = ctx->bs[2]; d.x0 = 0;
  if (ctx->win_x1 >= 0) {  // slab: only columns [x0, x1) of the global grid live in this context
    int x0 = ctx->win_x0 < 0 ? 0 : ctx->win_x0, x1 = ctx->win_x1 > ctx->bs[0] ? ctx->bs[0] : ctx->win_x1;
    if (x1 <= x0) return ctx->fail(WCSPH_EARG, "empty cell window");
    d.x0 = x0; d.nx = x1 - x0;
    ncell = (long)d.nx * d.ny * d.nz;
  }
  // tile order (device.h cell_lin): the longer of x / y is the outer axis
  d.perm = d.ny > d.nx ? 1 : 0;
  d.nt0 = ((d.perm ? d.ny : d.nx) + (1 << T0B) - 1) >> T0B;
  d.nt1 = ((d.perm ? d.nx : d.ny) + (1 << T1B) - 1) >> T1B;
  d.nt2 = (d.nz + (1 << T2B) - 1) >> T2B;
  const long ncell_p = ((long)d.nt0 * d.nt1 * d.nt2) << TILE_SHIFT;
  if (ncell_p > 0x7fff0000L) return ctx->fail(WCSPH_EARG, "grid has too many cells");
  d.ncell = (int)ncell; d.ncell_p = (int)ncell_p;
  if (ctx->win_x1 >= 0) DALLOC(d.gid, (size_t)ctx->capacity + 1);   // slab rows carry their global particle id
  d.minx = minb[0]; d.miny = minb[1]; d.minz = minb[2];
  const int nscan = d.ncell_p + 2;
  ctx->scan_blocks = ((nscan > d.table_size + 2 ? nscan : d.table_size + 2) + SCAN_TILE - 1) / SCAN_TILE;   // also scans the bucket table (bucket.h)
  DALLOC(ctx->cnt, nscan); DALLOC(d.cs_l, nscan); DALLOC(d.cs_s, nscan); DALLOC(ctx->bsum, 2 * ((size_t)ctx->scan_blocks + 1));
  DALLOC(d.cell_info, d.ncell_p);
  DALLOC(d.bbox, 8); DALLOC(ctx->bbox_part, 6 * (size_t)(nblk(ctx->capacity) + 1));
  if (d.faithful) {
    DALLOC(ctx->bcount, d.table_size + 1); DALLOC(ctx->bcount_s, d.table_size + 1); DALLOC(ctx->bcount_l, d.table_size + 1);
    ctx->hb_nbins = (d.table_size + (1 << HB_SHIFT) - 1) >> HB_SHIFT;
    if (ctx->hb_nbins <= HB_MAXBINS) {   // (larger tables keep the atomics)
      // a range holds table_size / nbins buckets of a pseudo-random hash: twice its share of the rows + a margin
      ctx->hb_cap = (int)(2 * ((size_t)ctx->capacity / ctx->hb_nbins + 1) + 8192);
#if defined(WCSPH_DEV) && defined(HB_TEST_CAP)   /* test build only (-DWCSPH_DEV -DHB_TEST_CAP=100): lists that overflow at once, i.e. the atomics fall-back on every update */
      ctx->hb_cap = HB_TEST_CAP;
#endif
      DALLOC(ctx->hb_bins, HB_MAXBINS + 1); DALLOC(ctx->hb_buf, (size_t)ctx->hb_nbins * ctx->hb_cap);
    }   // [T] stays 0: the scan's end sentinel
    DALLOC(ctx->bcount8, d.table_size + 1);
  }
  {  // neighborCount by regions (grid.h k_ncount_box), both neighbour modes
    DALLOC(ctx->cellval, d.ncell);
    const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
    ctx->n_regions = ((n0 + BC0 - 1) / BC0) * ((n1 + BC1 - 1) / BC1) * ((d.nz + BCZ - 1) / BCZ);
    DALLOC(ctx->region_occ, (size_t)ctx->n_regions);
  }
  // liquids: identity order until the first grid update
  std::vector<float4> tmp((size_t)(d.nl > d.ns ? d.nl : d.ns));
  std::vector<int> ident((size_t)d.nl);
  for (int i = 0; i < d.nl; i++) {
    tmp[i] = make_float4(pos_xyz[3 * i], pos_xyz[3 * i + 1], pos_xyz[3 * i + 2], __builtin_bit_cast(float, (int)-1));
    ident[i] = i;
  }
  HIPCHK(hipMemcpyAsync(d.pos, tmp.data(), (size_t)d.nl * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d.oid, ident.data(), (size_t)d.nl * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d.where, ident.data(), (size_t)d.nl * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // solids: static, sorted once (ref: ParticleData.py:120-126; they never move: dfsph.py:577-580 is liquid only)
  ctx->solid_host.assign(pos_xyz + 3 * (size_t)d.nl, pos_xyz + 3 * ((size_t)d.nl + d.ns));
  if (d.ns > 0) {
    float4 *spos_in = ctx->pos_b;  // scratch large enough only if ns <= nl; otherwise a temporary
    float4 *tmpdev = nullptr;
    if (d.ns > d.nl) { HIPCHK(hipMalloc((void **)&tmpdev, (size_t)d.ns * sizeof(float4))); spos_in = tmpdev; }
    for (int i = 0; i < d.ns; i++) {
      const float *q = pos_xyz + 3 * ((size_t)d.nl + i);
      tmp[i] = make_float4(q[0], q[1], q[2], 0.f);
    }
    HIPCHK(hipMemcpyAsync(spos_in, tmp.data(), (size_t)d.ns * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->cnt, 0, (size_t)nscan * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_cell_assign, dim3(nblk(d.ns)), dim3(BLOCK), 0, ctx->stream, d, (const float4 *)spos_in, d.ns, ctx->cnt,
                       ctx->pw, ctx->slot, d.faithful ? ctx->bcount_s : (int *)nullptr, (int *)nullptr, d.ns);
    LAUNCHCHK();
    int rc = run_scan(ctx, ctx->cnt, nscan, d.cs_s, nullptr, 1);   // leaves the counters zeroed for the liquid updates
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_solid, dim3(nblk(d.ns)), dim3(BLOCK), 0, ctx->stream, d, (const float4 *)spos_in, d.spos, d.cs_s,
                       ctx->pw, ctx->slot);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (tmpdev) hipFree(tmpdev);
  }
  hipLaunchKernelGGL(k_cell_solid_near, dim3(nblk(d.ncell)), dim3(BLOCK), 0, ctx->stream, d, d.cell_info);
  LAUNCHCHK();
  if (d.faithful) {
    int *d_n = nullptr;
    HIPCHK(hipMalloc((void **)&d_n, sizeof(int)));
    HIPCHK(hipMemsetAsync(d_n, 0, sizeof(int), ctx->stream));
    int nb = (d.ncell + ALIAS_TB - 1) / ALIAS_TB;
    hipLaunchKernelGGL(k_alias_detect, dim3(nb), dim3(ALIAS_TB), 0, ctx->stream, d, d.cell_info, d_n);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(&ctx->n_alias, d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(d_n);
    DALLOC(d.alias, (size_t)ctx->n_alias * ALIAS_STRIDE);
    if (ctx->n_alias > 0) {
      hipLaunchKernelGGL(k_alias_fill, dim3(nb), dim3(ALIAS_TB), 0, ctx->stream, d, (const int *)d.cell_info, d.alias);
      LAUNCHCHK();
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->is_setup = true;
  ctx->grid_valid = false;
  return WCSPH_OK;
}

int wcsph_get_block_size(wcsph_ctx *ctx, int32_t bs[3]) {
  NEED_SETUP();
  bs[0] = ctx->bs[0]; bs[1] = ctx->bs[1]; bs[2] = ctx->bs[2];
  return WCSPH_OK;
}


}  // extern "C"
