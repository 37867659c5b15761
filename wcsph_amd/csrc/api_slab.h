// api_slab.h -- slab decomposition: windows, halo pack / unpack, whole-state load / store, and the DFSPH step of one slab in one call (no reference counterpart).
#pragma once
extern "C" {

// ---- slab decomposition -----------------------------------------------------------------------------
int wcsph_set_cell_window(wcsph_ctx *ctx, int32_t x0, int32_t x1) {
  if (!ctx || x1 <= x0) return WCSPH_EARG;
  if (ctx->is_setup) return ctx->fail(WCSPH_EARG, "wcsph_set_cell_window() must precede wcsph_setup()");
  ctx->win_x0 = x0; ctx->win_x1 = x1;
  return WCSPH_OK;
}
int wcsph_set_owned_columns(wcsph_ctx *ctx, int32_t lo, int32_t hi) {
  if (!ctx || hi <= lo) return WCSPH_EARG;
  if (ctx->is_setup) return ctx->fail(WCSPH_EARG, "wcsph_set_owned_columns() must precede wcsph_setup()");
  ctx->d.own_lo = lo; ctx->d.own_hi = hi;
  return WCSPH_OK;
}
int wcsph_set_global_liquid(wcsph_ctx *ctx, int64_t n_liquid_global) {   // Q5: the CFL prefix is defined on GLOBAL particle ids
  if (!ctx || n_liquid_global < 0) return WCSPH_EARG;
  ctx->d.cfl_prefix = 0x7fffffff;
  if (ctx->d.faithful && n_liquid_global > 1) { long s = 1; while (s * 2 < n_liquid_global) s *= 2; ctx->d.cfl_prefix = (int)(s > 0x7fffffffL ? 0x7fffffffL : s); }
  return WCSPH_OK;
}
// halo pack / unpack: several fields of the rows idx[0..n) side by side in one buffer, one launch each way
static int halo_fields(wcsph_ctx *ctx, int nf, const int32_t *fields, HaloFields *h) {
  if (nf < 1 || nf > 8 || !fields) return ctx->fail(WCSPH_EARG, "halo: 1..8 fields");
  h->n = nf; h->W = 0; h->pred = nullptr;
  for (int k = 0; k < nf; k++) {
    FieldDesc fd = field_desc(ctx, fields[k]);
    if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "halo: vec3 / scalar fields only");
    h->ptr[k] = (const float *)fd.ptr; h->width[k] = fd.kind == 3 ? 3 : 1; h->stride[k] = fd.kind == 3 ? 4 : 1;
    h->W += h->width[k];
  }
  return 0;
}
int wcsph_slab_classify(wcsph_ctx *ctx, const void *d_rows, int64_t n, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right,
                        int64_t *d_idx, int32_t cap, int32_t *d_counts) {
  NEED_SETUP();
  if (n < 0 || cap < 0 || (n > 0 && !d_rows) || !d_idx || !d_counts) return WCSPH_EARG;
  HIPCHK(hipMemsetAsync(d_counts, 0, 5 * sizeof(int32_t), ctx->stream));
  if (n > 0)
    hipLaunchKernelGGL(k_slab_classify, dim3((unsigned)((n + BLOCK * CLS_ROWS - 1) / (BLOCK * CLS_ROWS))), dim3(BLOCK), 0, ctx->stream, (const float *)d_rows, (long long)n, STATE_W,
                       ctx->d.minx, ctx->d.inv_grid, lo, hi, ghost, has_left, has_right, (long long *)d_idx, cap, d_counts);
  LAUNCHCHK();
  return WCSPH_OK;
}

// ---- slab state resident in the library between steps (round 5): the driver no longer stores and reloads every row per step
// (2 x 120 MB at C3, 0.1 ms) -- it classifies the context's own rows, takes out the few that travel and applies what arrives.
int wcsph_state_classify(wcsph_ctx *ctx, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right, int64_t *d_idx, int32_t cap, int32_t *d_counts) {
  NEED_SETUP();
  if (cap < 0 || !d_idx || !d_counts) return WCSPH_EARG;
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "wcsph_state_classify: the state has pending changes (wcsph_state_apply) that no grid update has sorted in yet");
  HIPCHK(hipMemsetAsync(d_counts, 0, 5 * sizeof(int32_t), ctx->stream));
  // (in sorted order when the state is sorted -- always between two steps of wcsph_slab_step --, through where[] right after a load)
  const long long n = ctx->d.n_owned;
  if (n > 0 && ctx->grid_valid)
    hipLaunchKernelGGL(k_state_classify_sorted, dim3((unsigned)((ctx->d.nl + BLOCK * CLS_ROWS_SORTED - 1) / (BLOCK * CLS_ROWS_SORTED))), dim3(BLOCK), 0, ctx->stream, (const float4 *)ctx->d.pos,
                       (const int *)ctx->d.oid, ctx->d.nl, ctx->d.n_owned, ctx->d.minx, ctx->d.inv_grid, lo, hi, ghost, has_left, has_right, (long long *)d_idx, cap, d_counts, (const int *)nullptr);
  else if (n > 0)
    hipLaunchKernelGGL(k_slab_classify, dim3((unsigned)((n + BLOCK * CLS_ROWS - 1) / (BLOCK * CLS_ROWS))), dim3(BLOCK), 0, ctx->stream, (const float *)ctx->d.pos, n, 4,
                       ctx->d.minx, ctx->d.inv_grid, lo, hi, ghost, has_left, has_right, (long long *)d_idx, cap, d_counts, (const int *)ctx->d.where);
  LAUNCHCHK();
  return WCSPH_OK;
}
}  // extern "C"
// (the counts are left ZERO for the next classification: no memset launch per step -- a fill kernel costs ~5 us)
__global__ void k_counts_to_host(int *c, int *h, const int *pred, const int *nbr) {
  if (pred && *pred == 0) return;
  if (threadIdx.x < 5) { h[threadIdx.x] = c[threadIdx.x]; c[threadIdx.x] = 0; }
  if (nbr && threadIdx.x >= 8 && threadIdx.x < 24) h[threadIdx.x] = nbr[threadIdx.x - 8];
}
// wcsph_state_classify in the tail of wcsph_slab_step, gated like the launches around it; the counts go to pinned host memory
// in front of the mail that follows.  native != nullptr (the library's own RCCL plane): the counts also go to the two slab
// neighbours and theirs come back (one grouped send / recv of five words, unconditional like every exchange of the step) and
// travel in the same mail -- the next migration then knows every message size without a header round trip (api_migrate.h)
static int tail_classify(wcsph_ctx *ctx, const int *pred, NativeComm *native) {
  auto &t = ctx->tc;
  // The tail counts into the library's OWN five words (t.d_nbr + 16; the caller's d_counts belongs to wcsph_state_classify, which
  // leaves its result there).  They are zero here: zeroed at allocation and by every k_counts_to_host that read them; a gated-off
  // classification leaves them so.
  int *const mine = t.d_nbr + 16;
  // (the step has just written the positions in sorted order: walked so, coalesced -- k_state_classify_sorted)
  if (ctx->d.n_owned > 0)
    hipLaunchKernelGGL(k_state_classify_sorted, dim3((unsigned)((ctx->d.nl + BLOCK * CLS_ROWS_SORTED - 1) / (BLOCK * CLS_ROWS_SORTED))), dim3(BLOCK), 0, ctx->stream, (const float4 *)ctx->d.pos,
                       (const int *)ctx->d.oid, ctx->d.nl, ctx->d.n_owned, ctx->d.minx, ctx->d.inv_grid, t.lo, t.hi, t.ghost, t.has_l, t.has_r, t.d_idx, t.cap, mine, pred);
  t.nbr_queued = false;
  if (native) {
    int rc = comm_counts_exchange(native, mine, t.d_nbr, ctx->stream);
    if (rc) return rc;
    t.nbr_queued = true;
  }
  hipLaunchKernelGGL(k_counts_to_host, dim3(1), dim3(WAVE), 0, ctx->stream, mine, t.h_counts, pred, (const int *)(native ? t.d_nbr : nullptr));
  LAUNCHCHK();
  return 0;
}
// the tail of a step that ended on its plain path: its mail has not been waited for yet
static int tail_collect(wcsph_ctx *ctx) {
  auto &t = ctx->tc;
  if (!t.pending_seq) return 0;
  const unsigned long long seq = t.pending_seq;
  t.pending_seq = 0;
  const float err_keep = ctx->avg_density_err;
  int rc = collect_mail(ctx, true, seq);
  ctx->avg_density_err = err_keep;
  if (rc) return rc;
  t.valid = true; t.nbr_valid = t.nbr_queued;
  return 0;
}
extern "C" {
int wcsph_slab_set_tail_classify(wcsph_ctx *ctx, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right, int64_t *d_idx, int32_t cap, int32_t *d_counts) {
  NEED_SETUP();
  auto &t = ctx->tc;
  { int rc0 = tail_collect(ctx); if (rc0) return rc0; }   // (a plain-path tail still in flight writes the OLD buffers: let it land first)
  t.valid = false;      // (the neighbours' counts of the last step stay valid: they do not depend on this rank's buffers)
  if (!d_idx || !d_counts || cap <= 0) { t.on = false; return WCSPH_OK; }
  if (!t.h_counts) { HIPCHK(hipHostMalloc((void **)&t.h_counts, 32 * sizeof(int))); memset(t.h_counts, 0, 32 * sizeof(int)); }
  t.lo = lo; t.hi = hi; t.ghost = ghost; t.has_l = has_left; t.has_r = has_right; t.d_idx = (long long *)d_idx; t.cap = cap; t.d_counts = d_counts;
  // [0..5) / [8..13): the left / right neighbour's counts (native plane); [16..21): this rank's, zero between two tails
  if (!t.d_nbr) { HIPCHK(hipMalloc((void **)&t.d_nbr, 24 * sizeof(int))); HIPCHK(hipMemsetAsync(t.d_nbr, 0, 24 * sizeof(int), ctx->stream)); }
  t.on = true;
  return WCSPH_OK;
}
// WCSPH_OK / 2: counts[5] = the result of the classification taken in the last wcsph_slab_step's tail (consumed by this call; the
// index lists are in the registered buffer); 1: none available: call wcsph_state_classify.  WCSPH_OK: the step ended on its
// speculative path, the counts were on the host already; 2: it ended on its plain path and this call has waited for them.
int wcsph_slab_tail_classify_result(wcsph_ctx *ctx, int32_t counts[5]) {
  NEED_SETUP();
  if (!counts) return WCSPH_EARG;
  const bool waited = ctx->tc.pending_seq != 0;
  int rc = tail_collect(ctx);
  if (rc) return rc;
  if (!ctx->tc.on || !ctx->tc.valid || ctx->nl_phys > 0) return 1;
  for (int k = 0; k < 5; k++) counts[k] = ctx->tc.h_counts[k];
  ctx->tc.valid = false;
  return waited ? 2 : WCSPH_OK;
}
// WCSPH_OK: the slab neighbours' classification counts of the same moment (native data plane only: exchanged in the step's tail);
// 1: none available.  A missing neighbour's counts are zero.  Not consumed: wcsph_slab_migrate reads them too.
int wcsph_slab_tail_neighbour_counts(wcsph_ctx *ctx, int32_t left[5], int32_t right[5]) {
  NEED_SETUP();
  if (!left || !right) return WCSPH_EARG;
  int rc = tail_collect(ctx);
  if (rc) return rc;
  if (!ctx->tc.nbr_valid) return 1;
  for (int k = 0; k < 5; k++) { left[k] = ctx->tc.h_counts[8 + k]; right[k] = ctx->tc.h_counts[16 + k]; }
  return WCSPH_OK;
}
int wcsph_state_gather_rows(wcsph_ctx *ctx, const int64_t *d_idx, int32_t n, float *d_rows) {
  NEED_SETUP();
  if (n < 0 || (n > 0 && (!d_idx || !d_rows))) return WCSPH_EARG;
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "wcsph_state_gather_rows: the state has pending changes (wcsph_state_apply)");
  if (n > 0) hipLaunchKernelGGL(k_state_gather_rows, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, (const long long *)d_idx, n, d_rows);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_state_apply(wcsph_ctx *ctx, int32_t n_owned_new, const int64_t *d_fold_dst, const int64_t *d_fold_src, int32_t n_fold,
                      const int64_t *d_write_idx, const float *d_write_rows, int32_t n_write, const float *d_ghost_rows, int32_t n_ghost) {
  NEED_SETUP();
  if (n_owned_new < 0 || n_fold < 0 || n_write < 0 || n_ghost < 0 || (long)n_owned_new + n_ghost > ctx->capacity) return ctx->fail(WCSPH_EARG, "wcsph_state_apply: counts exceed the context capacity");
  if ((n_fold > 0 && (!d_fold_dst || !d_fold_src)) || (n_write > 0 && (!d_write_idx || !d_write_rows)) || (n_ghost > 0 && !d_ghost_rows)) return WCSPH_EARG;
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "wcsph_state_apply: called twice without a grid update in between");
  Dev &d = ctx->d;
  const int nl_old = d.nl;
  hipStream_t s = ctx->stream;
  if (n_fold > 0) hipLaunchKernelGGL(k_state_relabel, dim3(nblk(n_fold)), dim3(BLOCK), 0, s, d, (const long long *)d_fold_dst, (const long long *)d_fold_src, n_fold);
  if (n_write > 0) hipLaunchKernelGGL(k_state_write_rows, dim3(nblk(n_write)), dim3(BLOCK), 0, s, d, (const long long *)d_write_idx, 0, d_write_rows, n_write, nl_old);
  if (n_ghost > 0) hipLaunchKernelGGL(k_state_write_rows, dim3(nblk(n_ghost)), dim3(BLOCK), 0, s, d, (const long long *)nullptr, n_owned_new, d_ghost_rows, n_ghost, nl_old);
  LAUNCHCHK();
  d.n_owned = n_owned_new; d.nl = n_owned_new + n_ghost;
  ctx->nl_phys = nl_old > d.nl ? nl_old : d.nl;   // rows [d.nl, nl_old) by original index are dead until the sort drops them
  ctx->grid_valid = ctx->hist_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}
int wcsph_halo_pack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, void *d_out) {
  NEED_COMPACT();
  HaloFields h;
  int rc = halo_fields(ctx, n_fields, fields, &h);
  if (rc) return rc;
  if (n <= 0) return WCSPH_OK;
  hipLaunchKernelGGL(k_halo_pack, dim3(nblk(n * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, d_idx, n, (float *)d_out);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_halo_unpack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, const void *d_in) {
  NEED_COMPACT();
  HaloFields h;
  int rc = halo_fields(ctx, n_fields, fields, &h);
  if (rc) return rc;
  if (n <= 0) return WCSPH_OK;
  hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(n * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, d_idx, n, (const float *)d_in);
  LAUNCHCHK();
  for (int k = 0; k < n_fields; k++) if (fields[k] == WCSPH_F_POS) ctx->grid_valid = ctx->hist_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}
int wcsph_load_state(wcsph_ctx *ctx, int32_t n_owned, int32_t n_total, const float *d_rows) {
  NEED_SETUP();
  if (n_owned < 0 || n_total < n_owned || n_total > ctx->capacity) return ctx->fail(WCSPH_EARG, "load_state: counts exceed the context capacity");
  ctx->d.nl = n_total; ctx->d.n_owned = n_owned; ctx->nl_phys = 0;
  if (n_total > 0) {
    if (!d_rows) return WCSPH_EARG;
    hipLaunchKernelGGL(k_load_state, dim3(nblk(n_total)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_rows, n_total, 0);
    LAUNCHCHK();
  }
  ctx->grid_valid = ctx->hist_valid = false;
  return WCSPH_OK;
}
int wcsph_load_state_split(wcsph_ctx *ctx, int32_t n_owned, const float *d_owned_rows, int32_t n_ghost, const float *d_ghost_rows) {
  NEED_SETUP();
  if (n_owned < 0 || n_ghost < 0 || (long)n_owned + n_ghost > ctx->capacity) return ctx->fail(WCSPH_EARG, "load_state: counts exceed the context capacity");
  if ((n_owned > 0 && !d_owned_rows) || (n_ghost > 0 && !d_ghost_rows)) return WCSPH_EARG;
  ctx->d.nl = n_owned + n_ghost; ctx->d.n_owned = n_owned; ctx->nl_phys = 0;
  if (n_owned > 0) hipLaunchKernelGGL(k_load_state, dim3(nblk(n_owned)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_owned_rows, n_owned, 0);
  if (n_ghost > 0) hipLaunchKernelGGL(k_load_state, dim3(nblk(n_ghost)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_ghost_rows, n_ghost, n_owned);
  LAUNCHCHK();
  ctx->grid_valid = ctx->hist_valid = false;
  return WCSPH_OK;
}
int wcsph_store_state(wcsph_ctx *ctx, float *d_rows) {
  NEED_SETUP();
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "wcsph_store_state: the state has pending changes (wcsph_state_apply) that no grid update has sorted in yet");
  if (ctx->d.nl > 0) {
    if (!d_rows) return WCSPH_EARG;
    if (ctx->d.n_owned > 0) hipLaunchKernelGGL(k_store_state, dim3(nblk(ctx->d.n_owned)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_rows);
    LAUNCHCHK();
  }
  return WCSPH_OK;
}
int wcsph_get_acc(wcsph_ctx *ctx, int slot, double *v) {
  if (!ctx || slot < 0 || slot >= 8 || !v) return WCSPH_EARG;
  int rc = pull_scalars(ctx);
  if (rc) return rc;
  *v = ctx->h_acc[slot];
  return WCSPH_OK;
}
int wcsph_set_acc(wcsph_ctx *ctx, int slot, double v) {
  if (!ctx || slot < 0 || slot >= 8) return WCSPH_EARG;
  double row[ACC_STRIPES] = {0};
  row[0] = v;
  HIPCHK(hipMemcpyAsync(ctx->d.acc + slot * ACC_STRIPES, row, sizeof row, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // row lives on this stack
  return WCSPH_OK;
}
int wcsph_set_acc_buffer(wcsph_ctx *ctx, void *device_doubles) {
  NEED_SETUP();
  if (!device_doubles) return WCSPH_EARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->d.acc = (double *)device_doubles;   // the context's own buffer stays in ctx->allocs and is freed with it
  HIPCHK(hipMemsetAsync(ctx->d.acc, 0, 8 * ACC_STRIPES * sizeof(double), ctx->stream));
  return WCSPH_OK;
}
int wcsph_set_part(wcsph_ctx *ctx, int32_t mode, int32_t x_lo, int32_t x_hi) {
  if (!ctx || mode < 0 || mode > 2) return WCSPH_EARG;
  ctx->d.part_mode = mode; ctx->d.part_x0 = x_lo; ctx->d.part_x1 = x_hi > x_lo ? x_hi : x_lo;
  return WCSPH_OK;
}
int wcsph_phase(wcsph_ctx *ctx, int ph) {
  NEED_GRID();
  int rc = 0;
  const float dt = ctx->dt;
  switch (ph) {
    case WCSPH_PH_DENSITY_COFF: PASS(k_density_coff); ctx->pr_valid = false; break;   // ghost rows: rho arrives by halo exchange
    case WCSPH_PH_DENSITY_COFF_WARM: PASS(k_density_coff_warm, dt); ctx->pr_valid = false; break;
    case WCSPH_PH_VISC_MINV_RESIDUAL: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_DZERO))) return rc; RPASS(k_visc_minv_residual, dt); break;
    case WCSPH_PH_VISC_MINV_RESIDUAL_VORT: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_DZERO))) return rc; RPASS(k_visc_minv_residual_vort, dt); break;
    case WCSPH_PH_DRHO_DIV_WARM: HOT_DRHO_DIV(DRHO_WARM, dt); break;
    case WCSPH_PH_VEL_WARM: HOT_VEL(1, ctx->d.kappa_v, dt, 1); break;
    case WCSPH_PH_DRHO_DIV_BEGIN: HOT_DRHO_DIV(DRHO_BEGIN, dt); break;
    case WCSPH_PH_MAKE_K_DIV: ELEM(k_make_k, ctx->d.kappa_v, 0.0f); break;
    case WCSPH_PH_VEL_ITER: HOT_VEL(0, ctx->d.kbuf, dt, ctx->d.n_owned < ctx->d.nl); break;
    case WCSPH_PH_DRHO_DIV_ITER: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_ERR))) return rc; HOT_DRHO_DIV(DRHO_ITER, dt); break;
    case WCSPH_PH_END_DIV: ELEM(k_end_div, dt); break;
    case WCSPH_PH_WARM_PRESS_SCALE: ELEM(k_warm_press_scale, dt, 0); break;
    case WCSPH_PH_WARM_PRESS_VEL: PASS(k_warm_press_vel, dt); break;
    case WCSPH_PH_DRHO_PRESS_BEGIN: HOT_DRHO_PRESS(DRHO_BEGIN, dt); break;
    case WCSPH_PH_MAKE_K_PRESS: ELEM(k_make_k, ctx->d.kappa, 1.0f); break;
    case WCSPH_PH_DRHO_PRESS_ITER: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_ERR))) return rc; HOT_DRHO_PRESS(DRHO_ITER, dt); break;
    case WCSPH_PH_END_PRESS: ELEM(k_end_press, dt); break;
    case WCSPH_PH_CLEAR: ELEM(k_clear_nonpressure); break;
    // tension_fixed = 0: the reference loop literally, over its raw bucket rows (bucket.h).  On a slab window the table
    // spans all ranks (far aliased entries name particles of other slabs): bucket_build refuses with a message.
    case WCSPH_PH_TENSION_NORMAL:
      if (ctx->p.tension_fixed == 0) {
        if (ctx->d.n_owned < ctx->d.nl) return ctx->fail(WCSPH_EARG, "the literal compute_tension (tension_fixed = 0) needs a context that owns every row");
        if ((rc = bucket_build_rows(ctx))) return rc;
        PASS(k_tension_normal_literal, ctx->bk);
      }
      else RPASS(k_tension_normal);
      break;
    case WCSPH_PH_TENSION_ACCEL:
      if (ctx->p.tension_fixed == 0) {
        if (ctx->d.n_owned < ctx->d.nl) return ctx->fail(WCSPH_EARG, "the literal compute_tension (tension_fixed = 0) needs a context that owns every row");
        if ((rc = bucket_build_rows(ctx))) return rc;
        ELEM(k_tension_accel_literal, ctx->bk);
      }
      else PASS(k_tension_accel);
      break;
    case WCSPH_PH_VISC_GUESS: ELEM(k_visc_guess); break;
    case WCSPH_PH_VISC_MINV: RPASS(k_visc_minv, dt); break;
    case WCSPH_PH_VISC_RESIDUAL: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_DZERO))) return rc; RPASS(k_visc_residual, dt); break;
    case WCSPH_PH_VISC_AD: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_DAD))) return rc; RPASS(k_visc_Ad, dt); break;
    case WCSPH_PH_VISC_UPDATE: if ((rc = zero_acc(ctx, ACC_DNEW))) return rc; ELEM(k_visc_update, ctx->cg_delta); break;
    case WCSPH_PH_VISC_DIR: ELEM(k_visc_dir, ctx->cg_delta); break;
    case WCSPH_PH_VISC_END: ELEM(k_visc_end, dt); break;
    case WCSPH_PH_VORTICITY: RPASS(k_vorticity, dt); break;
    case WCSPH_PH_VORTICITY_INT: ELEM(k_vorticity_integrate, dt); break;
    case WCSPH_PH_UPDATE_VEL: ELEM(k_update_vel, dt); break;
    case WCSPH_PH_UPDATE_POS: ELEM(k_update_pos, dt); break;
    case WCSPH_PH_END_DIV_BEGIN_NONPRESSURE: ELEM(k_end_div_begin_nonpressure, dt, 0); break;
    case WCSPH_PH_UPDATE_VEL_WARM_SCALE: ELEM(k_update_vel_warm_scale, dt, 0); break;
    case WCSPH_PH_END_PRESS_UPDATE_POS: ELEM(k_end_press_update_pos, dt); break;
    default: return ctx->fail(WCSPH_EARG, "unknown phase id");
  }
  return WCSPH_OK;
}

// ---- the slab step in one call (include/wcsph.h "the whole DFSPH step of one slab") ----------------------------------
int wcsph_slab_set_halo(wcsph_ctx *ctx, const int32_t *d_send_idx, int32_t n_send, const int32_t *d_ghost_idx, int32_t n_ghost) {
  NEED_SETUP();
  if (n_send < 0 || n_ghost < 0 || (n_send > 0 && !d_send_idx) || (n_ghost > 0 && !d_ghost_idx)) return WCSPH_EARG;
  ctx->halo_send = d_send_idx; ctx->n_halo_send = n_send; ctx->halo_ghost = d_ghost_idx; ctx->n_halo_ghost = n_ghost;
  // new lists: how they split between the two neighbours has to be said again (wcsph_comm_set_halo_split) before the next
  // exchange of a native communicator -- a stale split against shorter lists would make a row count wrap around
  if (ctx->comm) ctx->comm->n_send_left = ctx->comm->n_ghost_left = -1;
  return WCSPH_OK;
}
}  // extern "C" (templates below)
// ghost rows' gather records.  (pos, rho) record: rho arrives with the first halo refresh of the step.  (pos, k) record of the
// warm-start velocity pass: k = the rescaled kappa_v (dfsph.py:416), formed HERE from the kappa_v the ghost row brought with it --
// the same expression on the same bits as its owner's k_density_coff_warm, so nothing has to travel (round 6: one exchange per
// step less).  Their positions do not change during the step.
__global__ void k_ghost_records(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || d.oid[i] < d.n_owned) return;
  const float4 p = d.pos[i];
  d.pk[i] = make_float4(p.x, p.y, p.z, 0.5f * fmaxf(d.kappa_v[i] / dt, -0.5f * d.rho_L0 * d.rho_L0));
  d.pr[i] = make_float4(p.x, p.y, p.z, 0.f);
}
namespace {
struct PartGuard { Dev &d; ~PartGuard() { d.part_mode = 0; } };
struct SlabRun {
  wcsph_ctx *ctx;
  const wcsph_slab_callbacks *cb;
  bool multi, overlap;
  int ilo, ihi, n_xchg = 0, n_wait = 0;
  HaloFields h;
  void hf_begin() { h.n = 0; h.W = 0; h.pred = nullptr; }
  void hf_add(const void *ptr, int width, int stride) { h.ptr[h.n] = (const float *)ptr; h.width[h.n] = width; h.stride[h.n] = stride; h.W += width; h.n++; }
  // refresh the ghost rows of the fields collected in `h`, then run `launch()`, a neighbour pass that reads them from
  // neighbours j -- overlapped: the pass runs on the slab interior while the exchange is in flight, on the rest once it
  // has landed.  launch == nullptr-like (no_pass): exchange only.
  // also: a second set of destinations for the same received columns (rho lands in the field AND in the (pos, rho) records)
  template <class F>
  int xp(bool has_pass, F &&launch, const HaloFields *also = nullptr) {
    int rc;
    if (!multi) return has_pass ? launch() : 0;
    void *snd = nullptr, *rcv = nullptr;
    if ((rc = cb->halo_buffers(cb->user, h.W, &snd, &rcv))) return ctx->fail(rc, "slab callback halo_buffers failed");
    h.pred = ctx->d.pred;   // a speculatively queued refresh packs / unpacks nothing when its predicate is 0 (the exchange itself is unconditional)
    if (ctx->n_halo_send > 0) {
      hipLaunchKernelGGL(k_halo_pack, dim3(nblk(ctx->n_halo_send * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, ctx->halo_send, ctx->n_halo_send, (float *)snd);
      LAUNCHCHK();
    }
    if ((rc = cb->halo_post(cb->user, h.W))) return ctx->fail(rc, "slab callback halo_post failed");
    n_xchg++;
    const bool split = overlap && has_pass;
    // Native plane, mode 2 (csrc/api_comm.h): the whole boundary chain -- exchange, unpack, the pass on the rows next to the cuts --
    // runs on the communicator's halo stream BESIDE the pass on the interior; the context's stream joins it once, at the end.
    // In stream order there is then no event hop between the exchange and the unpack, and the short boundary launch (latency-bound
    // on its own: ~20 us for an eighth of the rows) shares the GPU with the interior launch instead of following it.  Safe
    // because the two launches write disjoint rows, interior rows read no ghost row, and every reduction slot is only added to.
    NativeComm *nc = (split && ctx->comm && cb == &ctx->comm->cb && ctx->comm->mode == 2) ? ctx->comm : nullptr;
    if (nc) {
      ctx->d.part_mode = 1; ctx->d.part_x0 = ilo; ctx->d.part_x1 = ihi > ilo ? ihi : ilo;
      if ((rc = launch())) { ctx->d.part_mode = 0; return rc; }
      hipStream_t keep = ctx->stream;
      ctx->stream = nc->halo;
      rc = unpack(rcv, also);
      if (!rc) { ctx->d.part_mode = 2; rc = launch(); }
      ctx->stream = keep;
      ctx->d.part_mode = 0;
      if (rc) return rc;
      HIPCHK(hipEventRecord(nc->ev_chain, nc->halo));
      HIPCHK(hipStreamWaitEvent(ctx->stream, nc->ev_chain, 0));
      nc->halo_pending = false;
      return 0;
    }
    if (split) { ctx->d.part_mode = 1; ctx->d.part_x0 = ilo; ctx->d.part_x1 = ihi > ilo ? ihi : ilo; if ((rc = launch())) return rc; }
    if ((rc = cb->halo_wait(cb->user))) return ctx->fail(rc, "slab callback halo_wait failed");
    if ((rc = unpack(rcv, also))) return rc;
    if (!has_pass) return 0;
    if (split) ctx->d.part_mode = 2;
    rc = launch();
    ctx->d.part_mode = 0;
    return rc;
  }
  int unpack(const void *rcv, const HaloFields *also) {   // on ctx->stream (whichever stream that currently is)
    if (ctx->n_halo_ghost > 0) {
      hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ctx->n_halo_ghost * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, ctx->halo_ghost, ctx->n_halo_ghost, (const float *)rcv);
      if (also) {
        HaloFields a2 = *also; a2.pred = ctx->d.pred;
        hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ctx->n_halo_ghost * h.W)), dim3(BLOCK), 0, ctx->stream, a2, (const int *)ctx->d.where, ctx->halo_ghost, ctx->n_halo_ghost, (const float *)rcv);
      }
      LAUNCHCHK();
    }
    return 0;
  }
  int allreduce(int slot) {   // the stripes of a reduction slot summed over the ranks, in place on the device
    int rc;
    if (multi && (rc = cb->allreduce_acc(cb->user, slot))) return ctx->fail(rc, "slab callback allreduce_acc failed");
    return 0;
  }
  // slots a..b (contiguous; what lies between must be zero on every rank or not yet needed) in one call on the native plane, one
  // call per end slot behind callbacks
  int allreduce_range(int a, int b) {
    if (!multi) return 0;
    if (ctx->comm && cb == &ctx->comm->cb) {
      int rc = native_allreduce_slots(ctx->comm, a, b - a + 1);
      return rc ? ctx->fail(rc, "native all-reduce of the reduction slots failed") : 0;
    }
    int rc = allreduce(a);
    return rc ? rc : allreduce(b);
  }
  int fetch() { n_wait++; return pull_scalars(ctx); }   // one host wait: every slot total + the error flags
};
}  // namespace
extern "C" {
#define SRC(x) do { if ((rc = (x))) { ctx->d.part_mode = 0; ctx->d.pred = nullptr; ctx->d.dtp = nullptr; return rc; } } while (0)
#define SPASS_(kern, ...) [&]() -> int { PASS(kern, ##__VA_ARGS__); return 0; }
// deltaT from the GLOBAL velocity maximum on the device: the slab step's k_cfl_store leaves this rank's maximum in
// flags[FLAG_VMAX], the caller's MAX all-reduce makes it the global one, and this forms the time step (DtRule: the host rule
// of dfsph.py:113-129 in the same double expressions) for the kernels queued behind it (Dev::dtp)
__global__ void k_dt_from_vmax(Dev d, DtRule r) {
  PRED_GATE(d);
  if (threadIdx.x == 0) { const float dt = dt_rule(r, __int_as_float(d.flags[FLAG_VMAX])); *r.out = dt; d.flags[FLAG_DT] = __float_as_int(dt); }
}
int wcsph_slab_step(wcsph_ctx *ctx, const wcsph_slab_callbacks *cb, int64_t n_global, int32_t interior_lo, int32_t interior_hi,
                    int32_t flags, int32_t stats[2]) {
  NEED_SETUP();
  if (n_global < 1) return WCSPH_EARG;
  if (!cb && ctx->comm) cb = &ctx->comm->cb;   // the native RCCL data plane (api_comm.h)
  // flags bit 3: every exchange point calls the communicator even with one rank (empty or loop-back halos, all-reduces over one
  // rank): the host cost of the communicator path, and the data plane itself, can be exercised on one GPU
  const bool multi = cb && (cb->world > 1 || (flags & 8));
  SlabRun R{ctx, cb, multi, (flags & 1) != 0 && multi, interior_lo, interior_hi};
  if (R.multi && (!cb->halo_buffers || !cb->halo_post || !cb->halo_wait || !cb->allreduce_acc || !cb->allreduce_max || !cb->allreduce_words)) return WCSPH_EARG;
  Dev &d = ctx->d;
  int rc;
  d.pred = nullptr; d.dtp = nullptr;
  ctx->tc.valid = false; ctx->tc.nbr_valid = false; ctx->tc.pending_seq = 0;
  PartGuard part_guard{d};
  NativeComm *const tail_native = (cb && ctx->comm && cb == &ctx->comm->cb) ? ctx->comm : nullptr;
  const double N = (double)n_global;
  const bool tension = (flags & 2) && !(d.tension == 0.0f && d.tension_b == 0.0f);
  if (tension && ctx->p.tension_fixed == 0 && R.multi) return ctx->fail(WCSPH_EARG, "the literal compute_tension reads the reference's raw bucket rows, which span all slabs: tension_fixed = 1 on a decomposition");
  // a NaN error sum carried in would make `avg_density_err > err` false, i.e. zero divergence iterations, with every sum that
  // rides on the last iteration (vorticity) left stale
  if (!std::isfinite(ctx->avg_density_err)) return ctx->fail(WCSPH_ENONFINITE, "wcsph_slab_step: non-finite avg_density_err carried in");
  auto no_pass = []() -> int { return 0; };
  // ---- particle_data.hash_grid.update_grid(): with the reference's GLOBAL bucket table in faithful mode (HashGrid.py:108-114).
  // The all-reduce of this step's liquid histogram (count / 2 words: 40 MB per rank at C4) is POSTED as soon as the rank's own
  // share is packed and AWAITED only where neighborCount needs it: the scan, the sort and the list build -- most of the
  // update's time, reading neither the histogram nor neighborCount -- run while it is in flight.
  if (d.faithful && R.multi) {
    void *words = nullptr; int64_t nw = 0;
    SRC(wcsph_grid_update_begin(ctx));
    SRC(wcsph_bucket_hist_pack(ctx, &words, &nw));
    const bool split = cb->allreduce_words_post && cb->allreduce_words_wait;
    if (split) { if ((rc = cb->allreduce_words_post(cb->user, words, nw))) return ctx->fail(rc, "slab callback allreduce_words_post failed"); }
    else if ((rc = cb->allreduce_words(cb->user, words, nw))) return ctx->fail(rc, "slab callback allreduce_words failed");
    // (the posted all-reduce sums ctx->bcount in place while grid_sort / grid_lists are queued: neither touches bcount -- api_grid.h --
    // and a failure in between still awaits it, so that the next step's pack never overwrites a buffer under a pending all-reduce)
    rc = wcsph_grid_update_mid(ctx);
    if (split) { const int rc2 = cb->allreduce_words_wait(cb->user); if (!rc && rc2) rc = ctx->fail(rc2, "slab callback allreduce_words_wait failed"); }
    SRC(rc);
    SRC(wcsph_bucket_hist_unpack(ctx));
    SRC(wcsph_grid_update_end(ctx));
  } else SRC(wcsph_grid_update(ctx));
  if (d.nl == 0) { if (stats) stats[0] = stats[1] = 0; return WCSPH_OK; }
  if (d.n_owned < d.nl) ELEM(k_ghost_records, ctx->dt);
  if (R.overlap) {
    // The interior is taken to whole TILES of the cell order along x (4 columns when x is the middle axis, 2 when it is the outer
    // one; device.h cell_lin): 64 consecutive sorted particles are then all interior or all rest except where a run of tiles ends,
    // so a split launch does not run its waves half empty (measured on the 32 x 252 x 252 ring, profiles/r06_overlap.md).  Rounding
    // inwards keeps every interior particle at least two columns from a ghost.  (Also measured, and taken out again: a per-block
    // flag word that lets a workgroup without work leave on one scalar load -- it saved 35 us per split step and cost the
    // UNsplit developed step 40 us: the early exit in the passes' prologue perturbs their code, k_drho_div_iter_vort +8 %.)
    const int T = d.perm ? (1 << T1B) : (1 << T0B);
    if (R.ilo > d.x0) R.ilo = d.x0 + ((R.ilo - d.x0 + T - 1) / T) * T;
    if (R.ihi < d.x0 + d.nx) R.ihi = d.x0 + ((R.ihi - d.x0) / T) * T;
  }
  // ---- compute_density(); compute_dfsph_coff(); and the first pass of solve_vel_divergence() (dfsph.py:131-146): they read
  // pos_j / vel_j only, which the ghost rows brought with them
  PASS(k_density_coff_warm, ctx->dt);
  ctx->dv_iter = 0;
  // Halo refreshes of the step's opening (round 6: two where there were three).  The warm-start velocity pass (dfsph.py:420-439)
  // needs the neighbours' rescaled kappa_v: the ghost rows formed it themselves (k_ghost_records) -- no exchange.  rho_j (the field
  // and the (pos, rho) records) is first read by the last divergence iteration: it rides on the velocity refresh that
  // begin_divergence_iter needs anyway.
  SRC(SPASS_(k_vel_from_k<1>, (const float4 *)d.pk, ctx->dt)());      // warm-start velocities; interior and boundary alike: no ghost field is stale
  const float dt = ctx->dt;
  R.hf_begin(); R.hf_add(d.vel, 3, 4); R.hf_add(d.rho, 1, 1);
  {
    HaloFields also = R.h;                    // the rho column once more: -> pr.w
    also.ptr[0] = nullptr;
    also.ptr[1] = (const float *)d.pr + 3; also.stride[1] = 4;
    SRC(R.xp(true, SPASS_((k_drho_div<DRHO_BEGIN, true>), dt), &also));
  }
  ctx->pr_valid = true;                       // (pos, rho) records of every row: own rows by the density pass, ghosts just now

  // ---- the pieces of the step; every launch honours ctx->d.pred (a gated piece does nothing when its predicate is 0), the
  // exchanges in between are unconditional: a gated-off exchange moves a stale buffer that nobody unpacks
  bool vort_taken = false;                    // a k_drho_div_iter_vort ran: the vorticity sums are in d_omega / cg_s
  auto div_iter = [&]() -> int {              // divergence_iter (dfsph.py:449-477); the vorticity sums ride on it (k_drho_div_iter_vort)
    int rc;
    R.hf_begin(); R.hf_add((const float *)d.pk + 3, 1, 4);           // k_j = b_j alpha_j, written by the density-change pass
    if ((rc = R.xp(true, SPASS_(k_vel_from_k<0>, (const float4 *)d.pk, dt, d.kappa_v)))) return rc;
    R.hf_begin(); R.hf_add(d.vel, 3, 4);
    if ((rc = R.xp(true, SPASS_(k_drho_div_iter_vort, dt)))) return rc;
    return R.allreduce(ACC_ERR);
  };
  auto visc_open = [&](bool vort) -> int {    // end of solve_vel_divergence + compute_nonpressure_force() up to the sums of its first PCG iteration (dfsph.py:84-97)
    int rc;
    ELEM(k_end_div_begin_nonpressure, dt, vort ? 1 : 0);
    if (tension) {
      RPASS(k_tension_normal);
      R.hf_begin(); R.hf_add(d.normal, 3, 4);
      if ((rc = R.xp(true, SPASS_(k_tension_accel)))) return rc;
    }
    // (no refresh of vel_guess_j: k_end_div_begin_nonpressure advanced it on the ghost rows too -- from the vel_guess they
    // brought with them and the velocities the last refresh gave them, the owner's own arithmetic on the owner's own bits;
    // round 6: one exchange per step less)
    if (vort) { if ((rc = SPASS_(k_visc_minv_residual, dt)())) return rc; }
    else if ((rc = SPASS_(k_visc_minv_residual_vort, dt)())) return rc;
    R.hf_begin(); R.hf_add(d.cg_dir, 3, 4);
    if ((rc = R.xp(true, SPASS_(k_visc_Ad, dt)))) return rc;
    // delta_zero (first read by k_visc_update) and dAd in ONE all-reduce where the communicator can sum a range of slots
    if ((rc = R.allreduce_range(ACC_DAD, ACC_DZERO))) return rc;
    ELEM(k_visc_update, 0.0f, 1);
    return R.allreduce(ACC_DNEW);
  };
  const bool cfl_here = !(d.faithful && d.nl <= 1);
  // end_viscosity + the integration loop of compute_vorticity + this rank's CFL maximum -> flags[FLAG_VMAX], made global by
  // the caller's MAX all-reduce (device_dt: and the time step formed from it on the device, k_dt_from_vmax)
  auto cfl_device = [&](bool device_dt) -> int {
    if (cfl_here) {
      ELEM(k_finish_nonpressure, dt);
      hipLaunchKernelGGL(k_cfl_store, dim3(1), dim3(WAVE), 0, ctx->stream, ctx->d, DtRule{nullptr, 0.f, 0, 0.0, 0.0, 0.0, 0.0});
    } else {
      // (dfsph.py:108-111 with a single liquid particle: the reference launches nothing, vel_max[0] keeps its 0; every rank
      // still makes the collective call)
      ELEM(k_visc_end, dt); ELEM(k_vorticity_integrate, dt);
      HIPCHK(hipMemsetAsync(d.flags + FLAG_VMAX, 0, sizeof(int), ctx->stream));
    }
    LAUNCHCHK();
    int rc;
    if (R.multi && (rc = cb->allreduce_max_bits(cb->user, d.flags + FLAG_VMAX))) return ctx->fail(rc, "slab callback allreduce_max_bits failed");
    if (device_dt) {
      const int it = ctx->pr_iter > 1 ? ctx->pr_iter : 1;            // max(vs_iter = 1 on this branch, last step's pr_iter), Q6
      hipLaunchKernelGGL(k_dt_from_vmax, dim3(1), dim3(WAVE), 0, ctx->stream, ctx->d,
                         DtRule{ctx->dt_dev, ctx->dt, it, (double)ctx->p.eps, ctx->p.cfl_radius, ctx->p.user_max_t, ctx->p.user_min_t});
      LAUNCHCHK();
    }
    return 0;
  };
  auto press_iter = [&](float dtv) -> int {   // pressure_iter (dfsph.py:518-547)
    int rc;
    const bool prof = ctx->prof_on && ctx->prof_used + 2 <= ctx->prof_ev.size();
    if (prof) HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
    R.hf_begin(); R.hf_add((const float *)d.pk + 3, 1, 4);
    if ((rc = R.xp(true, SPASS_(k_vel_from_k<0>, (const float4 *)d.pk, dtv, d.kappa)))) return rc;
    R.hf_begin(); R.hf_add(d.vel, 3, 4);
    if ((rc = R.xp(true, SPASS_((k_drho_press<DRHO_ITER, true>), dtv)))) return rc;
    if (prof) { HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], ctx->stream)); ctx->prof_used += 2; }
    return 0;
  };
  auto press_open = [&](float dtv) -> int {   // update_vel(); solve_pressure() up to its first decision (dfsph.py:150-161 forces two iterations; the warm start changes no velocity, Q7)
    int rc;
    ELEM(k_update_vel_warm_scale, dtv, 0);
    PASS(k_warm_press_vel, dtv);
    R.hf_begin(); R.hf_add(d.vel, 3, 4);
    if ((rc = R.xp(true, SPASS_((k_drho_press<DRHO_BEGIN, true>), dtv)))) return rc;
    if ((rc = press_iter(dtv)) || (rc = press_iter(dtv))) return rc;
    return R.allreduce(ACC_ERR);
  };
  auto host_dt_rule = [&](double vmax) {      // optimize_time_step() (dfsph.py:113-129)
    if ((double)(float)vmax > (double)ctx->p.eps) {
      double time_step = 0.5 * 0.4 * ctx->p.cfl_radius * 2.0 / sqrt((double)(float)vmax);
      if (time_step > ctx->p.user_max_t) time_step = ctx->p.user_max_t;
      if (time_step < ctx->p.user_min_t) time_step = ctx->p.user_min_t;
      const int it = ctx->vs_iter > ctx->pr_iter ? ctx->vs_iter : ctx->pr_iter;  // max(vs, max(pr, vs)), Q6
      float d2 = ctx->dt;
      if (it > 10) d2 = (float)((double)d2 * 0.9);
      else if (it < 5) d2 = (float)((double)d2 * 1.1);
      if ((double)d2 > time_step) d2 = (float)time_step;
      ctx->dt = d2;
    }
  };

  // ---- Speculation (as wcsph_dfsph_step): every solver decision is taken ON THE DEVICE by the k_publish that mails
  // its (all-reduced, hence rank-identical) error sum, and the common-case continuation -- one divergence iteration, one PCG
  // iteration, two pressure iterations -- is queued behind it in gated segments: [A: divergence] -> [end of it + first PCG
  // iteration, B] -> [rest of compute_nonpressure_force + CFL + deltaT on the device, C] -> [update_vel + pressure solve up to its
  // first decision, D] -> [end of step].  The GPU never waits for the host in that case; a decision that falls the other way
  // leaves the gated launches as no-ops (their exchanges move stale buffers nobody unpacks) and the host goes on from there one
  // decision at a time.  Needs the device-side MAX all-reduce; the surface-tension passes keep the plain control flow.
  const double thr_div = 0.001 * N / (double)dt;
  enum { ST_DIV, ST_VISC, ST_CFL, ST_PRESS, ST_END } stage = ST_DIV;
  double err = -0.1;
  const bool spec = (flags & 4) == 0 && ctx->speculate && !tension && cfl_here && (!R.multi || cb->allreduce_max_bits) &&
                    (double)ctx->avg_density_err > err;
  float dtn = dt;
  if (spec) {
    int *pA = ctx->pred_dev + 1, *pB = ctx->pred_dev + 2, *pD = ctx->pred_dev + 0;
    SRC(div_iter());
    SRC(publish_async(ctx, {PRED_DIV, thr_div, 1.0, pA, nullptr}));
    const unsigned long long seqA = ctx->mail_seq;
    d.pred = pA;
    SRC(visc_open(true));
    d.pred = nullptr;
    SRC(publish_async(ctx, {PRED_VISC, (double)ctx->p.viscosity_err, (double)ctx->p.eps, pB, pA}));
    const unsigned long long seqB = ctx->mail_seq;
    // the host blocks here, the GPU does not: the viscosity walks are queued behind A (profiles/r03_experiments.md (14): queueing
    // the rest up front costs more than it hides whenever the divergence solve goes on)
    SRC(collect_mail(ctx, true, seqA));
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    err = thr_div;
    ctx->dv_iter = 1; vort_taken = true;
    if (ctx->h_pred) {
      d.pred = pB;
      SRC(cfl_device(true));
      d.pred = nullptr;
      SRC(publish_async(ctx, {PRED_PASS, 0.0, 1.0, nullptr, pB}));
      const unsigned long long seqC = ctx->mail_seq;
      const size_t prof_mark = ctx->prof_used;
      d.pred = pB; d.dtp = ctx->dt_dev;
      SRC(press_open(dt));
      d.pred = nullptr; d.dtp = nullptr;
      SRC(publish_async(ctx, {PRED_PRESS, 0.001, N, pD, pB}));
      const unsigned long long seqD = ctx->mail_seq;
      d.pred = pD; d.dtp = ctx->dt_dev;
      ELEM(k_end_press_update_pos, dt);
      d.pred = nullptr; d.dtp = nullptr;
      // the next step's migration / ghost classification on the positions just written, behind the same gate, and one more mail
      // (E) behind it: when the step ends here -- the common case -- its counts are on the host without a wait of their own
      unsigned long long seqE = 0;
      if (ctx->tc.on && R.multi) {
        SRC(tail_classify(ctx, pD, tail_native));
        SRC(publish_async(ctx, {PRED_PASS, 0.0, 1.0, nullptr, pD}));
        seqE = ctx->mail_seq;
      }
      SRC(collect_mail(ctx, true, seqB));
      ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
      ctx->cg_delta_old = ctx->cg_delta_zero;
      ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
      ctx->vs_iter = 1;
      if (ctx->h_pred) {
        SRC(collect_mail(ctx, true, seqC));
        memcpy(&ctx->vel_max0, &ctx->h_flags[FLAG_VMAX], 4);
        { float host_dt_before = ctx->dt; host_dt_rule((double)ctx->vel_max0);   // the host rule on the same inputs: compared, counted
          float ddt; memcpy(&ddt, &ctx->h_flags[FLAG_DT], 4);
          if (memcmp(&ddt, &ctx->dt, 4) != 0) ctx->n_dt_mismatch++;
          ctx->dt = host_dt_before; }
        memcpy(&ctx->dt, &ctx->h_flags[FLAG_DT], 4);       // the device's deltaT is the one the queued kernels ran with: adopted
        dtn = ctx->dt;
        SRC(collect_mail(ctx, true, seqD));
        R.n_wait++;                                          // the one wait of the step: its last mail
        ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
        err = (double)ctx->avg_density_err / N;
        ctx->pr_iter = 2;
        stage = ctx->h_pred ? ST_END : ST_PRESS;             // ST_PRESS: go on from iteration 3
        if (stage == ST_END) {
          if (seqE) {   // (queued right behind D: it has arrived, or is about to -- the GPU is not idle meanwhile)
            const float err_keep = ctx->avg_density_err;
            SRC(collect_mail(ctx, true, seqE));
            ctx->avg_density_err = err_keep;
            ctx->tc.valid = true; ctx->tc.nbr_valid = ctx->tc.nbr_queued;
          }
          if (stats) { stats[0] = R.n_xchg; stats[1] = R.n_wait; }
          return WCSPH_OK;
        }
      } else { ctx->prof_used = prof_mark; stage = ST_VISC; R.n_wait++; }   // the PCG goes on; everything behind B did nothing
    } else { stage = ST_DIV; R.n_wait++; }                   // the divergence solve goes on; everything behind A did nothing
  }
  // ---- the plain control flow: one host wait per decision (also where a speculated decision fell the other way)
  if (stage == ST_DIV) {
    while ((double)ctx->avg_density_err > err && ctx->dv_iter < 10) {
      SRC(div_iter());
      SRC(R.fetch());
      ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
      err = thr_div;
      ctx->dv_iter++; vort_taken = true;
    }
    // ---- compute_nonpressure_force() (dfsph.py:84-103)
    SRC(visc_open(vort_taken));
    SRC(R.fetch());
    ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
    ctx->cg_delta_old = ctx->cg_delta_zero;
    ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
    ctx->vs_iter = 1;
    stage = ST_VISC;
  }
  if (stage == ST_VISC) {
    while (ctx->vs_iter < 100) {
      if ((double)ctx->cg_delta <= (double)ctx->p.viscosity_err * (double)ctx->cg_delta_zero || (double)ctx->cg_delta_zero < (double)ctx->p.eps) break;
      ELEM(k_visc_dir, ctx->cg_delta_old);        // beta = delta' (global, on the device) / delta the last iteration started with
      R.hf_begin(); R.hf_add(d.cg_dir, 3, 4);
      SRC(R.xp(true, SPASS_(k_visc_Ad, dt)));
      SRC(R.allreduce(ACC_DAD));
      ELEM(k_visc_update, ctx->cg_delta);
      SRC(R.allreduce(ACC_DNEW));
      SRC(R.fetch());
      ctx->cg_delta_old = ctx->cg_delta;
      ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
      ctx->vs_iter++;
    }
    // ---- optimize_time_step() (dfsph.py:107-129) on the GLOBAL velocity maximum
    double vmax;
    if (R.multi && cb->allreduce_max_bits) {
      // the maximum over the ranks is formed on the device (a MAX all-reduce of the bit pattern, queued behind k_cfl_store) and
      // read with the one wait this decision needs anyway
      SRC(cfl_device(false));
      SRC(pull_scalars(ctx));
      memcpy(&ctx->vel_max0, &ctx->h_flags[FLAG_VMAX], 4);
      R.n_wait++;
      vmax = (double)ctx->vel_max0;
    } else {
      if (cfl_here) { ELEM(k_finish_nonpressure, dt); ctx->cfl_done = true; }
      else { ELEM(k_visc_end, dt); ELEM(k_vorticity_integrate, dt); }
      SRC(wcsph_dfsph_cfl_reduce(ctx));
      R.n_wait++;
      vmax = (double)ctx->vel_max0;
      if (R.multi && (rc = cb->allreduce_max(cb->user, &vmax))) return ctx->fail(rc, "slab callback allreduce_max failed");
    }
    host_dt_rule(vmax);
    dtn = ctx->dt;
    SRC(press_open(dtn));
    SRC(R.fetch());
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    err = (double)ctx->avg_density_err / N;
    ctx->pr_iter = 2;
    stage = ST_PRESS;
  }
  // ---- solve_pressure() from its third iteration on (dfsph.py:160-163)
  while (err > 0.001 && ctx->pr_iter < 100) {
    SRC(press_iter(dtn));
    ctx->pr_iter++;
    SRC(R.allreduce(ACC_ERR));
    SRC(R.fetch());
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    err = (double)ctx->avg_density_err / N;
  }
  ELEM(k_end_press_update_pos, dtn);
  // the same tail on the plain path (every rank is here or none: the decisions were taken on all-reduced sums): the classification
  // and, on the native plane, the neighbours' counts are queued now, their mail is waited for by whoever asks for the result
  // (wcsph_slab_tail_classify_result) -- the migration's message sizes are then known without a header round trip on this path too
  if (ctx->tc.on && R.multi) {
    SRC(tail_classify(ctx, nullptr, tail_native));
    SRC(publish_async(ctx));
    ctx->tc.pending_seq = ctx->mail_seq;
  }
  if (stats) { stats[0] = R.n_xchg; stats[1] = R.n_wait; }
  return WCSPH_OK;
}
#undef SPASS_
#undef SRC


}  // extern "C"
