// api_fields.h -- field transfer in the caller's original particle order (ref: the Taichi field protocol, dfsph.py:113,129,141,162), scalars, iteration counters.
#pragma once
extern "C" {

// ---- field transfer ---------------------------------------------------------------------------
struct FieldDesc { int kind; /*3 vec, 1 scalar, 9 mat, -1 int*/ void *ptr; };
static FieldDesc field_desc(wcsph_ctx *ctx, int f) {
  Dev &d = ctx->d;
  switch (f) {
    case WCSPH_F_POS: return {3, d.pos};
    case WCSPH_F_VEL: return {3, d.vel};
    case WCSPH_F_VEL_GUESS: return {3, d.vel_guess};
    case WCSPH_F_D_VEL: return {3, d.d_vel};
    case WCSPH_F_OMEGA: return {3, d.omega};
    case WCSPH_F_D_OMEGA: return {3, d.d_omega};
    case WCSPH_F_NORMAL: return {3, d.normal};
    case WCSPH_F_CG_R: return {3, d.cg_r};
    case WCSPH_F_CG_DIR: return {3, d.cg_dir};
    case WCSPH_F_CG_AD: return {3, d.cg_Ad};
    case WCSPH_F_CG_S: return {3, d.cg_s};
    case WCSPH_F_RHO: return {1, d.rho};
    case WCSPH_F_ADV_RHO: return {1, d.adv_rho};
    case WCSPH_F_PRESSURE: return {1, d.pressure};
    case WCSPH_F_VEL_MAX: return {1, d.vel_max};
    case WCSPH_F_ALPHA: return {1, d.alpha};
    case WCSPH_F_KAPPA: return {1, d.kappa};
    case WCSPH_F_KAPPA_V: return {1, d.kappa_v};
    case WCSPH_F_CG_MINV: return {9, nullptr};
    case WCSPH_F_NEIGHBOR_COUNT: return {-1, d.ncount};
    case WCSPH_F_KBUF: return {1, d.kbuf};
    case WCSPH_F_D_VEL_PRE: return {3, d.d_vel_pre};
    case WCSPH_F_POS_STAR: return {3, d.pos_star};
    case WCSPH_F_VEL_STAR: return {3, d.vel_star};
    case WCSPH_F_D_II: return {3, d.d_ii};
    case WCSPH_F_DIJ_PJ: return {3, d.dij_pj};
    case WCSPH_F_A_II: return {1, d.a_ii};
    case WCSPH_F_PRESSURE_PRE: return {1, d.pressure_pre};
    case WCSPH_F_COLOR: return {ctx->surf.color ? 1 : 0, ctx->surf.color};
    case WCSPH_F_COLOR_GRAD: return {ctx->surf.color ? 3 : 0, ctx->surf.color_grad};
    case WCSPH_F_POS_AVR: return {ctx->surf.color ? 3 : 0, ctx->surf.pos_avr};
    case WCSPH_F_G: return {ctx->surf.color ? 10 : 0, nullptr};
  }
  return {0, nullptr};
}

int wcsph_field_to_host(wcsph_ctx *ctx, int field, void *dst, size_t bytes) {
  NEED_COMPACT();
  if (!dst) return WCSPH_EARG;
  Dev &d = ctx->d;
  FieldDesc fd = field_desc(ctx, field);
  if (!fd.kind) return ctx->fail(WCSPH_EARG, "unknown field id (or a surface field before the surface kernels ran)");
  hipStream_t s = ctx->stream;
  const int nl = d.nl;
  size_t rows = (field == WCSPH_F_POS) ? (size_t)nl + d.ns : (size_t)nl;
  size_t per = fd.kind == 3 ? 3 : (fd.kind == 9 || fd.kind == 10) ? 9 : 1;
  if (bytes != rows * per * 4) return ctx->fail(WCSPH_EARG, "wcsph_field_to_host: wrong byte count");
  if (nl > 0) {
    if (fd.kind == 3) hipLaunchKernelGGL(k_unpermute3, dim3(nblk(nl)), dim3(BLOCK), 0, s, (const float4 *)fd.ptr, (const int *)d.oid, nl, ctx->stage);
    else if (fd.kind == 1 || fd.kind == -1)
      hipLaunchKernelGGL(k_unpermute1, dim3(nblk(nl)), dim3(BLOCK), 0, s, (const float *)fd.ptr, (const int *)d.oid, nl, ctx->stage);
    else if (fd.kind == 9) {  // cg_Minv: six stored floats per particle -> the symmetric [n][3][3]
      std::vector<float4> a(nl);
      std::vector<float2> b(nl);
      std::vector<int> oid(nl);
      HIPCHK(hipMemcpyAsync(a.data(), d.minv_a, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(b.data(), d.minv_b, (size_t)nl * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(oid.data(), d.oid, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      float *o = (float *)dst;
      for (int i = 0; i < nl; i++) {
        float *m = o + (size_t)oid[i] * 9;
        m[0] = a[i].x; m[1] = a[i].y; m[2] = a[i].z; m[3] = a[i].y; m[4] = a[i].w; m[5] = b[i].x; m[6] = a[i].z; m[7] = b[i].x; m[8] = b[i].y;
      }
      return WCSPH_OK;
    } else {  // G rows -> [n][3][3]
      std::vector<float4> r0(nl), r1(nl), r2(nl);
      std::vector<int> oid(nl);
      HIPCHK(hipMemcpyAsync(r0.data(), ctx->surf.g0, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(r1.data(), ctx->surf.g1, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(r2.data(), ctx->surf.g2, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(oid.data(), d.oid, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      float *o = (float *)dst;
      for (int i = 0; i < nl; i++) {
        float *m = o + (size_t)oid[i] * 9;
        m[0] = r0[i].x; m[1] = r0[i].y; m[2] = r0[i].z; m[3] = r1[i].x; m[4] = r1[i].y; m[5] = r1[i].z;
        m[6] = r2[i].x; m[7] = r2[i].y; m[8] = r2[i].z;
      }
      return WCSPH_OK;
    }
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(dst, ctx->stage, (size_t)nl * per * 4, hipMemcpyDeviceToHost, s));
  }
  // a transfer is a synchronisation point: error flags raised since the last one travel with it (the data is delivered
  // either way; the return code says whether an overflow / non-finite condition was raised while it was produced)
  int frc = publish_async(ctx);
  if (frc) return frc;
  HIPCHK(hipStreamSynchronize(s));
  if (field == WCSPH_F_POS && d.ns > 0) memcpy((float *)dst + (size_t)nl * 3, ctx->solid_host.data(), (size_t)d.ns * 12);
  return collect_mail(ctx, false);
}

int wcsph_field_from_host(wcsph_ctx *ctx, int field, const void *src, size_t bytes) {
  NEED_COMPACT();
  if (!src) return WCSPH_EARG;
  Dev &d = ctx->d;
  FieldDesc fd = field_desc(ctx, field);
  if (!fd.kind || fd.kind == 9 || fd.kind == 10 || fd.kind == -1) return ctx->fail(WCSPH_EARG, "field is not writable from the host");
  hipStream_t s = ctx->stream;
  const int nl = d.nl;
  size_t per = fd.kind == 3 ? 3 : 1;
  size_t rows_full = (field == WCSPH_F_POS) ? (size_t)nl + d.ns : (size_t)nl;
  if (bytes != rows_full * per * 4 && bytes != (size_t)nl * per * 4) return ctx->fail(WCSPH_EARG, "wcsph_field_from_host: wrong byte count");
  if (nl == 0) return WCSPH_OK;
  HIPCHK(hipMemcpyAsync(ctx->stage, src, (size_t)nl * per * 4, hipMemcpyHostToDevice, s));
  if (fd.kind == 3) hipLaunchKernelGGL(k_permute3, dim3(nblk(nl)), dim3(BLOCK), 0, s, (float4 *)fd.ptr, (const int *)d.where, nl, (const float *)ctx->stage);
  else hipLaunchKernelGGL(k_permute1, dim3(nblk(nl)), dim3(BLOCK), 0, s, (float *)fd.ptr, (const int *)d.where, nl, (const float *)ctx->stage);
  LAUNCHCHK();
  HIPCHK(hipStreamSynchronize(s));
  if (field == WCSPH_F_POS) ctx->grid_valid = ctx->hist_valid = false;  // cells are stale until the next grid update
  ctx->pr_valid = false;
  return WCSPH_OK;
}

// rows [i0, i0 + n) of a field in the caller's original order (Field.__getitem__ / __setitem__: one element, not the field)
__global__ void k_rows_out(const float *f, int stride, int per, const int *where, int i0, int n, float *out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * per) return;
  out[t] = f[(size_t)where[i0 + t / per] * stride + t % per];
}
__global__ void k_rows_in(float *f, int stride, int per, const int *where, int i0, int n, const float *in) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * per) return;
  f[(size_t)where[i0 + t / per] * stride + t % per] = in[t];
}
int wcsph_field_rows_to_host(wcsph_ctx *ctx, int field, int32_t i0, int32_t n, void *dst) {
  NEED_COMPACT();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1 && fd.kind != -1) return ctx->fail(WCSPH_EARG, "field_rows: vec3 / scalar fields only");
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || !dst) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  const int per = fd.kind == 3 ? 3 : 1, stride = fd.kind == 3 ? 4 : 1;
  if ((size_t)n * per > ((size_t)ctx->capacity + ctx->d.ns) * 3) return WCSPH_EARG;
  hipLaunchKernelGGL(k_rows_out, dim3(nblk(n * per)), dim3(BLOCK), 0, ctx->stream, (const float *)fd.ptr, stride, per, (const int *)ctx->d.where, i0, n, ctx->stage);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(dst, ctx->stage, (size_t)n * per * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return WCSPH_OK;
}
int wcsph_field_rows_from_host(wcsph_ctx *ctx, int field, int32_t i0, int32_t n, const void *src) {
  NEED_COMPACT();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "field_rows: writable vec3 / scalar fields only");
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || !src) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  const int per = fd.kind == 3 ? 3 : 1, stride = fd.kind == 3 ? 4 : 1;
  HIPCHK(hipMemcpyAsync(ctx->stage, src, (size_t)n * per * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rows_in, dim3(nblk(n * per)), dim3(BLOCK), 0, ctx->stream, (float *)fd.ptr, stride, per, (const int *)ctx->d.where, i0, n, (const float *)ctx->stage);
  LAUNCHCHK();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (field == WCSPH_F_POS) ctx->grid_valid = ctx->hist_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}

int wcsph_field_gather_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, void *d_dst) {
  NEED_COMPACT();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "gather_rows: vec3 / scalar fields only");
  if (n <= 0) return WCSPH_OK;
  if (fd.kind == 3) hipLaunchKernelGGL(k_gather_rows3, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (const float4 *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (float *)d_dst);
  else hipLaunchKernelGGL(k_gather_rows1, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (const float *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (float *)d_dst);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_field_scatter_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, const void *d_src) {
  NEED_COMPACT();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "scatter_rows: vec3 / scalar fields only");
  if (n <= 0) return WCSPH_OK;
  if (fd.kind == 3) hipLaunchKernelGGL(k_scatter_rows3, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (float4 *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (const float *)d_src);
  else hipLaunchKernelGGL(k_scatter_rows1, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (float *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (const float *)d_src);
  LAUNCHCHK();
  if (field == WCSPH_F_POS) ctx->grid_valid = ctx->hist_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}
int wcsph_set_owned(wcsph_ctx *ctx, int32_t n_owned) {
  if (!ctx || n_owned < 0 || n_owned > ctx->d.nl) return WCSPH_EARG;
  ctx->d.n_owned = n_owned;
  ctx->hist_valid = false;   // which rows the bucket table counts has changed
  return WCSPH_OK;
}

int wcsph_get_scalar(wcsph_ctx *ctx, int id, float *v) {
  if (!ctx || !v) return WCSPH_EARG;
  switch (id) {
    case WCSPH_S_DELTA_T: *v = ctx->dt; break;
    case WCSPH_S_AVG_DENSITY_ERR: *v = ctx->avg_density_err; break;
    case WCSPH_S_CG_DELTA: *v = ctx->cg_delta; break;
    case WCSPH_S_CG_DELTA_OLD: *v = ctx->cg_delta_old; break;
    case WCSPH_S_CG_DELTA_ZERO: *v = ctx->cg_delta_zero; break;
    case WCSPH_S_VISCOSITY: *v = ctx->d.visc; break;
    case WCSPH_S_VISCOSITY_B: *v = ctx->d.visc_b; break;
    case WCSPH_S_TENSION: *v = ctx->d.tension; break;
    case WCSPH_S_TENSION_B: *v = ctx->d.tension_b; break;
    case WCSPH_S_VEL_MAX0: *v = ctx->vel_max0; break;
    case WCSPH_S_RHO_ERR: *v = ctx->rho_err; break;
    case WCSPH_S_TABLE_UPDATES_IN_PLACE: *v = (float)ctx->n_hist_inc; break;
    case WCSPH_S_DT_RULE_MISMATCHES: *v = (float)ctx->n_dt_mismatch; break;
    default: return ctx->fail(WCSPH_EARG, "unknown scalar id");
  }
  return WCSPH_OK;
}
int wcsph_set_scalar(wcsph_ctx *ctx, int id, float v) {
  if (!ctx) return WCSPH_EARG;
  switch (id) {
    case WCSPH_S_DELTA_T: ctx->dt = v; break;
    case WCSPH_S_AVG_DENSITY_ERR: ctx->avg_density_err = v; break;
    case WCSPH_S_CG_DELTA: ctx->cg_delta = v; break;
    case WCSPH_S_CG_DELTA_OLD: ctx->cg_delta_old = v; break;
    case WCSPH_S_CG_DELTA_ZERO: ctx->cg_delta_zero = v; break;
    case WCSPH_S_VISCOSITY: ctx->d.visc = v; break;
    case WCSPH_S_VISCOSITY_B: ctx->d.visc_b = v; break;
    case WCSPH_S_TENSION: ctx->d.tension = v; break;
    case WCSPH_S_TENSION_B: ctx->d.tension_b = v; break;
    default: return ctx->fail(WCSPH_EARG, "scalar is read-only or unknown");
  }
  return WCSPH_OK;
}
int wcsph_get_iters(wcsph_ctx *ctx, int32_t *vs, int32_t *dv, int32_t *pr) {
  if (!ctx) return WCSPH_EARG;
  if (vs) *vs = ctx->vs_iter;
  if (dv) *dv = ctx->dv_iter;
  if (pr) *pr = ctx->pr_iter;
  return WCSPH_OK;
}
int wcsph_set_iters(wcsph_ctx *ctx, int32_t vs, int32_t dv, int32_t pr) {
  if (!ctx) return WCSPH_EARG;
  ctx->vs_iter = vs; ctx->dv_iter = dv; ctx->pr_iter = pr;
  return WCSPH_OK;
}


}  // extern "C"
