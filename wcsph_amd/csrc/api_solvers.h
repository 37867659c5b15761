// api_solvers.h -- sesph.py, pcisph.py and iisph.py behind the same boundary (ref: sesph.py:130-236, pcisph.py:147-327, iisph.py:114-441).
#pragma once
extern "C" {

// ---- sesph.py ---------------------------------------------------------------------------------
static Dev sesph_view(const wcsph_ctx *ctx) {
  Dev s = ctx->d;  // same arrays, the script's private kernel constants (sesph.py:40-45, pi = 3.1415926)
  s.h = ctx->p.ss_searchR; s.h2 = s.h * s.h; s.inv_h = 1.0f / s.h;
  s.mk_h3 = ctx->p.ss_mk; s.ml_h3 = ctx->p.ss_ml;
  return s;
}
#define SPASS(kern, ...)                                                                          \
  do {                                                                                            \
    Dev sv_ = sesph_view(ctx);                                                                    \
    if (sv_.nl > 0) hipLaunchKernelGGL(kern, dim3(nblk8(sv_.nl)), dim3(BLOCK), 0, ctx->stream, sv_, ##__VA_ARGS__); \
    LAUNCHCHK();                                                                                  \
  } while (0)
int wcsph_sesph_reset_param(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_ss_reset); ctx->dt = 0.001f; return WCSPH_OK; }
int wcsph_sesph_density(wcsph_ctx *ctx) { NEED_GRID(); SPASS(k_ss_density); ctx->pr_valid = false; return WCSPH_OK; }
int wcsph_sesph_pressure(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_ss_pressure); return WCSPH_OK; }
int wcsph_sesph_force(wcsph_ctx *ctx) { NEED_GRID(); SPASS(k_ss_force); return WCSPH_OK; }
int wcsph_sesph_integrate(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_ss_integrate, ctx->dt); return WCSPH_OK; }
int wcsph_sesph_step(wcsph_ctx *ctx) {
  NEED_SETUP();
  int rc;
  if ((rc = wcsph_grid_update(ctx))) return rc;
  if ((rc = wcsph_sesph_density(ctx))) return rc;
  if ((rc = wcsph_sesph_pressure(ctx))) return rc;
  if ((rc = wcsph_sesph_force(ctx))) return rc;
  return wcsph_sesph_integrate(ctx);
}

// ---- pcisph.py ----------------------------------------------------------------------------------
int wcsph_pcisph_reset_param(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_pc_reset); ctx->dt = 0.001f; return WCSPH_OK; }
int wcsph_pcisph_compute_nonpressure_force(wcsph_ctx *ctx) {
  NEED_GRID();
  SPASS(k_ss_density);   // rho_i = (VL0 (W0 + sum) + VS0 sum_s) rho_0, pcisph.py:203,212,215
  ctx->pr_valid = false;
  SPASS(k_pc_force, ctx->p.pc_visc, ctx->p.pc_visc_b);
  return WCSPH_OK;
}
int wcsph_pcisph_init_iter_info(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_pc_init_iter); return WCSPH_OK; }
int wcsph_pcisph_update_iter_info(wcsph_ctx *ctx) {
  NEED_SETUP();
  ELEM(k_pc_update_iter, ctx->dt);
  ctx->rho_err = 0.0f;
  return WCSPH_OK;
}
int wcsph_pcisph_predict_density(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = zero_acc(ctx, ACC_ERR);
  if (rc) return rc;
  SPASS(k_pc_predict, ctx->dt, ctx->p.pci_coff);
  SPASS(k_pc_pressure_force);
  if ((rc = pull_scalars(ctx))) return rc;
  ctx->rho_err = (float)((double)ctx->rho_err + ctx->h_acc[ACC_ERR]);
  return WCSPH_OK;
}
int wcsph_pcisph_update_pos(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_pc_update_pos, ctx->dt); return WCSPH_OK; }
int wcsph_pcisph_solve_pressure(wcsph_ctx *ctx) {  // ref: pcisph.py:147-157
  NEED_GRID();
  int rc;
  ctx->pr_iter = 0;
  double err = 0.0;
  if ((rc = wcsph_pcisph_init_iter_info(ctx))) return rc;
  while ((err > 0.01 || ctx->pr_iter < 3) && ctx->pr_iter < 50) {
    if ((rc = wcsph_pcisph_update_iter_info(ctx))) return rc;
    if ((rc = wcsph_pcisph_predict_density(ctx))) return rc;
    err = (double)ctx->rho_err / (double)ctx->d.n_owned;
    ctx->pr_iter++;
  }
  return WCSPH_OK;
}
int wcsph_pcisph_step(wcsph_ctx *ctx) {
  NEED_SETUP();
  int rc;
  if ((rc = wcsph_grid_update(ctx))) return rc;
  if ((rc = wcsph_pcisph_compute_nonpressure_force(ctx))) return rc;
  if ((rc = wcsph_pcisph_solve_pressure(ctx))) return rc;
  return wcsph_pcisph_update_pos(ctx);
}

// ---- iisph.py -------------------------------------------------------------------------------------
static Dev iisph_view(const wcsph_ctx *ctx) {   // sesph kernel constants + the script's own viscosity setup
  Dev s = sesph_view(ctx);
  s.visc = ctx->p.ii_visc; s.visc_b = ctx->p.ii_visc_b; s.mass = ctx->p.ss_mass; s.rho_S0 = ctx->p.ss_rho0;
  s.VS0 = ctx->p.ss_VS0; s.dim_coff = ctx->p.ss_dim;
  return s;
}
#define IPASS(kern, ...)                                                                          \
  do {                                                                                            \
    Dev iv_ = iisph_view(ctx);                                                                    \
    if (!ctx->pr_valid && iv_.nl > 0) {                                                           \
      hipLaunchKernelGGL(k_pack_pr, dim3(nblk(iv_.nl)), dim3(BLOCK), 0, ctx->stream, iv_);        \
      ctx->pr_valid = true;                                                                       \
    }                                                                                             \
    if (iv_.nl > 0) hipLaunchKernelGGL(kern, dim3(nblk8(iv_.nl)), dim3(BLOCK), 0, ctx->stream, iv_, ##__VA_ARGS__); \
    LAUNCHCHK();                                                                                  \
  } while (0)
int wcsph_iisph_reset_param(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_ss_reset); ctx->dt = 0.001f; return WCSPH_OK; }
int wcsph_iisph_compute_density(wcsph_ctx *ctx) { NEED_GRID(); SPASS(k_ss_density); ctx->pr_valid = false; return WCSPH_OK; }
int wcsph_iisph_visc_init(wcsph_ctx *ctx) {
  NEED_GRID();
  ELEM(k_visc_guess);
  int rc = zero_acc(ctx, ACC_DZERO);
  if (rc) return rc;
  IPASS(k_visc_minv_residual, ctx->dt);
  if ((rc = pull_scalars(ctx))) return rc;
  ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
  ctx->cg_delta = ctx->cg_delta_zero;
  return WCSPH_OK;
}
int wcsph_iisph_visc_iter(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = zero_acc(ctx, ACC_DAD);
  if (rc) return rc;
  IPASS(k_visc_Ad, ctx->dt);
  if ((rc = zero_acc(ctx, ACC_DNEW))) return rc;
  ELEM(k_visc_update, ctx->cg_delta);
  ELEM(k_visc_dir, ctx->cg_delta);
  if ((rc = pull_scalars(ctx))) return rc;
  ctx->cg_delta_old = ctx->cg_delta;
  ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
  return WCSPH_OK;
}
int wcsph_iisph_combine_nonpressure(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_ii_combine, ctx->dt); return WCSPH_OK; }
int wcsph_iisph_compute_nonpressure_force(wcsph_ctx *ctx) {  // ref: iisph.py:114-127
  NEED_GRID();
  int rc;
  if ((rc = wcsph_iisph_visc_init(ctx))) return rc;
  ctx->vs_iter = 0;
  while (ctx->vs_iter < 100) {
    if ((rc = wcsph_iisph_visc_iter(ctx))) return rc;
    ctx->vs_iter++;
    if ((double)ctx->cg_delta <= (double)ctx->p.viscosity_err * (double)ctx->cg_delta_zero ||
        (double)ctx->cg_delta_zero < (double)ctx->p.eps)
      break;
  }
  return wcsph_iisph_combine_nonpressure(ctx);
}
int wcsph_iisph_compute_advection(wcsph_ctx *ctx) {
  NEED_GRID();
  SPASS(k_ii_dii, ctx->dt);
  SPASS(k_ii_aii, ctx->dt);
  return WCSPH_OK;
}
int wcsph_iisph_update_iter_info(wcsph_ctx *ctx) { NEED_GRID(); SPASS(k_ii_dijpj); ctx->avg_density_err = 0.0f; return WCSPH_OK; }
int wcsph_iisph_update_pressure_force(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = zero_acc(ctx, ACC_ERR);
  if (rc) return rc;
  SPASS(k_ii_pressure, ctx->dt, ctx->p.ii_omega);
  if ((rc = pull_scalars(ctx))) return rc;
  ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
  return WCSPH_OK;
}
int wcsph_iisph_solve_pressure(wcsph_ctx *ctx) {  // ref: iisph.py:130-139
  NEED_GRID();
  int rc;
  ctx->pr_iter = 0;
  double err = 0.0;
  while ((err > 0.001 || ctx->pr_iter < 2) && ctx->pr_iter < 100) {
    if ((rc = wcsph_iisph_update_iter_info(ctx))) return rc;
    if ((rc = wcsph_iisph_update_pressure_force(ctx))) return rc;
    err = (double)ctx->avg_density_err / (double)ctx->d.n_owned;
    ctx->pr_iter++;
  }
  return WCSPH_OK;
}
int wcsph_iisph_update_pos(wcsph_ctx *ctx) {
  NEED_GRID();
  SPASS(k_ii_pressure_accel);
  ELEM(k_ss_integrate, ctx->dt);
  return WCSPH_OK;
}
int wcsph_iisph_step(wcsph_ctx *ctx) {
  NEED_SETUP();
  int rc;
  if ((rc = wcsph_grid_update(ctx))) return rc;
  if ((rc = wcsph_iisph_compute_density(ctx))) return rc;
  if ((rc = wcsph_iisph_compute_nonpressure_force(ctx))) return rc;
  if ((rc = wcsph_iisph_compute_advection(ctx))) return rc;
  if ((rc = wcsph_iisph_solve_pressure(ctx))) return rc;
  return wcsph_iisph_update_pos(ctx);
}


}  // extern "C"
