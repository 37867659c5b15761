// api_dfsph.h -- one entry point per @ti.kernel of dfsph.py (ref: dfsph.py:167-580), the host loops of dfsph.py:84-164 and the fused step of :606-617.
#pragma once
extern "C" {

// ---- dfsph.py kernels -------------------------------------------------------------------------
#define PASS(kern, ...)                                                                           \
  do {                                                                                            \
    if (ctx->d.nl > 0) hipLaunchKernelGGL(kern, dim3(nblk8(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, ##__VA_ARGS__); \
    ctx->d.rev ^= 1;   /* the next neighbour pass walks the particles the other way round (device.h sorted_index) */ \
    LAUNCHCHK();                                                                                  \
    PROF_LAP(ctx, #kern);                                                                         \
  } while (0)
/* walks over the (pos, rho) records: packed first unless this context's own density pass just wrote them for every row */
#define RPASS(kern, ...)                                                                          \
  do {                                                                                            \
    if (!ctx->pr_valid && ctx->d.nl > 0) {                                                        \
      hipLaunchKernelGGL(k_pack_pr, dim3(nblk(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d);  \
      ctx->pr_valid = true;                                                                       \
    }                                                                                             \
    PASS(kern, ##__VA_ARGS__);                                                                    \
  } while (0)
#define ELEM(kern, ...)                                                                           \
  do {                                                                                            \
    if (ctx->d.nl > 0) hipLaunchKernelGGL(kern, dim3(nblk(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, ##__VA_ARGS__); \
    LAUNCHCHK();                                                                                  \
    PROF_LAP(ctx, #kern);                                                                         \
  } while (0)

#define HOT_DRHO_DIV(mode, dtv) PASS(k_drho_div<mode>, dtv)
#define HOT_DRHO_PRESS(mode, dtv) PASS(k_drho_press<mode>, dtv)
/* pack != 0: rebuild the (pos, k) gather records first (k from another array, or ghosts refreshed by a halo
 * exchange since k_make_k wrote them) */
#define HOT_VEL(cond, kbp, dtv, pack)                                                             \
  do {                                                                                            \
    if (pack) ELEM(k_pack_k, (const float *)(kbp));                                               \
    PASS(k_vel_from_k<cond>, (const float4 *)ctx->d.pk, dtv);                                     \
  } while (0)

int wcsph_dfsph_reset_param(wcsph_ctx *ctx) {
  NEED_SETUP();
  ELEM(k_reset);
  ctx->dt = 0.001f;  // ref: dfsph.py:178
  return WCSPH_OK;
}
int wcsph_dfsph_compute_density(wcsph_ctx *ctx) { NEED_GRID(); PASS(k_density); ctx->pr_valid = ctx->d.n_owned == ctx->d.nl; return WCSPH_OK; }
int wcsph_dfsph_compute_coff(wcsph_ctx *ctx) { NEED_GRID(); PASS(k_coff); return WCSPH_OK; }

int wcsph_dfsph_warmstart_divergence(wcsph_ctx *ctx) {
  NEED_GRID();
  HOT_DRHO_DIV(DRHO_WARM, ctx->dt);
  HOT_VEL(1, ctx->d.kappa_v, ctx->dt, 1);
  return WCSPH_OK;
}
int wcsph_dfsph_begin_divergence(wcsph_ctx *ctx) { NEED_GRID(); HOT_DRHO_DIV(DRHO_BEGIN, ctx->dt); return WCSPH_OK; }

// launches one iteration; the error sum lands in h_acc[ACC_ERR] after pull_scalars()
static int divergence_iter_async(wcsph_ctx *ctx) {
  ELEM(k_make_k, ctx->d.kappa_v, 0.0f);
  HOT_VEL(0, ctx->d.kbuf, ctx->dt, 0);
  int rc = zero_acc(ctx, ACC_ERR);
  if (rc) return rc;
  HOT_DRHO_DIV(DRHO_ITER, ctx->dt);
  return 0;
}
static int pressure_iter_async(wcsph_ctx *ctx) {
  ELEM(k_make_k, ctx->d.kappa, 1.0f);
  HOT_VEL(0, ctx->d.kbuf, ctx->dt, 0);
  int rc = zero_acc(ctx, ACC_ERR);
  if (rc) return rc;
  HOT_DRHO_PRESS(DRHO_ITER, ctx->dt);
  return 0;
}
// fused-loop variants: k_i comes from the previous drho launch's epilogue (FUSEK), kappa += k_i in the vel launch
static int divergence_iter_fused(wcsph_ctx *ctx) {
  if (ctx->d.n_owned < ctx->d.nl) return divergence_iter_async(ctx);
  ctx->lap_scope = "divergence_iter";
  PASS(k_vel_from_k<0>, (const float4 *)ctx->d.pk, ctx->dt, ctx->d.kappa_v);   // also clears ACC_ERR
  if (ctx->fuse_elem && ctx->vort_fuse && ctx->pr_valid) {
    PASS(k_drho_div_iter_vort, ctx->dt);
    ctx->vort_in_div = true;
  } else PASS((k_drho_div<DRHO_ITER, true>), ctx->dt);
  ctx->lap_scope = nullptr;
  return 0;
}
static int pressure_iter_fused(wcsph_ctx *ctx) {
  if (ctx->d.n_owned < ctx->d.nl) return pressure_iter_async(ctx);
  const bool prof = ctx->prof_on && ctx->prof_used + 2 <= ctx->prof_ev.size();
  if (prof) HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
  ctx->lap_scope = "pressure_iter";
  PASS(k_vel_from_k<0>, (const float4 *)ctx->d.pk, ctx->dt, ctx->d.kappa);     // also clears ACC_ERR
  PASS((k_drho_press<DRHO_ITER, true>), ctx->dt);
  ctx->lap_scope = nullptr;
  if (prof) { HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], ctx->stream)); ctx->prof_used += 2; }
  return 0;
}
static int fused_iter_sync(wcsph_ctx *ctx) {
  int rc = pull_scalars(ctx);
  if (rc) return rc;
  ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
  return 0;
}

int wcsph_dfsph_divergence_iter(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = divergence_iter_async(ctx);
  if (rc) return rc;
  rc = pull_scalars(ctx);
  if (rc) return rc;
  ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
  return WCSPH_OK;
}
int wcsph_dfsph_end_divergence(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_end_div, ctx->dt); return WCSPH_OK; }

// the velocity pass of warmstart_pressure as a small launch that does nothing unless the elementwise launch before it (same
// adv_rho) saw the gate adv_rho > rho_L0 open and raised flags[FLAG_WARM] to this generation number (dfsph.py:487-508, Q7)
static int warm_press_vel_flagged(wcsph_ctx *ctx, int gen) {
  if (ctx->d.nl > 0) {
    const int nb = nblk8(ctx->d.nl);
    hipLaunchKernelGGL(k_warm_press_vel_flagged, dim3(nb < 1024 ? nb : 1024), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->dt, gen);
  }
  ctx->d.rev ^= 1;   // (as PASS: the alternation of the neighbour passes stays what it was)
  LAUNCHCHK();
  return 0;
}
static int next_warm_gen(wcsph_ctx *ctx) { return ctx->warm_gen = ctx->warm_gen == 0x7fffffff ? 1 : ctx->warm_gen + 1; }
int wcsph_dfsph_warmstart_pressure(wcsph_ctx *ctx) {
  NEED_GRID();
  if (ctx->warm_scaled) {   // wcsph_dfsph_step without speculation: the scale was done by k_update_vel_warm_scale (no flag taken)
    ctx->warm_scaled = false;
    PASS(k_warm_press_vel, ctx->dt);
    return WCSPH_OK;
  }
  const int gen = next_warm_gen(ctx);
  ELEM(k_warm_press_scale, ctx->dt, gen);
  return warm_press_vel_flagged(ctx, gen);
}
int wcsph_dfsph_begin_pressure(wcsph_ctx *ctx) { NEED_GRID(); HOT_DRHO_PRESS(DRHO_BEGIN, ctx->dt); return WCSPH_OK; }
int wcsph_dfsph_pressure_iter(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = pressure_iter_async(ctx);
  if (rc) return rc;
  rc = pull_scalars(ctx);
  if (rc) return rc;
  ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
  return WCSPH_OK;
}
int wcsph_dfsph_end_pressure(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_end_press, ctx->dt); return WCSPH_OK; }
int wcsph_dfsph_clear_nonpressure(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_clear_nonpressure); return WCSPH_OK; }

int wcsph_dfsph_tension(wcsph_ctx *ctx) {
  NEED_GRID();
  // gamma = gamma_b = 0 (the reference default): both loops add exactly +-0 to d_vel; `normal` (order
  // dependent garbage in the reference, Q8) is then left untouched.
  if (ctx->d.tension == 0.0f && ctx->d.tension_b == 0.0f) return WCSPH_OK;
  if (ctx->p.tension_fixed == 0) {   // the reference loop literally, over the raw bucket rows (bucket.h)
    // its second loop gathers (normal_j, rho_j) from the records the first loop writes for OWNED rows only: with ghost rows
    // in the context (n_owned < nl) those would be stale
    if (ctx->d.n_owned < ctx->d.nl) return ctx->fail(WCSPH_EARG, "the literal compute_tension (tension_fixed = 0) needs a context that owns every row");
    int rc = bucket_build_rows(ctx);
    if (rc) return rc;
    PASS(k_tension_normal_literal, ctx->bk);
    ELEM(k_tension_accel_literal, ctx->bk);
    return WCSPH_OK;
  }
  RPASS(k_tension_normal);
  PASS(k_tension_accel);
  return WCSPH_OK;
}

int wcsph_dfsph_visc_init(wcsph_ctx *ctx) {
  NEED_GRID();
  ELEM(k_visc_guess);                   // also clears ACC_DZERO and ACC_DAD
  RPASS(k_visc_minv_residual, ctx->dt);  // dfsph.py:202-223 in one neighbour walk
  int rc = pull_scalars(ctx);
  if (rc) return rc;
  ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
  ctx->cg_delta = ctx->cg_delta_zero;
  return WCSPH_OK;
}
int wcsph_dfsph_visc_iter(wcsph_ctx *ctx) {
  NEED_GRID();
  // ACC_DAD was cleared by k_visc_guess / the previous k_visc_dir, ACC_DNEW is cleared by k_visc_Ad
  RPASS(k_visc_Ad, ctx->dt);
  ELEM(k_visc_update, ctx->cg_delta);
  ELEM(k_visc_dir, ctx->cg_delta);  // delta_old == the delta this iteration started with (dfsph.py:234)
  int rc = pull_scalars(ctx);
  if (rc) return rc;
  ctx->cg_delta_old = ctx->cg_delta;
  ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
  return WCSPH_OK;
}
int wcsph_dfsph_visc_end(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_visc_end, ctx->dt); return WCSPH_OK; }

int wcsph_dfsph_vorticity(wcsph_ctx *ctx) {
  NEED_GRID();
  RPASS(k_vorticity, ctx->dt);
  ELEM(k_vorticity_integrate, ctx->dt);
  return WCSPH_OK;
}

int wcsph_dfsph_cfl_reduce(wcsph_ctx *ctx) {
  NEED_SETUP();
  // ref: dfsph.py:108-111 `while size < particleLiquidNum`: with a single liquid particle the reference launches
  // nothing, vel_max[0] keeps its initial 0 and the time step is never adapted
  if (ctx->d.faithful && ctx->d.nl <= 1) { ctx->vel_max0 = 0.0f; return WCSPH_OK; }
  int rc;
  if (ctx->spec_cfl_seq) {   // queued behind the viscosity decision by solve_divergence, and that decision was "stop": only the mail is left
    const unsigned long long seq = ctx->spec_cfl_seq;
    ctx->spec_cfl_seq = 0; ctx->cfl_done = false;
    if ((rc = collect_mail(ctx, true, seq))) { ctx->spec_press_seq = 0; return rc; }
    ctx->spec_dt = true;
  } else {
    if (ctx->cfl_done) ctx->cfl_done = false;   // k_finish_nonpressure already took the per-particle maxima
    else ELEM(k_cfl, ctx->dt);
    hipLaunchKernelGGL(k_cfl_store, dim3(1), dim3(WAVE), 0, ctx->stream, ctx->d, DtRule{nullptr, 0.f, 0, 0.0, 0.0, 0.0, 0.0});
    LAUNCHCHK();
    if ((rc = pull_scalars(ctx))) return rc;
  }
  int bits = ctx->h_flags[FLAG_VMAX];
  memcpy(&ctx->vel_max0, &bits, 4);
  return WCSPH_OK;
}
int wcsph_dfsph_update_vel(wcsph_ctx *ctx) { NEED_SETUP(); ELEM(k_update_vel, ctx->dt); return WCSPH_OK; }
int wcsph_dfsph_update_pos(wcsph_ctx *ctx) {
  NEED_SETUP();
  ELEM(k_update_pos, ctx->dt);  // cells and lists go stale; like the reference's they stay in use until update_grid()
  return WCSPH_OK;
}

// ---- host-side control ------------------------------------------------------------------------
// wcsph_dfsph_step: update_vel + the scale of warmstart_pressure (one elementwise launch) + its velocity pass as a small launch that
// does nothing unless the first one saw the gate adv_rho > rho_L0 open (dfsph.py:487-508, Q7)
static int update_vel_warm_start(wcsph_ctx *ctx) {
  const int gen = next_warm_gen(ctx);
  ELEM(k_update_vel_warm_scale, ctx->dt, gen);
  return warm_press_vel_flagged(ctx, gen);
}
// init_viscosity_para + the first compute_viscosity_force, up to the publish of their sums (dfsph.py:90-97).  gate != nullptr:
// queued speculatively behind the decision that sets *gate (solve_divergence), every launch gated by it.
static int nonpressure_visc_open(wcsph_ctx *ctx, const int *gate, PredSpec ps) {
  const bool vort_taken = ctx->vort_in_div;   // by the last divergence iteration: only the viscosity part is left
  ctx->vort_in_div = false;
  ctx->d.pred = gate;
  if (ctx->vort_fuse && !vort_taken) RPASS(k_visc_minv_residual_vort, ctx->dt); else RPASS(k_visc_minv_residual, ctx->dt);
  RPASS(k_visc_Ad, ctx->dt);
  ELEM(k_visc_update, 0.0f, 1);
  ctx->d.pred = nullptr;
  return publish_async(ctx, ps);
}
// can the fused step queue its common-case continuation behind a solver decision (see k_publish)?
static bool spec_ok(const wcsph_ctx *ctx) {
  return ctx->fuse_elem && ctx->in_step_spec && ctx->fix_dv == 0 && ctx->fix_vs == 0 && ctx->fix_pr == 0 && ctx->pr_valid &&
         ctx->d.tension == 0.0f && ctx->d.tension_b == 0.0f && !(ctx->d.faithful && ctx->d.nl <= 1);
}

int wcsph_dfsph_compute_nonpressure_force(wcsph_ctx *ctx) {  // ref: dfsph.py:84-103
  NEED_GRID();
  int rc;
  const bool begun = ctx->nonp_begun;   // solve_divergence's last launch already cleared d_vel and advanced v_guess
  ctx->nonp_begun = false;
  const bool fuse_vort = ctx->vort_fuse;
  if (ctx->spec_visc_seq) {
    // solve_divergence queued everything up to the first mail behind its own decision, and the END of this function + the
    // CFL reduction behind the viscosity decision (the common case: one PCG iteration)
    const unsigned long long seq = ctx->spec_visc_seq;
    ctx->spec_visc_seq = 0;
    if ((rc = collect_mail(ctx, true, seq))) { ctx->spec_cfl_seq = ctx->spec_press_seq = 0; return rc; }
  } else {
    if (!begun && (rc = wcsph_dfsph_clear_nonpressure(ctx))) return rc;
    if ((rc = wcsph_dfsph_tension(ctx))) return rc;
    // init_viscosity_para + the first compute_viscosity_force with ONE host wait: the loop below always runs once, and
    // its first iteration can take delta_zero from the device (the reference waits after each, dfsph.py:86-91)
    if (!begun) ELEM(k_visc_guess);
    if ((rc = nonpressure_visc_open(ctx, nullptr, {PRED_NONE, 0.0, 1.0, nullptr, nullptr}))) return rc;
    if ((rc = collect_mail(ctx, true))) return rc;
    ctx->h_pred = 0;
  }
  const bool spec_finished = ctx->spec_cfl_seq != 0 && ctx->h_pred != 0;   // the device stopped the PCG and ran the gated end
  if (!spec_finished && ctx->spec_press_seq) ctx->prof_used = ctx->spec_prof_mark;
  if (!spec_finished) ctx->spec_cfl_seq = ctx->spec_press_seq = 0;         // (otherwise: those launches did nothing)
  ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
  ctx->cg_delta_old = ctx->cg_delta_zero;
  ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
  ctx->vs_iter = 1;
  if (spec_finished) { ctx->cfl_done = true; return WCSPH_OK; }
  // the search direction of the NEXT iteration (dfsph.py:243-246) is only formed if there is one: with the usual single
  // iteration k_visc_dir never runs (cg_dir is scratch of this loop, rebuilt by the next init_viscosity_para)
  while (ctx->vs_iter < 100) {
    if (ctx->fix_vs > 0) {  // fixed count, but never past an exactly-zero residual (beta would be 0/0)
      if (ctx->vs_iter >= ctx->fix_vs || (double)ctx->cg_delta_zero < (double)ctx->p.eps || ctx->cg_delta == 0.0f) break;
    } else if ((double)ctx->cg_delta <= (double)ctx->p.viscosity_err * (double)ctx->cg_delta_zero ||
               (double)ctx->cg_delta_zero < (double)ctx->p.eps)
      break;
    ELEM(k_visc_dir, ctx->cg_delta_old);   // beta = delta_new (still in ACC_DNEW) / delta the last iteration started with
    RPASS(k_visc_Ad, ctx->dt);
    ELEM(k_visc_update, ctx->cg_delta);
    if ((rc = pull_scalars(ctx))) return rc;
    ctx->cg_delta_old = ctx->cg_delta;
    ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
    ctx->vs_iter++;
  }
  if (fuse_vort && ctx->in_step && !(ctx->d.faithful && ctx->d.nl <= 1)) {
    // inside wcsph_dfsph_step the CFL reduction follows immediately: its per-particle part rides along
    ELEM(k_finish_nonpressure, ctx->dt);
    ctx->cfl_done = true;
    return WCSPH_OK;
  }
  if ((rc = wcsph_dfsph_visc_end(ctx))) return rc;
  if (fuse_vort) { ELEM(k_vorticity_integrate, ctx->dt); return WCSPH_OK; }   // the sums were taken in the residual walk
  return wcsph_dfsph_vorticity(ctx);
}

int wcsph_dfsph_optimize_time_step(wcsph_ctx *ctx) {  // ref: dfsph.py:107-129
  NEED_SETUP();
  int rc = wcsph_dfsph_cfl_reduce(ctx);
  if (rc) return rc;
  float vmax0 = ctx->vel_max0;
  if ((double)vmax0 > (double)ctx->p.eps) {
    double time_step = 0.5 * 0.4 * ctx->p.cfl_radius * 2.0 / sqrt((double)vmax0);
    if (time_step > ctx->p.user_max_t) time_step = ctx->p.user_max_t;
    if (time_step < ctx->p.user_min_t) time_step = ctx->p.user_min_t;
    int it = ctx->vs_iter > ctx->pr_iter ? ctx->vs_iter : ctx->pr_iter;  // max(vs, max(pr, vs)), Q6
    float dt = ctx->dt;
    if (it > 10) dt = (float)((double)dt * 0.9);
    else if (it < 5) dt = (float)((double)dt * 1.1);
    if ((double)dt > time_step) dt = (float)time_step;
    ctx->dt = dt;
  }
  if (ctx->spec_dt) {
    // The kernels queued behind the CFL reduction ran with the deltaT the DEVICE formed (k_cfl_store, DtRule: the same IEEE
    // double expressions in the same order): on this path that value is the single source of truth and the host adopts it
    //  -- the state has already advanced with it.  Development builds check that both rules agree bit for bit.
    ctx->spec_dt = false;
    float ddt; memcpy(&ddt, &ctx->h_flags[FLAG_DT], 4);
    // always on: the two rules are compared on every speculative step and a difference is COUNTED
    // (WCSPH_S_DT_RULE_MISMATCHES; the reference-text fixture tests assert 0) -- a development build refuses it
    if (memcmp(&ddt, &ctx->dt, 4) != 0) {
      ctx->n_dt_mismatch++;
#ifdef WCSPH_DEV
      ctx->spec_press_seq = 0;
      return ctx->fail(WCSPH_EHIP, "deltaT formed on the device differs from the host rule (dfsph.py:113-129)");
#endif
    }
    ctx->dt = ddt;
  }
  return WCSPH_OK;
}

int wcsph_dfsph_solve_divergence(wcsph_ctx *ctx) {  // ref: dfsph.py:131-146
  NEED_GRID();
  int rc;
  ctx->dv_iter = 0;
  if (ctx->warm_fused) {   // wcsph_dfsph_step ran k_density_coff_warm: adv_rho, kappa_v and the (pos, kappa_v) records are in place
    ctx->warm_fused = false;
    HOT_VEL(1, ctx->d.kappa_v, ctx->dt, 0);
  } else if ((rc = wcsph_dfsph_warmstart_divergence(ctx))) return rc;
  double err = -0.1;
  const bool fusek = ctx->d.n_owned == ctx->d.nl;
  if (fusek) PASS((k_drho_div<DRHO_BEGIN, true>), ctx->dt);
  else if ((rc = wcsph_dfsph_begin_divergence(ctx))) return rc;
  const float dt = ctx->dt;
  while (ctx->fix_dv > 0 ? ctx->dv_iter < ctx->fix_dv : ((double)ctx->avg_density_err > err && ctx->dv_iter < 10)) {
    if ((rc = divergence_iter_fused(ctx))) return rc;
    if (spec_ok(ctx) && ctx->dv_iter + 1 < 10) {
      // Speculation: the loop's next test is taken on the device (mail A) and the common-case continuation is queued behind
      // it in gated segments: [end of this solver + opening of compute_nonpressure_force up to its first mail B] gated by
      // A; once A has arrived with "stop": [end of compute_nonpressure_force + CFL reduction with deltaT formed on the
      // device, mail C] and [update_vel + pressure solve up to its first decision, mail D] gated by B's own decision (one
      // PCG iteration suffices), [end of the step] gated by D's.  The GPU runs from here to the end of the step without
      // waiting for the host; a decision that falls the other way leaves the gated launches as no-ops and the host
      // continues as before.
      int *pA = ctx->pred_dev + 1, *pB = ctx->pred_dev + 2;
      const double thr = 0.001 * (double)ctx->d.n_owned / (double)dt;
      if ((rc = publish_async(ctx, {PRED_DIV, thr, 1.0, pA, nullptr}))) return rc;
      const unsigned long long seqA = ctx->mail_seq;
      const bool vort = ctx->vort_in_div;
      ctx->d.pred = pA;
      ELEM(k_end_div_begin_nonpressure, ctx->dt, vort ? 1 : 0);
      ctx->d.pred = nullptr;
      if ((rc = nonpressure_visc_open(ctx, pA, {PRED_VISC, (double)ctx->p.viscosity_err, (double)ctx->p.eps, pB, pA}))) return rc;
      const unsigned long long seqB = ctx->mail_seq;
      // the rest of the step behind B's decision (see below); seqC / seqD: its two mails
      unsigned long long seqC = 0, seqD = 0;
      auto queue_rest = [&]() -> int {
        int rc = 0;
        ctx->d.pred = pB;
        ELEM(k_finish_nonpressure, ctx->dt);
        // deltaT is formed on the device too: vs_iter = 1 on this branch (gated by "one PCG iteration sufficed"), pr_iter is last step's
        const int it = ctx->pr_iter > 1 ? ctx->pr_iter : 1;
        // (the one-wave tail of the CFL reduction and the mail that carries its result: one dispatch)
        rc = cfl_store_publish_async(ctx, DtRule{ctx->dt_dev, ctx->dt, it, (double)ctx->p.eps, ctx->p.cfl_radius, ctx->p.user_max_t, ctx->p.user_min_t},
                                     {PRED_PASS, 0.0, 1.0, nullptr, pB});
        ctx->d.pred = nullptr;
        if (rc) return rc;
        seqC = ctx->mail_seq;
        // ... and the pressure solve up to its first decision (update_vel, warm start, begin, two iterations: dfsph.py:150-161
        // forces two), still gated by B, taking deltaT from the device; its end gated by its own decision (mail D)
        int *pD = ctx->pred_dev + 0;
        ctx->spec_prof_mark = ctx->prof_used;   // in-situ timing: the event pairs of iterations that turn out gated off are dropped
        ctx->d.pred = pB; ctx->d.dtp = ctx->dt_dev;
        if ((rc = update_vel_warm_start(ctx))) return rc;
        PASS((k_drho_press<DRHO_BEGIN, true>), ctx->dt);
        rc = pressure_iter_fused(ctx);
        if (!rc) rc = pressure_iter_fused(ctx);
        ctx->d.pred = nullptr; ctx->d.dtp = nullptr;
        if (rc) return rc;
        if ((rc = publish_async(ctx, {PRED_PRESS, 0.001, (double)ctx->d.n_owned, pD, pB}))) return rc;
        seqD = ctx->mail_seq;
        ctx->d.pred = pD; ctx->d.dtp = ctx->dt_dev;
        ELEM(k_end_press_update_pos, ctx->dt);
        ctx->d.pred = nullptr; ctx->d.dtp = nullptr;
        return 0;
      };
      // Small problems (config C2: a kernel takes 5-10 us) cannot wait for mail A before queueing the rest -- the GPU would
      // run dry while the host issues twenty launches; large ones do: when the divergence solve goes on -- one step in four
      // once the column collapses -- every gated launch costs 3-5 us as a no-op, and twenty of them cost more than the waits
      // they hide, while the viscosity walks keep the GPU busy for 0.2 ms after mail A.
      const bool up_front = ctx->d.nl < 400000;
      ctx->spec_prof_mark = ctx->prof_used;
      if (up_front && (rc = queue_rest())) return rc;
      if ((rc = collect_mail(ctx, true, seqA))) return rc;
      ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
      err = thr;
      ctx->dv_iter++;
      if (ctx->h_pred) {   // the device left the loop: the continuation is running
        if (!up_front && (rc = queue_rest())) return rc;
        ctx->spec_visc_seq = seqB; ctx->spec_cfl_seq = seqC; ctx->spec_press_seq = seqD;
        ctx->nonp_begun = true;
        return WCSPH_OK;
      }
      ctx->prof_used = ctx->spec_prof_mark;
      ctx->vort_in_div = vort;   // nothing of the continuation ran
      continue;
    }
    if ((rc = fused_iter_sync(ctx))) return rc;
    err = 0.001 * (double)ctx->d.n_owned / (double)dt;
    ctx->dv_iter++;
  }
  if (ctx->fuse_elem) {   // + clear_nonpressure + the v_guess update: compute_nonpressure_force follows immediately
    ELEM(k_end_div_begin_nonpressure, ctx->dt, ctx->vort_in_div ? 1 : 0);
    ctx->nonp_begun = true;
    return WCSPH_OK;
  }
  return wcsph_dfsph_end_divergence(ctx);
}

int wcsph_dfsph_solve_pressure(wcsph_ctx *ctx) {  // ref: dfsph.py:150-164
  NEED_GRID();
  int rc;
  double err = 0.0;
  const bool fusek = ctx->d.n_owned == ctx->d.nl;
  if (ctx->pr_resume) {   // wcsph_dfsph_step: the first two iterations ran speculatively and the device's decision was "go on"
    ctx->pr_resume = false;
    err = (double)ctx->avg_density_err / (double)ctx->d.n_owned;
  } else {
    if ((rc = wcsph_dfsph_warmstart_pressure(ctx))) return rc;
    ctx->pr_iter = 0;
    if (fusek) PASS((k_drho_press<DRHO_BEGIN, true>), ctx->dt);
    else if ((rc = wcsph_dfsph_begin_pressure(ctx))) return rc;
  }
  while (ctx->fix_pr > 0 ? ctx->pr_iter < ctx->fix_pr : ((err > 0.001 || ctx->pr_iter < 2) && ctx->pr_iter < 100)) {
    if ((rc = pressure_iter_fused(ctx))) return rc;
    ctx->pr_iter++;
    // the reference reads the error after every iteration, but `pr_iter < 2` forces the second one whatever the
    // first reports: the host only has to wait where the value can change the control flow
    if (ctx->fix_pr == 0 && ctx->pr_iter < 2 && fusek) { err = 1.0; continue; }
    if (ctx->fix_pr == 0 && ctx->fuse_elem && ctx->speculate && ctx->pr_iter < 100) {
      // speculation: the stopping decision is taken on the device by the kernel that publishes the error sum, and the
      // step's last kernel (end_pressure_iter + update_pos) is already queued behind it, gated by that decision -- in the
      // common case (two iterations suffice) the GPU finishes the step while the mail travels
      if ((rc = publish_async(ctx, {PRED_PRESS, 0.001, (double)ctx->d.n_owned, ctx->pred_dev + 0, nullptr}))) return rc;
      ctx->d.pred = ctx->pred_dev + 0;
      ELEM(k_end_press_update_pos, ctx->dt);
      ctx->d.pred = nullptr;
      if ((rc = collect_mail(ctx, true))) return rc;
      ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
      err = (double)ctx->avg_density_err / (double)ctx->d.n_owned;
      if (ctx->h_pred) return WCSPH_OK;      // the device stopped the solver and has run the gated kernel
      continue;                              // err > 0.001 by the device's own evaluation: the gated kernel did nothing
    }
    if ((rc = fused_iter_sync(ctx))) return rc;
    err = (double)ctx->avg_density_err / (double)ctx->d.n_owned;
  }
  if (ctx->fuse_elem) { ELEM(k_end_press_update_pos, ctx->dt); return WCSPH_OK; }   // + update_pos: the step ends here
  return wcsph_dfsph_end_pressure(ctx);
}

int wcsph_dfsph_step(wcsph_ctx *ctx) {  // ref: dfsph.py:606-617
  NEED_SETUP();
  int rc;
  // a step that failed half-way (launch error, reported device condition) must not leave its speculation state behind
  ctx->d.pred = nullptr; ctx->d.dtp = nullptr;
  ctx->spec_visc_seq = ctx->spec_cfl_seq = ctx->spec_press_seq = 0; ctx->spec_dt = false; ctx->pr_resume = false;
  PROF_LAP(ctx, "step_boundary");   // since the last launch of the previous step: the host between two steps
  if ((rc = wcsph_grid_update(ctx))) return rc;
  if (ctx->d.n_owned == ctx->d.nl) {
    PASS(k_density_coff_warm, ctx->dt);  // compute_density + compute_dfsph_coff + first warm-start pass in one walk
    ctx->warm_fused = true;
    ctx->pr_valid = true;                // ... and the (pos, rho) records of every row
  } else PASS(k_density_coff);           // compute_density + compute_dfsph_coff in one walk
  const bool fuse = ctx->d.n_owned == ctx->d.nl;
  ctx->fuse_elem = fuse;
  ctx->in_step_spec = fuse && ctx->speculate;
  rc = wcsph_dfsph_solve_divergence(ctx);
  ctx->fuse_elem = false; ctx->in_step_spec = false;
  if (rc) { ctx->nonp_begun = false; ctx->vort_in_div = false; ctx->spec_visc_seq = ctx->spec_cfl_seq = ctx->spec_press_seq = 0; return rc; }
  ctx->in_step = true;
  rc = wcsph_dfsph_compute_nonpressure_force(ctx);
  ctx->in_step = false;
  ctx->nonp_begun = false; ctx->vort_in_div = false;
  if (rc) { ctx->cfl_done = false; return rc; }
  if ((rc = wcsph_dfsph_optimize_time_step(ctx))) return rc;
  ctx->fuse_elem = fuse;
  if (ctx->spec_press_seq) {
    // update_vel and the pressure solve up to its first decision ran behind the CFL reduction (solve_divergence queued them)
    const unsigned long long seq = ctx->spec_press_seq;
    ctx->spec_press_seq = 0;
    if ((rc = collect_mail(ctx, true, seq))) { ctx->fuse_elem = false; return rc; }
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    ctx->pr_iter = 2;
    if (ctx->h_pred) { ctx->fuse_elem = false; return WCSPH_OK; }   // two iterations sufficed and the step's last kernel has run
    ctx->pr_resume = true;                                            // go on from iteration 3, one decision at a time
  } else if (fuse) { ELEM(k_update_vel_warm_scale, ctx->dt, 0); ctx->warm_scaled = true; }
  else if ((rc = wcsph_dfsph_update_vel(ctx))) return rc;
  rc = wcsph_dfsph_solve_pressure(ctx);
  ctx->fuse_elem = false; ctx->warm_scaled = false;
  if (rc) return rc;
  return fuse ? WCSPH_OK : wcsph_dfsph_update_pos(ctx);
}

int wcsph_dfsph_step_fixed(wcsph_ctx *ctx, int32_t dv, int32_t pr, int32_t vs) {
  NEED_SETUP();
  if (dv < 0 || pr < 0 || vs < 0) return WCSPH_EARG;
  ctx->fix_dv = dv; ctx->fix_pr = pr; ctx->fix_vs = vs;
  int rc = wcsph_dfsph_step(ctx);
  ctx->fix_dv = ctx->fix_pr = ctx->fix_vs = 0;
  return rc;
}


}  // extern "C"
