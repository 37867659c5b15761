// api_measure.h -- in-situ HIP-event timing of the pressure_iter launch groups and kernel-group replays (bench.py, tools/).
#pragma once
extern "C" {

// ---- measurement --------------------------------------------------------------------------------
int wcsph_profile_begin(wcsph_ctx *ctx, int32_t max_launches) {
  NEED_SETUP();
  if (max_launches < 1 || max_launches > (1 << 20)) return WCSPH_EARG;
  while (ctx->prof_ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    ctx->prof_ev.push_back(e);
  }
  ctx->prof_used = 0;
  ctx->prof_on = true;
  return WCSPH_OK;
}
int wcsph_profile_end(wcsph_ctx *ctx, float *avg_ms, int32_t *launches) {
  NEED_SETUP();
  if (!avg_ms || !launches) return WCSPH_EARG;
  ctx->prof_on = false;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (size_t k = 0; k + 1 < ctx->prof_used; k += 2) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, ctx->prof_ev[k], ctx->prof_ev[k + 1]));
    tot += t;
  }
  *launches = (int32_t)(ctx->prof_used / 2);
  *avg_ms = *launches ? (float)(tot / *launches) : 0.0f;
  return WCSPH_OK;
}

// lap profiling of the steps that follow (wcsph_ctx::lap_ev): per tag the summed time and the number of laps, as JSON text
// {"<tag>": [ms, laps, ms_of_laps_that_followed_a_host_wait], ...}
int wcsph_profile_laps_begin(wcsph_ctx *ctx, int32_t max_laps) {
  NEED_SETUP();
  if (max_laps < 2 || max_laps > (1 << 22)) return WCSPH_EARG;
  while (ctx->lap_ev.size() < (size_t)max_laps) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    ctx->lap_ev.push_back(e);
  }
  ctx->lap_tag.clear(); ctx->lap_waited.clear();
  ctx->lap_on = true; ctx->lap_wait_flag = false;
  return WCSPH_OK;
}
int wcsph_profile_laps_end(wcsph_ctx *ctx, char *json_out, size_t cap) {
  NEED_SETUP();
  if (!json_out || cap < 64) return WCSPH_EARG;
  ctx->lap_on = false;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  struct Acc { double ms = 0, ms_wait = 0; long n = 0; };
  std::vector<std::pair<std::string, Acc>> acc;
  for (size_t k = 1; k < ctx->lap_tag.size(); k++) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, ctx->lap_ev[k - 1], ctx->lap_ev[k]));
    std::string tag(ctx->lap_tag[k]);
    size_t q = 0;
    while (q < acc.size() && acc[q].first != tag) q++;
    if (q == acc.size()) acc.push_back({tag, Acc()});
    acc[q].second.ms += t; acc[q].second.n++;
    if (ctx->lap_waited[k]) acc[q].second.ms_wait += t;
  }
  std::string out = "{";
  for (size_t q = 0; q < acc.size(); q++) {
    char buf[128];
    snprintf(buf, sizeof buf, "[%.6f, %ld, %.6f]", acc[q].second.ms, acc[q].second.n, acc[q].second.ms_wait);
    std::string tag;
    for (char ch : acc[q].first) if (ch != '"' && ch != '\\') tag += ch;
    out += (q ? ", \"" : "\"") + tag + "\": " + buf;
  }
  out += "}";
  ctx->lap_tag.clear(); ctx->lap_waited.clear();
  if (out.size() + 1 > cap) return ctx->fail(WCSPH_EARG, "wcsph_profile_laps_end: buffer too small");
  memcpy(json_out, out.c_str(), out.size() + 1);
  return WCSPH_OK;
}

int wcsph_time_kernel(wcsph_ctx *ctx, const char *name, int reps, float *ms) {
  NEED_GRID();
  if (!name || reps < 1 || !ms) return WCSPH_EARG;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  std::string nm(name);
  int rc = 0;
  auto one = [&]() -> int {
    if (nm == "pressure_iter") return pressure_iter_fused(ctx);      // what wcsph_dfsph_solve_pressure launches per iteration
    if (nm == "divergence_iter") return divergence_iter_fused(ctx);
    if (nm == "divergence_iter_vort") { PASS(k_vel_from_k<0>, (const float4 *)ctx->d.pk, ctx->dt, ctx->d.kappa_v); RPASS(k_drho_div_iter_vort, ctx->dt); return 0; }
    if (nm == "pressure_iter_unfused") return pressure_iter_async(ctx);  // k_make_k + vel + drho (per-kernel API)
    if (nm == "pressure_vel_loop") { HOT_VEL(0, ctx->d.kbuf, ctx->dt, 0); return 0; }
    if (nm == "pressure_drho_loop") { HOT_DRHO_PRESS(DRHO_ITER, ctx->dt); return 0; }
    if (nm == "density") { PASS(k_density); return 0; }
    if (nm == "density_coff") { PASS(k_density_coff); return 0; }
    if (nm == "density_coff_warm") { PASS(k_density_coff_warm, ctx->dt); return 0; }
    if (nm == "grid_update") return wcsph_grid_update(ctx);
    if (nm == "visc_minv_residual_vort") { RPASS(k_visc_minv_residual_vort, ctx->dt); return 0; }
    if (nm == "vorticity") { RPASS(k_vorticity, ctx->dt); return 0; }
    if (nm == "visc_minv_residual") { int r = zero_acc(ctx, ACC_DZERO); if (r) return r; RPASS(k_visc_minv_residual, ctx->dt); return 0; }
    if (nm == "build_lists") { PASS(k_build_lists, 1); return 0; }
    if (nm == "visc_iter") { int r = zero_acc(ctx, ACC_DAD); if (r) return r; RPASS(k_visc_Ad, ctx->dt); return 0; }
    if (nm == "drho_press_rec") { PASS((k_drho_press<DRHO_ITER, true>), ctx->dt); return 0; }   // the density-change pass as the fused loop launches it
    return ctx->fail(WCSPH_EARG, "wcsph_time_kernel: unknown kernel group");
  };
  if (nm == "pressure_iter" && (rc = pressure_iter_async(ctx))) return rc;      // primes the (pos, k) records
  if (nm == "divergence_iter" && (rc = divergence_iter_async(ctx))) return rc;
  if ((rc = one())) return rc;  // warm-up
  HIPCHK(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < reps && !rc; r++) rc = one();
  HIPCHK(hipEventRecord(e1, ctx->stream));
  HIPCHK(hipEventSynchronize(e1));
  float t = 0.f;
  HIPCHK(hipEventElapsedTime(&t, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc) return rc;
  *ms = t / (float)reps;
  return WCSPH_OK;
}


}  // extern "C"
