// api_tile.h -- host side of the LDS tile walk (tile.h): tables, per-update build, the two pressure passes.
#pragma once

static bool tile_usable(const wcsph_ctx *ctx) {
  return ctx->tile_on && ctx->d.tl_n && ctx->d.n_owned == ctx->d.nl && ctx->d.part_mode == 0 && ctx->win_x1 < 0 && ctx->d.nl > 0;
}
static int tile_setup(wcsph_ctx *ctx) {   // end of wcsph_setup
  Dev &d = ctx->d;
  const char *env = getenv("WCSPH_TILE");
  ctx->tile_on = env && atoi(env) != 0;
  if (!ctx->tile_on || ctx->win_x1 >= 0 || d.nl == 0) { ctx->tile_on = false; return 0; }
  d.tg0n = (d.nt0 + (1 << GT0B) - 1) >> GT0B;
  d.tg1n = (d.nt1 + (1 << GT1B) - 1) >> GT1B;
  d.tl_ngroups = d.tg0n * d.tg1n * d.nt2;
  const size_t max_active = (size_t)(d.tl_ngroups < d.nl ? d.tl_ngroups : d.nl);
  DALLOC(d.tl_n, 8); DALLOC(d.tl_own, (size_t)d.tl_ngroups); DALLOC(d.tl_act, max_active + 1);
  DALLOC(d.tl_grp, max_active * TL_DESC);
  DALLOC(d.tl_tbl, 8 * (size_t)d.nl + 16 * max_active + 64);   // a cell lies in the boxes of at most 2 x 2 x 2 groups
  DALLOC(d.nbt, (size_t)d.nbr_stride * (d.nbr_cap / 8)); DALLOC(d.ntl, (size_t)d.nl);
  int dev = 0, ncu = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  ctx->tile_cus = ncu > 0 ? (ncu + 7) / 8 * 8 : 256;
  ctx->tile_lds_build = TL_CAP * 16 + (TL_NBOX + 1) * 2 + 14;
  HIPCHK(hipFuncSetAttribute((const void *)k_tile_build, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->tile_lds_build));
  HIPCHK(hipFuncSetAttribute((const void *)k_tile_vel<TILE_NT_V>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_CAP * 16));
  HIPCHK(hipFuncSetAttribute((const void *)k_tile_drho<TILE_NT_D>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_CAP * 32));
  return 0;
}
// after the counting sort (grid_lists): occupied groups, their boxes and slot tables, every owner's slot list
static int tile_build_launches(wcsph_ctx *ctx) {
  if (!ctx->d.tl_n || ctx->d.n_owned != ctx->d.nl || ctx->d.nl == 0) return 0;   // (built whenever the tables exist: wcsph_tile_enable may switch the walk on later)
  Dev &d = ctx->d;
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_tile_mark, dim3(nblk(d.tl_ngroups)), dim3(BLOCK), 0, s, d);
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, d);
  hipLaunchKernelGGL(k_tile_build, dim3(ctx->tile_cus * 3), dim3(BLOCK), ctx->tile_lds_build, s, d);
  LAUNCHCHK();
  return 0;
}
static int tile_vel_launch(wcsph_ctx *ctx, float *kap) {
  hipLaunchKernelGGL(k_tile_vel<TILE_NT_V>, dim3(ctx->tile_cus * TILE_WG_V), dim3(TILE_NT_V), TL_CAP * 16, ctx->stream, ctx->d, (const float4 *)ctx->d.pk, ctx->dt, kap);
  ctx->d.rev ^= 1;   // (as PASS: the alternation of the gather passes stays what it was)
  LAUNCHCHK();
  return 0;
}
static int tile_drho_launch(wcsph_ctx *ctx) {
  hipLaunchKernelGGL(k_tile_drho<TILE_NT_D>, dim3(ctx->tile_cus * TILE_WG_D), dim3(TILE_NT_D), TL_CAP * 32, ctx->stream, ctx->d, ctx->dt);
  ctx->d.rev ^= 1;
  LAUNCHCHK();
  return 0;
}
