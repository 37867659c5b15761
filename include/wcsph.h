/*
 * wcsph.h -- C ABI of libwcsph_hip.so: the MI355X (gfx950) SPH hot path behind the
 * lyd405121/wcsph Python surface (ParticleData / HashGrid / dfsph.py / sesph.py kernels).
 *
 * The reference has no FFI; its "operator interface" for this path is the set of
 * zero-argument @ti.kernel functions the driver loops call.  Each entry
 * point below replaces exactly one of them and cites it ("ref: file:line" into the
 * reference tree).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns int: 0 = OK, <0 = error
 *       WCSPH_EARG (-1) bad argument        WCSPH_EHIP  (-2) HIP runtime error
 *       WCSPH_EBUCKET (-3) > maxInGrid particles in one hash bucket  (ref: HashGrid.py:72-74)
 *       WCSPH_ENEIGH  (-4) > maxNeighbour list entries for a particle (ref: HashGrid.py:100-103)
 *       WCSPH_ENONFINITE (-5) non-finite reduction result
 *   - wcsph_last_error(ctx) returns a human readable message for the last failure
 *   - a context owns all device memory; host arrays are caller-owned and copied synchronously
 *   - one host thread drives one context; all work is queued on one HIP stream
 *     (wcsph_set_stream); kernel entry points are asynchronous, *_to_host / get_* synchronise
 *   - host-visible particle order is ALWAYS the caller's original order (liquid first, then
 *     solid, ref: ParticleData.py:100-126); the cell-sorted device order is internal
 */
#ifndef WCSPH_H
#define WCSPH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WCSPH_OK 0
#define WCSPH_EARG (-1)
#define WCSPH_EHIP (-2)
#define WCSPH_EBUCKET (-3)
#define WCSPH_ENEIGH (-4)
#define WCSPH_ENONFINITE (-5)

/* neighbour semantics (DESIGN.md "Neighbour modes") */
#define WCSPH_MODE_FAITHFUL 0 /* reference hash table with `count` buckets: bucket multiset, HashGrid.py:56-114 */
#define WCSPH_MODE_CLEAN 1    /* collision-free hash: every particle of the 5x5x5 in-box cells once */

typedef struct wcsph_ctx wcsph_ctx;

/* Python-scope constants of ParticleData / HashGrid / the kernels (ref: ParticleData.py:12-31,61-87;
 * HashGrid.py:10-19; kernels/CubicKernel.py:12-17; dfsph.py:19-41).  Doubles where the reference
 * keeps Python floats that take part in host-side (double) arithmetic. */
typedef struct wcsph_params {
  int32_t n_liquid, n_solid;
  int32_t neighbor_mode;     /* WCSPH_MODE_* */
  int32_t max_in_grid;       /* 64   (ParticleData.py:27) */
  int32_t max_neighbour;     /* 2048 (ParticleData.py:27) */
  int32_t hash_table_size;   /* 0 => n_liquid + n_solid (HashGrid.py:114); slabs pass the global count */
  int32_t tension_fixed;     /* 1: n = h * sum, curvature over the support (documented repair of Q8); 0: the loop of
                              * dfsph.py:264-305 literally over the raw bucket rows (faithful mode, one context = whole domain) */
  int32_t list_cap;          /* entries per particle of the compact lists; 0 => 256 slots.  pcisph needs the reference's 2048 */
  double particle_radius;    /* what the script passes to ParticleData(): radius (dfsph) or diameter (sesph, Q11) */
  double cfl_radius;         /* dfsph.py:27 particleRadius used by optimize_time_step */
  double user_max_t, user_min_t; /* dfsph.py:39-40 */
  float rho_L0, rho_S0, VL0, VS0, liqiudMass;
  float gravity[3];
  float dim_coff, viscosity, viscosity_b, viscosity_err;
  float tension_coff, tension_coff_b;
  float viscosity_omega, vorticity_coff, vorticity_init;
  float eps;
  /* sesph.py private constants (sesph.py:23-62) */
  float ss_searchR, ss_mk, ss_ml, ss_VL0, ss_VS0, ss_mass, ss_rho0, ss_stiff, ss_dim, ss_visc, ss_visc_b;
  float list_radius;         /* radius of the compact per-step neighbour lists; 0 => HashGrid.searchR; < 0 => unbounded:
                              * every particle of the walked cells, i.e. the reference's raw list (HashGrid.py:88-106) */
  /* pcisph.py:68-70,299 and iisph.py:80-83 private constants */
  float pci_coff, pc_visc, pc_visc_b, ii_visc, ii_visc_b, ii_omega;
} wcsph_params;

/* field ids for wcsph_field_to_host / wcsph_field_from_host (ref: ParticleData.py:33-76, dfsph.py:46-48) */
enum wcsph_field {
  WCSPH_F_POS = 0,     /* [count][3] f32 (liquid rows are dynamic, solid rows static) */
  WCSPH_F_VEL,         /* [n_liquid][3] */
  WCSPH_F_VEL_GUESS, WCSPH_F_D_VEL, WCSPH_F_OMEGA, WCSPH_F_D_OMEGA, WCSPH_F_NORMAL,
  WCSPH_F_CG_R, WCSPH_F_CG_DIR, WCSPH_F_CG_AD, WCSPH_F_CG_S,
  WCSPH_F_RHO,         /* [n_liquid] f32 */
  WCSPH_F_ADV_RHO, WCSPH_F_PRESSURE, WCSPH_F_VEL_MAX, WCSPH_F_ALPHA, WCSPH_F_KAPPA, WCSPH_F_KAPPA_V,
  WCSPH_F_CG_MINV,     /* [n_liquid][3][3] f32 */
  WCSPH_F_NEIGHBOR_COUNT, /* [n_liquid] i32  (HashGrid.neighborCount) */
  WCSPH_F_KBUF,        /* [n_liquid] f32 scratch: k_i = b_i * alpha_i of the running iteration (dfsph.py:456,525) */
  WCSPH_F_D_VEL_PRE, WCSPH_F_POS_STAR, WCSPH_F_VEL_STAR,   /* pcisph.py:52-54  [n_liquid][3] */
  WCSPH_F_D_II, WCSPH_F_DIJ_PJ,                            /* iisph.py:58-59   [n_liquid][3] */
  WCSPH_F_A_II, WCSPH_F_PRESSURE_PRE,                      /* iisph.py:57,60   [n_liquid] */
  WCSPH_F_COLOR,       /* ParticleData.py:37   [n_liquid]        (after wcsph_surface_color_map) */
  WCSPH_F_COLOR_GRAD, WCSPH_F_POS_AVR,                     /* ParticleData.py:38,40 [n_liquid][3] */
  WCSPH_F_G,           /* ParticleData.py:41   [n_liquid][3][3]  (after wcsph_surface_anisotropic_kernel) */
  WCSPH_F__COUNT
};

/* scalar ids for wcsph_get_scalar / wcsph_set_scalar (shape-(1) fields, ref: ParticleData.py:66-69, dfsph.py:41) */
enum wcsph_scalar {
  WCSPH_S_DELTA_T = 0, WCSPH_S_AVG_DENSITY_ERR, WCSPH_S_CG_DELTA, WCSPH_S_CG_DELTA_OLD, WCSPH_S_CG_DELTA_ZERO,
  WCSPH_S_VISCOSITY, WCSPH_S_VISCOSITY_B, WCSPH_S_TENSION, WCSPH_S_TENSION_B, WCSPH_S_VEL_MAX0,
  WCSPH_S_RHO_ERR,     /* pcisph.py:58 rho_err */
  WCSPH_S_TABLE_UPDATES_IN_PLACE,   /* read only, diagnostics: grid updates of this context that updated the reference bucket table in place instead of rebuilding it (api_grid.h grid_assign) */
  WCSPH_S_DT_RULE_MISMATCHES,       /* read only, diagnostics: speculative steps whose device-formed deltaT (k_cfl_store, DtRule) differed from the host rule of dfsph.py:113-129 evaluated on the same inputs; stays 0 (the device value is the one the step ran with, and is adopted) */
  WCSPH_S__COUNT
};

/* ---- lifetime ------------------------------------------------------------------------ */
int wcsph_create(const wcsph_params *p, wcsph_ctx **out);
int wcsph_destroy(wcsph_ctx *ctx);
const char *wcsph_last_error(wcsph_ctx *ctx);
int wcsph_set_stream(wcsph_ctx *ctx, void *hip_stream); /* NULL => the context's own stream */
int wcsph_sync(wcsph_ctx *ctx);                          /* drain the stream, surface -3/-4/-5 */
int wcsph_device_count(void);

/* ref: ParticleData.setup_data_cpu (ParticleData.py:180-185) + HashGrid.setup_grid_cpu (HashGrid.py:44-54):
 * uploads [count][3] positions (liquid first), fixes the grid box from maxb/minb, sorts the static solids. */
int wcsph_setup(wcsph_ctx *ctx, const float *pos_xyz, const float *maxb, const float *minb);
int wcsph_get_block_size(wcsph_ctx *ctx, int32_t bs[3]);

/* Field transfer in the caller's ORIGINAL particle order (ref: field.to_numpy() / from_numpy(), dfsph.py:113,129,141,162).
 * A transfer to the host is a SYNCHRONISATION POINT: error conditions raised on the device since the last one (bucket / list
 * overflow: -3 / -4, non-finite position: -5; sticky on the device until handed over once) are reported by ITS return code -- the
 * data is delivered either way, the code says whether such a condition arose while it was produced.  A caller that keeps
 * stepping after a non-zero return does what the reference does after its in-kernel `print` (HashGrid.py:73,103). */
int wcsph_field_to_host(wcsph_ctx *ctx, int field, void *dst, size_t bytes);
int wcsph_field_from_host(wcsph_ctx *ctx, int field, const void *src, size_t bytes);
/* rows [i0, i0+n) of a liquid field, original order: Python-scope element access `f[i]` / `f[i] = v`
 * (ref: dfsph.py:98 `particle_data.cg_delta[0]`, :645 `pos.to_numpy()[test_id]`) without moving the whole field */
int wcsph_field_rows_to_host(wcsph_ctx *ctx, int field, int32_t i0, int32_t n, void *dst);
int wcsph_field_rows_from_host(wcsph_ctx *ctx, int field, int32_t i0, int32_t n, const void *src);
int wcsph_get_scalar(wcsph_ctx *ctx, int id, float *v);
int wcsph_set_scalar(wcsph_ctx *ctx, int id, float v);
int wcsph_get_iters(wcsph_ctx *ctx, int32_t *vs_iter, int32_t *dv_iter, int32_t *pr_iter);
int wcsph_set_iters(wcsph_ctx *ctx, int32_t vs_iter, int32_t dv_iter, int32_t pr_iter);

/* ---- HashGrid ------------------------------------------------------------------------ */
/* ref: HashGrid.update_grid (HashGrid.py:56-85): rebuild cells + per-particle neighbour multiset.
 * Everything the call queues is on the context's stream.  In faithful mode one context keeps the reference's bucket table up to date
 * between two of its own updates (a particle's previous cell travels in pos.w) instead of rebuilding it. */
int wcsph_grid_update(wcsph_ctx *ctx);
/* ref: HashGrid.neighbor / neighborCount (HashGrid.py:24-25,88-106): materialise the explicit list the
 * reference stores, original indices, for particles [i0, i0+n): counts[n], lists[n][cap] (-1 padded) */
int wcsph_neighbors_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists);
/* ref: HashGrid.gridCount (HashGrid.py:20,71): occupancy of every bucket of the reference's table, n = table size
 * (faithful mode).  ref: HashGrid.grid (HashGrid.py:21,76): rows [h0, h0+n) of the table, global particle ids
 * (liquid: original index; solid: n_liquid + index), ascending inside a bucket, -1 padded to cap.  The table is
 * materialised on demand (csrc/bucket.h); wcsph_neighbors_to_host() reads its rows from it, so in faithful mode they are
 * the reference's rows entry for entry, far aliased particles included. */
int wcsph_hashgrid_count_to_host(wcsph_ctx *ctx, int32_t *dst, int64_t n);
int wcsph_hashgrid_rows_to_host(wcsph_ctx *ctx, int32_t h0, int32_t n, int32_t cap, int32_t *rows);
/* diagnostics: the compact per-step lists the neighbour passes actually iterate -- the in-support subset of the row
 * above (`while k < cur_neighbor` loops, e.g. dfsph.py:254-256, weigh everything else by exactly zero), decoded from
 * the 16-bit delta streams; original indices (solids offset by n_liquid), -1 padded.
 * counts[n][3] = {liquid entries, solid entries, hop slots}; lists[n][cap] = liquid entries then solid entries */
int wcsph_compact_lists_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists);
/* diagnostics: how evenly the compact lists load the 64-lane waves that walk them (a wave runs as many trips as its
 * longest list).  Histograms over 8-slot stream blocks, 0..63 (64 bins each):
 * hist[0] particles by liquid blocks, hist[1] by solid blocks, hist[2] / hist[3] waves (64 consecutive cell-sorted
 * particles) by their LONGEST liquid / solid list, hist[4] waves by their largest liquid + solid block count (the trips a
 * neighbour pass makes for the wave: one trip sequence serves both streams) */
int wcsph_list_stats(wcsph_ctx *ctx, int64_t hist[5][64]);

/* ---- dfsph.py kernels (one entry per @ti.kernel) ------------------------------------- */
int wcsph_dfsph_reset_param(wcsph_ctx *ctx);            /* ref: dfsph.py:167-178 */
int wcsph_dfsph_compute_density(wcsph_ctx *ctx);        /* ref: dfsph.py:248-262 */
int wcsph_dfsph_compute_coff(wcsph_ctx *ctx);           /* ref: dfsph.py:345-372 */
int wcsph_dfsph_warmstart_divergence(wcsph_ctx *ctx);   /* ref: dfsph.py:415-439 */
int wcsph_dfsph_begin_divergence(wcsph_ctx *ctx);       /* ref: dfsph.py:441-446 */
int wcsph_dfsph_divergence_iter(wcsph_ctx *ctx);        /* ref: dfsph.py:449-477 */
int wcsph_dfsph_end_divergence(wcsph_ctx *ctx);         /* ref: dfsph.py:480-484 */
int wcsph_dfsph_warmstart_pressure(wcsph_ctx *ctx);     /* ref: dfsph.py:487-508 */
int wcsph_dfsph_begin_pressure(wcsph_ctx *ctx);         /* ref: dfsph.py:511-516 */
int wcsph_dfsph_pressure_iter(wcsph_ctx *ctx);          /* ref: dfsph.py:518-547 */
int wcsph_dfsph_end_pressure(wcsph_ctx *ctx);           /* ref: dfsph.py:549-552 */
int wcsph_dfsph_clear_nonpressure(wcsph_ctx *ctx);      /* ref: dfsph.py:333-336 */
int wcsph_dfsph_tension(wcsph_ctx *ctx);                /* ref: dfsph.py:264-305 */
int wcsph_dfsph_visc_init(wcsph_ctx *ctx);              /* ref: dfsph.py:197-223 */
int wcsph_dfsph_visc_iter(wcsph_ctx *ctx);              /* ref: dfsph.py:225-246 */
int wcsph_dfsph_visc_end(wcsph_ctx *ctx);               /* ref: dfsph.py:339-343 */
int wcsph_dfsph_vorticity(wcsph_ctx *ctx);              /* ref: dfsph.py:307-330 */
int wcsph_dfsph_cfl_reduce(wcsph_ctx *ctx);             /* ref: dfsph.py:107-111,555-568 (all launches of the tree) */
int wcsph_dfsph_update_vel(wcsph_ctx *ctx);             /* ref: dfsph.py:572-575 */
int wcsph_dfsph_update_pos(wcsph_ctx *ctx);             /* ref: dfsph.py:577-580 */

/* ---- dfsph.py host-side control (same stopping rules, loop kept next to the device) --- */
int wcsph_dfsph_compute_nonpressure_force(wcsph_ctx *ctx); /* ref: dfsph.py:84-103  */
int wcsph_dfsph_optimize_time_step(wcsph_ctx *ctx);        /* ref: dfsph.py:107-129 */
int wcsph_dfsph_solve_divergence(wcsph_ctx *ctx);          /* ref: dfsph.py:131-146 */
int wcsph_dfsph_solve_pressure(wcsph_ctx *ctx);            /* ref: dfsph.py:150-164 */
int wcsph_dfsph_step(wcsph_ctx *ctx);                      /* ref: dfsph.py:606-617 */
/* the same step with the three solver loops run for a FIXED number of iterations (equal-work micro-benchmark,
 * SURVEY 8(d)); 0 for a count keeps that loop's reference stopping rule */
int wcsph_dfsph_step_fixed(wcsph_ctx *ctx, int32_t dv_iters, int32_t pr_iters, int32_t vs_iters);

/* ---- sesph.py kernels ----------------------------------------------------------------- */
int wcsph_sesph_reset_param(wcsph_ctx *ctx);   /* ref: sesph.py:130-135 */
int wcsph_sesph_density(wcsph_ctx *ctx);       /* ref: sesph.py:138-154 */
int wcsph_sesph_pressure(wcsph_ctx *ctx);      /* ref: sesph.py:158-166 */
int wcsph_sesph_force(wcsph_ctx *ctx);         /* ref: sesph.py:168-188 */
int wcsph_sesph_integrate(wcsph_ctx *ctx);     /* ref: sesph.py:191-196 */
int wcsph_sesph_step(wcsph_ctx *ctx);          /* ref: sesph.py:217-225 */

/* ---- pcisph.py kernels ---------------------------------------------------- */
int wcsph_pcisph_reset_param(wcsph_ctx *ctx);               /* ref: pcisph.py:193-197 */
int wcsph_pcisph_compute_nonpressure_force(wcsph_ctx *ctx); /* ref: pcisph.py:199-218 */
int wcsph_pcisph_init_iter_info(wcsph_ctx *ctx);            /* ref: pcisph.py:220-226 */
int wcsph_pcisph_update_iter_info(wcsph_ctx *ctx);          /* ref: pcisph.py:228-235 */
int wcsph_pcisph_predict_density(wcsph_ctx *ctx);           /* ref: pcisph.py:237-278 */
int wcsph_pcisph_update_pos(wcsph_ctx *ctx);                /* ref: pcisph.py:281-285 */
int wcsph_pcisph_solve_pressure(wcsph_ctx *ctx);            /* ref: pcisph.py:147-157 (sovel_pressure) */
int wcsph_pcisph_step(wcsph_ctx *ctx);                      /* ref: pcisph.py:303-312 */

/* ---- iisph.py kernels ------------------------------------------------------ */
int wcsph_iisph_reset_param(wcsph_ctx *ctx);                /* ref: iisph.py:178-182 */
int wcsph_iisph_compute_density(wcsph_ctx *ctx);            /* ref: iisph.py:254-268 */
int wcsph_iisph_visc_init(wcsph_ctx *ctx);                  /* ref: iisph.py:200-229 */
int wcsph_iisph_visc_iter(wcsph_ctx *ctx);                  /* ref: iisph.py:231-252 */
int wcsph_iisph_combine_nonpressure(wcsph_ctx *ctx);        /* ref: iisph.py:270-274 */
int wcsph_iisph_compute_nonpressure_force(wcsph_ctx *ctx);  /* ref: iisph.py:114-127 */
int wcsph_iisph_compute_advection(wcsph_ctx *ctx);          /* ref: iisph.py:276-316 */
int wcsph_iisph_update_iter_info(wcsph_ctx *ctx);           /* ref: iisph.py:318-334 */
int wcsph_iisph_update_pressure_force(wcsph_ctx *ctx);      /* ref: iisph.py:336-370 */
int wcsph_iisph_solve_pressure(wcsph_ctx *ctx);             /* ref: iisph.py:130-139 */
int wcsph_iisph_update_pos(wcsph_ctx *ctx);                 /* ref: iisph.py:372-396 */
int wcsph_iisph_step(wcsph_ctx *ctx);                       /* ref: iisph.py:419-428 */

/* ---- ParticleData surface-reconstruction kernels ------------------------- */
int wcsph_surface_color_map(wcsph_ctx *ctx);                          /* ref: ParticleData.py:187-215 compute_color_map */
int wcsph_surface_anisotropic_kernel(wcsph_ctx *ctx, float mc_searchR); /* ref: ParticleData.py:220-289 cal_anistropic_kernel */

/* ---- Canvas.py, headless -------------------------------------------------------------------------- */
/* ref: dfsph.py:585-593 draw_particle over Canvas.py:138-211 -- clears the canvas, draws every liquid particle as a
 * circle outline (r = 3 px, color_sphere) and every particle as a point (color_point) through proj_view (row-major
 * 4x4 = Canvas.proj @ Canvas.view), nearest fragment wins; img[size_x][size_y][3] float RGB like Canvas.img */
int wcsph_canvas_draw(wcsph_ctx *ctx, const float *proj_view, int32_t size_x, int32_t size_y, const float *color_sphere,
                      const float *color_point, float *img);

/* ---- MarchingCubeGrid.py (grid half) --------------------------------------------------------- */
/* the generated 256-case tables the device uses: edge_table[256], tri_table[256][16] (-1 terminated rows); replaces
 * the data file MCData.txt read at MarchingCubeGrid.py:82-94 */
int wcsph_mc_tables(int32_t *edge_table, int32_t *tri_table);
/* ref: MCGrid.setup_grid_gpu / setup_grid_cpu (MarchingCubeGrid.py:60-103): node pitch grid_r, kernel support search_r,
 * box [minb, maxb]; dims[a] = int((maxb[a] - minb[a]) / grid_r + 1) */
int wcsph_mc_setup(wcsph_ctx *ctx, double grid_r, double search_r, const float *maxb, const float *minb, int32_t dims[3]);
/* ref: MCGrid.cal_surface_point (MarchingCubeGrid.py:159-184); needs the grid of this step and a current rho */
int wcsph_mc_surface_value(wcsph_ctx *ctx);
int wcsph_mc_value_to_host(wcsph_ctx *ctx, float *out, int64_t n_nodes);
/* ref: MCGrid.marching_cube (MarchingCubeGrid.py:232-328): triangle soup, three consecutive vertices per triangle, at
 * most cap_vertices stored; *n_vertices = vertices generated (> cap_vertices: truncated, the reference's MAX_VERTEX) */
int wcsph_mc_marching_cube(wcsph_ctx *ctx, float isolevel, float *tri_xyz, int32_t cap_vertices, int32_t *n_vertices);

/* ---- boundary sampler ------------------------------------------------------- */
/* ref: boundry.py:100-460 -- Poisson-disk sampling of a triangle mesh (tri_xyz: [n_tri][3 vertices][3]) into
 * boundary particles no closer than particle_radius (normal-corrected distance, boundry.py:340-370).
 * Self-contained (no context), synchronous; out_xyz receives at most `cap` points, *n_out the number produced
 * (if *n_out > cap the output was truncated). */
int wcsph_sample_surface(const float *tri_xyz, int32_t n_tri, double particle_radius, uint32_t seed, int32_t trials,
                         float *out_xyz, int32_t cap, int32_t *n_out);

/* ---- measurement ---------------------------------------------------------------------- */
/* In-situ timing of the dominant kernel pair: between begin and end every fused `pressure_iter` launch group
 * (k_vel_from_k + k_drho_press<ITER>, dfsph.py:518-547) inside wcsph_dfsph_step / _solve_pressure is bracketed by HIP
 * events on the context's stream (at most max_launches of them); end returns their average duration. */
int wcsph_profile_begin(wcsph_ctx *ctx, int32_t max_launches);
int wcsph_profile_end(wcsph_ctx *ctx, float *avg_ms, int32_t *launches);
/* lap profiling (bench.py's per-group breakdown of a step): between _begin and _end every tagged launch of the fused step is
 * followed by a HIP event; the time between two consecutive events is booked to the tag of the later one (the launch itself plus
 * whatever idle time preceded it).  _end waits for the stream and writes JSON text {"<tag>": [ms, laps, ms of the laps that
 * began behind a host wait], ...} (tags: kernel names, "grid:cells", "grid:scan+sort", "grid:neighborCount", "step_boundary").
 * The events perturb the step by a few per cent: time a window WITHOUT them for the rate. */
int wcsph_profile_laps_begin(wcsph_ctx *ctx, int32_t max_laps);
int wcsph_profile_laps_end(wcsph_ctx *ctx, char *json_out, size_t cap);
/* Times `reps` back-to-back launches of one named kernel group with hipEvents on the context's
 * stream; *ms_per_launch is the average.  Names: "pressure_iter", "divergence_iter", "density",
 * "grid_update", "visc_iter".  State is not preserved across the call. */
int wcsph_time_kernel(wcsph_ctx *ctx, const char *name, int reps, float *ms_per_launch);

/* ---- slab decomposition: one context per GPU, host code exchanges halos between phases -- */
/* Restrict the dense cell table to global cell columns [x0, x1) along x; call BEFORE wcsph_setup().  Cell
 * coordinates (and the reference hash) stay global, so every rank sees the same cells (HashGrid.py:68). */
int wcsph_set_cell_window(wcsph_ctx *ctx, int32_t x0, int32_t x1);
/* Communication boundary of the decomposition (SURVEY 8(b) lists wcsph_comm_init / wcsph_halo_exchange): the library
 * owns everything device-side of an exchange -- which rows travel, packing, unpacking, the bucket histogram -- and the
 * driver owns the communicator (torch.distributed process group == RCCL, created by whoever launched the ranks, as
 * bench.py's contract prescribes): between wcsph_halo_pack() and wcsph_halo_unpack() it moves ONE contiguous device
 * buffer per neighbour with ncclSend/ncclRecv (INTEGRATION.md "Multi-GPU").
 * Faithful neighbour mode on slabs: the reference hash table is global (HashGrid.py:108-114, `count` buckets), so
 *  - params.hash_table_size = global particle count, wcsph_set_global_liquid() = global liquid count (CFL prefix, Q5),
 *  - wcsph_set_owned_columns(): the global cell columns whose boundary particles THIS rank counts (each exactly once),
 *  - wcsph_bucket_hist(which = 1): static boundary histogram, summed over ranks ONCE after wcsph_setup();
 *    (which = 0): liquid histogram of the step, int32[n] on the device, summed over ranks between
 *    wcsph_grid_update_begin() [cells, this rank's histogram], wcsph_grid_update_mid() [scan, sort, list build: reads nothing of
 *    the histogram, so the all-reduce can be in flight behind it; optional -- _end() runs whatever has not run] and
 *    wcsph_grid_update_end() [neighborCount] (wcsph_grid_update() = all of it, for a single context). */
int wcsph_set_owned_columns(wcsph_ctx *ctx, int32_t lo, int32_t hi);
int wcsph_set_global_liquid(wcsph_ctx *ctx, int64_t n_liquid_global);
int wcsph_grid_update_begin(wcsph_ctx *ctx);
int wcsph_grid_update_mid(wcsph_ctx *ctx);
int wcsph_grid_update_end(wcsph_ctx *ctx);
int wcsph_bucket_hist(wcsph_ctx *ctx, int which, void **d_ptr, int64_t *n);
/* the per-step liquid histogram packed for the wire: two 16-bit counts per int32 word (an int32 SUM all-reduce adds the
 * halves independently; half the bytes of the plain histogram); _unpack() after the all-reduce, before _grid_update_end() */
int wcsph_bucket_hist_pack(wcsph_ctx *ctx, void **d_ptr, int64_t *n_words);
int wcsph_bucket_hist_unpack(wcsph_ctx *ctx);
/* Once per step, before the rows change owner (no reference counterpart: the reference is single-device): classify the n
 * state rows at d_rows (wcsph_load_state's row format, device memory) by their GLOBAL cell column c (HashGrid.py:68) against
 * this slab's cuts [lo, hi): list 0 / 1 = lo <= c < lo + ghost / hi - ghost <= c < hi (ghosts of the left / right neighbour),
 * list 2 / 3 = c < lo / c >= hi (change owner), count 4 = further than `ghost` columns beyond a cut (the caller's error).
 * has_left / has_right: whether there is a neighbour on that side.  d_idx: device int64 [4][cap] row indices, d_counts:
 * device int32 [5] (zeroed here); a count above cap means that list is truncated.  One launch on the context's stream. */
int wcsph_slab_classify(wcsph_ctx *ctx, const void *d_rows, int64_t n, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right,
                        int64_t *d_idx, int32_t cap, int32_t *d_counts);
/* rows d_idx[0..n) (original indices, device array) of n_fields fields (enum wcsph_field, vec3 / scalar) side by side:
 * d_out / d_in is a device buffer [n][sum of widths] f32; one launch each */
int wcsph_halo_pack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, void *d_out);
int wcsph_halo_unpack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, const void *d_in);
/* Replace the liquid set: d_rows is a DEVICE array [n_total][15] f32 = pos3 vel3 omega3 vel_guess3 kappa kappa_v gid
 * (the fields the reference carries across steps + the particle's global id as int32 bits).  Rows [0,n_owned) are this rank's particles, the rest are
 * ghost copies: read as neighbours, never updated, excluded from reductions.  n_total <= params.n_liquid. */
int wcsph_load_state(wcsph_ctx *ctx, int32_t n_owned, int32_t n_total, const float *d_rows);
/* the same from two row blocks (owned rows, then ghost rows): saves the driver a concatenation of the whole state */
int wcsph_load_state_split(wcsph_ctx *ctx, int32_t n_owned, const float *d_owned_rows, int32_t n_ghost, const float *d_ghost_rows);
int wcsph_store_state(wcsph_ctx *ctx, float *d_rows /* device [n_owned][14] */);
/* reduction slots of the running phase (double): 0 density-error sum, 1 d.Ad, 2 r.s, 3 delta_zero */
int wcsph_get_acc(wcsph_ctx *ctx, int slot, double *v);   /* synchronises */
int wcsph_set_acc(wcsph_ctx *ctx, int slot, double v);
/* One launch group of the DFSPH step without any host synchronisation (ids: enum wcsph_phase).  The fused
 * entry points above are sequences of these; a slab driver interleaves halo exchanges and all-reduces. */
enum wcsph_phase {
  WCSPH_PH_DENSITY_COFF = 0,  /* dfsph.py:248-262 + 345-372 in one walk */
  WCSPH_PH_DRHO_DIV_WARM,     /* dfsph.py:417-418 */
  WCSPH_PH_VEL_WARM,          /* dfsph.py:420-439 */
  WCSPH_PH_DRHO_DIV_BEGIN,    /* dfsph.py:441-446 */
  WCSPH_PH_MAKE_K_DIV,        /* dfsph.py:454-458 (k_i, kappa_v += k_i) */
  WCSPH_PH_VEL_ITER,          /* dfsph.py:460-471 / 529-541 */
  WCSPH_PH_DRHO_DIV_ITER,     /* dfsph.py:474-477, slot 0 <- local sum */
  WCSPH_PH_END_DIV,           /* dfsph.py:480-484 */
  WCSPH_PH_WARM_PRESS_SCALE,  /* dfsph.py:489-490 */
  WCSPH_PH_WARM_PRESS_VEL,    /* dfsph.py:492-508 */
  WCSPH_PH_DRHO_PRESS_BEGIN,  /* dfsph.py:511-516 */
  WCSPH_PH_MAKE_K_PRESS,      /* dfsph.py:523-527 */
  WCSPH_PH_DRHO_PRESS_ITER,   /* dfsph.py:544-547, slot 0 <- local sum */
  WCSPH_PH_END_PRESS,         /* dfsph.py:549-552 */
  WCSPH_PH_CLEAR,             /* dfsph.py:333-336 */
  WCSPH_PH_TENSION_NORMAL,    /* dfsph.py:266-279 */
  WCSPH_PH_TENSION_ACCEL,     /* dfsph.py:281-305 */
  WCSPH_PH_VISC_GUESS,        /* dfsph.py:199-200 */
  WCSPH_PH_VISC_MINV,         /* dfsph.py:202-215 */
  WCSPH_PH_VISC_RESIDUAL,     /* dfsph.py:217-223, slot 3 <- local sum */
  WCSPH_PH_VISC_AD,           /* dfsph.py:227-230, slot 1 <- local sum */
  WCSPH_PH_VISC_UPDATE,       /* dfsph.py:233-240, reads slot 1 and scalar cg_delta, slot 2 <- local sum */
  WCSPH_PH_VISC_DIR,          /* dfsph.py:243-246, reads slot 2 and scalar cg_delta (as delta_old) */
  WCSPH_PH_VISC_END,          /* dfsph.py:339-343 */
  WCSPH_PH_VORTICITY,         /* dfsph.py:309-327 */
  WCSPH_PH_VORTICITY_INT,     /* dfsph.py:329-330 */
  WCSPH_PH_UPDATE_VEL,        /* dfsph.py:572-575 */
  WCSPH_PH_UPDATE_POS,        /* dfsph.py:577-580 */
  WCSPH_PH_DENSITY_COFF_WARM, /* DENSITY_COFF + DRHO_DIV_WARM in one walk (neither reads a neighbour field the other writes) */
  WCSPH_PH_VISC_MINV_RESIDUAL, /* VISC_MINV + VISC_RESIDUAL in one walk, slot 3 <- local sum */
  WCSPH_PH_VISC_MINV_RESIDUAL_VORT, /* ... + the sums of VORTICITY (reads vel_j, omega_j, rho_j: unchanged by the viscosity solve) */
  WCSPH_PH_END_DIV_BEGIN_NONPRESSURE, /* END_DIV + CLEAR + VISC_GUESS in one elementwise launch (dfsph.py:480-484, 333-336, 199-200) */
  WCSPH_PH_UPDATE_VEL_WARM_SCALE,     /* UPDATE_VEL + WARM_PRESS_SCALE (dfsph.py:572-575, 489-490) */
  WCSPH_PH_END_PRESS_UPDATE_POS,      /* END_PRESS + UPDATE_POS (dfsph.py:549-552, 577-580) */
  WCSPH_PH__COUNT
};
int wcsph_phase(wcsph_ctx *ctx, int phase);
/* Use a caller-owned device buffer of 8 x 64 doubles (slot s = doubles [64 s, 64 s + 64), striped partial sums) for the
 * reduction slots, so that a slab driver can all-reduce a slot IN PLACE on the device (torch.distributed on the
 * context's stream): elementwise SUM over ranks of the stripes leaves the global total to every reader.  The buffer
 * must outlive the context's use of it. */
int wcsph_set_acc_buffer(wcsph_ctx *ctx, void *device_doubles);
/* Halo/compute overlap: restrict the NEIGHBOUR passes launched next to a part of the owned particles.  mode 0: all;
 * 1: only particles in global cell columns [x_lo, x_hi) -- the slab interior, whose 5^3 cells hold no ghost, so the
 * pass may run while a halo exchange is in flight; 2: only the others (run after the exchange; reduction slots are
 * not cleared again).  Elementwise phases ignore it. */
int wcsph_set_part(wcsph_ctx *ctx, int32_t mode, int32_t x_lo, int32_t x_hi);

/* ---- the whole DFSPH step of one slab in ONE call (wcsph_comm_init / wcsph_halo_exchange) -----------------
 * The choreography of the reference main-loop body (dfsph.py:606-617) on a slab -- every phase, the halo refresh of
 * exactly the field the next phase reads from neighbours j, the overlap of that refresh with the slab interior, the
 * solver loops and their stopping rules (dfsph.py:95-99, 141-144, 160-163), the dt rule (:116-129) -- runs in the
 * library; the COMMUNICATOR stays the caller's: at every exchange point the library calls back.  A callback returns 0 or
 * a negative error code (which wcsph_slab_step returns).  All device work of the callbacks must be ordered on the
 * context's stream (wcsph_set_stream), as the library's own launches are.
 *   halo_buffers(user, W, &send, &recv): device buffers for rows of W floats: send [n_send][W], recv [n_ghost][W]
 *   halo_post(user, W):  send holds rows send_idx of the fields being refreshed: post the sends / receives (no wait)
 *   halo_wait(user):     return when recv holds the neighbours' rows (as far as the stream is concerned)
 *   allreduce_acc(user, slot): SUM over ranks of the 64 stripes of reduction slot `slot`, in place in the buffer given to
 *                        wcsph_set_acc_buffer() (kernels launched next read the global total)
 *   allreduce_max(user, &v): MAX over ranks of a host double (the CFL velocity maximum, dfsph.py:113-114)
 *   allreduce_words(user, d_words, n): SUM over ranks of n int32 words on the device (the packed bucket histogram)
 *   allreduce_max_bits(user, d_word): MAX over ranks of ONE int32 word on the device, in place -- the bit pattern of the
 *                        non-negative f32 CFL velocity maximum (bit patterns of non-negative floats order like the floats), so
 *                        the global maximum costs no host round trip of its own: the host reads it with the wait it needs
 *                        anyway.  May be NULL: the library then reads the local maximum and calls allreduce_max on the host
 * world <= 1 or cb == NULL: no callback is made (one slab = the whole domain). */
typedef struct wcsph_slab_callbacks {
  void *user;
  int32_t world;
  int (*halo_buffers)(void *user, int32_t W, void **d_send, void **d_recv);
  int (*halo_post)(void *user, int32_t W);
  int (*halo_wait)(void *user);
  int (*allreduce_acc)(void *user, int32_t slot);
  int (*allreduce_max)(void *user, double *v);
  int (*allreduce_words)(void *user, void *d_words, int64_t n);
  int (*allreduce_max_bits)(void *user, void *d_word);
  /* the packed bucket histogram's all-reduce in two halves, so that it is in flight while the library scans, sorts and builds
   * the lists (which read none of it): _post starts it (ordered behind the work already on the context's stream), _wait makes
   * the context's stream wait for it.  Both NULL: allreduce_words (blocking, on the critical path) is used instead */
  int (*allreduce_words_post)(void *user, void *d_words, int64_t n);
  int (*allreduce_words_wait)(void *user);
} wcsph_slab_callbacks;
/* the rows this rank sends (original indices of OWNED rows, left neighbour's first) and the ghost rows it receives into,
 * both device int32 arrays that stay valid until the next call; after every wcsph_load_state*() */
int wcsph_slab_set_halo(wcsph_ctx *ctx, const int32_t *d_send_idx, int32_t n_send, const int32_t *d_ghost_idx, int32_t n_ghost);
/* n_global: liquid particles of ALL slabs (error thresholds are per global particle, dfsph.py:143, 162);
 * interior [x_lo, x_hi): global cell columns whose particles have no ghost within reach (wcsph_set_part);
 * flags: bit 0 = overlap halo refreshes with the interior, bit 1 = surface tension passes on, bit 2 = no speculation (one host
 * wait per solver decision, the control flow of rounds 1-3), bit 3 = call the communicator at every exchange point even when
 * world == 1 (empty or loop-back halos, all-reduces over one rank: the communicator path on one GPU).
 * cb == NULL on a context with a native communicator (wcsph_comm_init): the library's own RCCL data plane is used.
 * Speculation (default; needs allreduce_max_bits): every solver decision is taken on the device from the all-reduced -- hence
 * rank-identical -- error sum and the common-case continuation is queued behind it in gated segments, halo exchanges included
 * (a gated-off exchange moves a stale buffer that nobody unpacks): the GPU runs the step without waiting for the host.
 * Afterwards: wcsph_get_scalar(DELTA_T), wcsph_get_iters(), stats[0] = halo exchanges, stats[1] = host waits of this step
 * with the GPU idle behind them (1 in the common case: the step's last mail). */
int wcsph_slab_step(wcsph_ctx *ctx, const wcsph_slab_callbacks *cb, int64_t n_global, int32_t interior_lo, int32_t interior_hi,
                    int32_t flags, int32_t stats[2]);

/* ---- slab state resident in the library (round 5) -------------------------------------------------------------------
 * Instead of storing every row after a step and loading every row before the next (wcsph_store_state / wcsph_load_state*:
 * 2 x 120 MB per step at C3), the driver leaves the state where it is and only handles the rows that travel:
 *   wcsph_state_classify(ctx, lo, hi, ghost, has_left, has_right, d_idx, cap, d_counts): wcsph_slab_classify over the context's
 *        OWN owned rows (original indices), by their current positions
 *   wcsph_state_gather_rows(ctx, d_idx, n, d_rows): the STATE_W-float rows of the listed original indices
 *   wcsph_state_apply(ctx, n_owned_new, d_fold_dst, d_fold_src, n_fold, d_write_idx, d_write_rows, n_write, d_ghost_rows, n_ghost):
 *        original row fold_src[k] is relabelled fold_dst[k] (a staying row of the shrinking table's tail takes the index of a row
 *        that left; no data moves); then d_write_rows[k] becomes original row write_idx[k] (arrivals: into the slots of rows that
 *        left, or appended); then the n_ghost ghost rows become original rows n_owned_new ... (replacing the old ghosts).  The
 *        arrays then have holes; the next grid update (wcsph_slab_step) drops them -- nothing else may run in between.
 * (all index arrays int64 on the device, as wcsph_slab_classify returns them) */
/* (round 6) the same classification taken in the TAIL of wcsph_slab_step, on the positions the step has just written: register the
 * parameters and buffers of wcsph_state_classify once (d_idx == NULL: off; the tail fills d_idx, its counts go to the host only -- d_counts
 * stays wcsph_state_classify's); when a step ends on its speculative path -- the common
 * case -- the counts travel with its last mail, and wcsph_slab_tail_classify_result returns them (WCSPH_OK, consumed) without a
 * host wait of its own; 2 = the step ended on its plain path: the counts are returned after this call has waited for them;
 * 1 = none available (the caller runs wcsph_state_classify as before) */
int wcsph_slab_set_tail_classify(wcsph_ctx *ctx, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right, int64_t *d_idx, int32_t cap, int32_t *d_counts);
int wcsph_slab_tail_classify_result(wcsph_ctx *ctx, int32_t counts[5]);
int wcsph_state_classify(wcsph_ctx *ctx, int32_t lo, int32_t hi, int32_t ghost, int32_t has_left, int32_t has_right, int64_t *d_idx, int32_t cap, int32_t *d_counts);
int wcsph_state_gather_rows(wcsph_ctx *ctx, const int64_t *d_idx, int32_t n, float *d_rows);
int wcsph_state_apply(wcsph_ctx *ctx, int32_t n_owned_new, const int64_t *d_fold_dst, const int64_t *d_fold_src, int32_t n_fold,
                      const int64_t *d_write_idx, const float *d_write_rows, int32_t n_write, const float *d_ghost_rows, int32_t n_ghost);

/* (round 6) the whole migration + ghost rebuild of a resident slab state inside the library (csrc/api_migrate.h; no reference
 * counterpart, dfsph.py:606-617 is single-device).  counts[5] / d_idx: a classification of the present state, as
 * wcsph_state_classify / the step's tail leave it (lists 0..3 = near_left, near_right, go_left, go_right; 4 lists of `cap` int64).
 *   wcsph_slab_migrate_begin(ctx, d_idx, cap, counts, &d_to_left, &d_to_right): the rows that travel, gathered into the library's
 *        own send buffer: to_left = [go_left | near_left] (counts[2] + counts[0] rows of STATE_W floats), to_right = [go_right | near_right]
 *   wcsph_slab_migrate_end(ctx, d_from_left, n_from_left, a_left, d_from_right, n_from_right, a_right, out): what the neighbours
 *        sent; the first a_left / a_right rows of a message change owner to this rank, the others are its new ghosts.  Computes the
 *        swap plan on the device (arrivals into the slots of what left, the tail folded by a swap of labels), writes the rows,
 *        builds the halo lists of the coming step in the library's own buffers and installs them (wcsph_slab_set_halo; on a
 *        native communicator wcsph_comm_set_halo_split too).  out[5] = {n_owned_new, n_send_left, n_send_right, n_ghost_left,
 *        n_ghost_right}.  Five launches, no host wait; as after wcsph_state_apply only a grid update / wcsph_slab_step may follow.
 *   wcsph_slab_halo_lists(ctx, &d_send_idx, &n_send, &d_ghost_idx, &n_ghost): the installed lists (int32, device)
 *   wcsph_slab_tail_neighbour_counts(ctx, left, right): native data plane: the slab neighbours' counts of the classification
 *        taken in the last step's tail (exchanged by one grouped ncclSend / ncclRecv of five words behind it, mailed with it);
 *        WCSPH_OK, or 1 = none available.  A missing neighbour's counts are zero.
 *   wcsph_slab_migrate(ctx, d_idx, cap, counts, nbr_left, nbr_right, out): begin + ONE grouped ncclSend / ncclRecv of exact sizes
 *        on the library's communicator + end; nothing on the host in between
 *   wcsph_slab_migrate_set_seam(ctx, shift_left, shift_right, ghost_gid_base): a periodic seam (a ring of slabs): x of what
 *        arrives from the left / right is shifted; ghost_gid_base >= 0 numbers the ghost copies from there (-1: keep their ids) */
int wcsph_slab_migrate_begin(wcsph_ctx *ctx, const int64_t *d_idx, int32_t cap, const int32_t counts[5], void **d_to_left, void **d_to_right);
int wcsph_slab_migrate_end(wcsph_ctx *ctx, const void *d_from_left, int32_t n_from_left, int32_t a_left,
                           const void *d_from_right, int32_t n_from_right, int32_t a_right, int32_t out[5]);
int wcsph_slab_halo_lists(wcsph_ctx *ctx, void **d_send_idx, int32_t *n_send, void **d_ghost_idx, int32_t *n_ghost);
int wcsph_slab_tail_neighbour_counts(wcsph_ctx *ctx, int32_t left[5], int32_t right[5]);
int wcsph_slab_migrate(wcsph_ctx *ctx, const int64_t *d_idx, int32_t cap, const int32_t counts[5], const int32_t nbr_left[5],
                       const int32_t nbr_right[5], int32_t out[5]);
int wcsph_slab_migrate_set_seam(wcsph_ctx *ctx, float shift_left, float shift_right, int32_t ghost_gid_base);

/* ---- native data plane (wcsph_comm_init / wcsph_halo_exchange) -----------------------------------------
 * The library's own implementation of the communicator above on RCCL: grouped ncclSend / ncclRecv with the slab neighbours
 * and ncclAllReduce, no host code per exchange.  Inside wcsph_slab_step a halo refresh runs on the communicator's own stream,
 * fenced by events against the context's (pack -> send/recv -> unpack), so that the pass on the slab interior queued between
 * post and wait runs while the rows travel; the all-reduces of the reduction slots are on the context's stream.  New work -- the reference step is single-device
 * (dfsph.py:606-617).  RCCL is loaded at run time (the copy already in the process, else ROCm's librccl.so.1); without it
 * wcsph_comm_init fails with WCSPH_EHIP and the caller keeps its own communicator behind wcsph_slab_callbacks.
 *   wcsph_comm_unique_id(id):  rank 0 fills the 128-byte ncclUniqueId; the caller hands it to every rank (any channel)
 *   wcsph_comm_last_error():   why the last wcsph_comm_unique_id() of this process failed (it has no context to leave a message in)
 *   wcsph_comm_init(ctx, id, rank, nranks): ncclCommInitRank on the calling thread's current device (the context's); slab
 *                              neighbours rank - 1 / rank + 1.  wcsph_slab_step(ctx, NULL, ...) then uses this communicator
 *   wcsph_comm_set_neighbours: other neighbour ranks, -1 = none; left == right == rank makes a LOOP-BACK communicator (what is
 *                              sent to the left arrives from the right, as on a ring of one): the one-GPU test of the data plane
 *   wcsph_comm_set_mode(ctx, m): where the halo refreshes of wcsph_slab_step run: 2 = exchange, unpack and the pass on the rows
 *                              next to the cuts on the communicator's own stream, beside the pass on the slab interior, one join at
 *                              the end; 1 (default) = only the exchange there (the context's stream waits for it before the unpack) -- both
 *                              with flags bit 0 of wcsph_slab_step; 0 = inline on the context's stream between pack and unpack
 *                              (no event hops; for steps whose passes are not split)
 *   wcsph_comm_set_halo_split: of the lists given to wcsph_slab_set_halo, the first n_send_left rows go to / the first
 *                              n_ghost_left rows come from the left neighbour, the others the right one; after every wcsph_slab_set_halo
 *   wcsph_halo_exchange(ctx, W, d_send, n_send, d_recv, n_ghost): ONE grouped exchange of rows of W floats, stream-ordered on the
 *                              context's stream (the caller's buffers); WCSPH_EARG when the split does not fit n_send / n_ghost
 *   wcsph_comm_allreduce(ctx, d_buf, n, op): in place; op 0 = int32 SUM, 1 = float64 SUM, 2 = int32 MAX
 *   wcsph_comm_stats(ctx, out): out[0] = exchanges, out[1] = all-reduces issued so far */
#define WCSPH_COMM_ID_BYTES 128
int wcsph_comm_unique_id(void *id_out);
const char *wcsph_comm_last_error(void);
int wcsph_comm_init(wcsph_ctx *ctx, const void *unique_id, int32_t rank, int32_t nranks);
int wcsph_comm_destroy(wcsph_ctx *ctx);
int wcsph_comm_set_neighbours(wcsph_ctx *ctx, int32_t left, int32_t right);
int wcsph_comm_set_mode(wcsph_ctx *ctx, int32_t mode);
int wcsph_comm_set_halo_split(wcsph_ctx *ctx, int32_t n_send_left, int32_t n_ghost_left);
int wcsph_halo_exchange(wcsph_ctx *ctx, int32_t W, const void *d_send, int32_t n_send, void *d_recv, int32_t n_ghost);
int wcsph_comm_allreduce(wcsph_ctx *ctx, void *d_buf, int64_t n, int32_t dtype_op);
int wcsph_comm_stats(wcsph_ctx *ctx, int64_t out[2]);
/* hand a communicator from one context of this process to another (a rebalance rebuilds the context: new cell window) */
int wcsph_comm_detach(wcsph_ctx *ctx, void **handle);
int wcsph_comm_attach(wcsph_ctx *ctx, void *handle);

/* gather / scatter `n` rows of a field by ORIGINAL liquid indices (device int32 array) into / from a
 * packed device buffer (rows of 3 f32 for vec3 fields, 1 f32 for scalar fields) for halo exchange */
int wcsph_field_gather_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, void *d_dst);
int wcsph_field_scatter_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, const void *d_src);
/* mark liquid particles [n_owned, n_liquid) as ghosts: they are read as neighbours but never updated,
 * and are excluded from reductions */
int wcsph_set_owned(wcsph_ctx *ctx, int32_t n_owned);

#ifdef __cplusplus
}
#endif
#endif /* WCSPH_H */
