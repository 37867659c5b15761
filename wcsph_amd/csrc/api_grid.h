// api_grid.h -- HashGrid.update_grid (ref: HashGrid.py:56-114) on the kernels of grid.h, and the reference bucket table on demand (bucket.h).
#pragma once
extern "C" {

// ---- HashGrid.update_grid -------------------------------------------------------------------
// first half: cells, counting sort, this context's share of the reference bucket histogram (owned liquid particles)
// this step's liquid bucket histogram of the owned rows (ref: HashGrid.py:67-76), from the packed cells k_cell_assign left:
// binned per workgroup and counted in LDS (grid.h k_hash_bin / k_bin_hist), atomics when the table is too large for that
static int bucket_hist_launches(wcsph_ctx *ctx, hipStream_t s2) {
  Dev &d = ctx->d;
  // (a resident slab state: owned rows are not a prefix of the arrays -- every row is looked at, ownership by its original index)
  const int *live = ctx->nl_phys > 0 ? (const int *)d.oid : (const int *)nullptr;
  const int nh = live ? ctx->nl_phys : (d.nl < d.n_owned ? d.nl : d.n_owned);
  if (nh > 0 && ctx->hb_bins) {
    if (ctx->hb_gen == 0x7fffffff) { HIPCHK(hipMemsetAsync(ctx->hb_bins + HB_FLAG, 0, sizeof(int), s2)); ctx->hb_gen = 0; }   // (wrap-around)
    const int gen = ++ctx->hb_gen;
    hipLaunchKernelGGL(k_hash_bin, dim3((nh + BLOCK * HB_ROWS - 1) / (BLOCK * HB_ROWS)), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh,
                       ctx->hb_nbins, ctx->hb_cap, ctx->hb_bins, ctx->hb_buf, gen, live);
    hipLaunchKernelGGL(k_bin_hist, dim3(ctx->hb_nbins), dim3(1024), 0, s2, d, ctx->hb_cap, ctx->hb_bins, (const int *)ctx->hb_buf, ctx->bcount_l, gen);
    hipLaunchKernelGGL(k_bucket_hist_gated, dim3(nblk(nh) < 512 ? nblk(nh) : 512), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh, ctx->bcount_l, ctx->hb_nbins, ctx->hb_bins, gen, live);
  } else if (nh > 0) hipLaunchKernelGGL(k_bucket_hist, dim3(nblk(nh)), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh, ctx->bcount_l, live);
  return 0;
}
// The bucket histogram is not taken inside k_cell_assign (its 2 M random device-scope atomics were 71 of that kernel's 88 us at
// C3) but by kernels of its own.  (Rounds 3-4 ran them, the bucket sums and the neighborCount box sums on a side stream beside the
// scan, the sort and the list build.  Since the table is updated in place that side work is two kernels, 57 us alone at C3, and the two
// cross-stream waits of the fork cost more than the overlap hides: 8 000 particles 0.217 -> 0.203 ms per step without it, C2 0.252 ->
// 0.243, C3 1.653 -> 1.646 at rest and 3.30 -> 3.28 developed -- profiles/r04u_busy_small.txt, r04u_ab_fork_threshold.txt,
// r04u_ab_fork_c3.txt.  One stream now.)
static void neighbor_count_launches(wcsph_ctx *ctx, hipStream_t s, bool box);
// HashGrid.gridCount as ints (bucket rows, gridCount.to_numpy()): current after a rebuilding update, expanded from the byte table
// after an incremental one
static int ensure_bcount_int(wcsph_ctx *ctx) {
  if (ctx->bcount_int_valid || !ctx->d.faithful) return 0;
  hipLaunchKernelGGL(k_bucket_expand, dim3(nblk(ctx->d.table_size)), dim3(BLOCK), 0, ctx->stream, (const unsigned char *)ctx->bcount8, ctx->bcount, ctx->d.table_size);
  LAUNCHCHK();
  ctx->bcount_int_valid = true;
  return 0;
}
// The update in four parts.  ONE context: assign -> sort -> counts -> lists.  A slab of a decomposition in faithful
// mode: assign [+ this rank's share of the bucket histogram] -> the histogram's all-reduce is POSTED -> sort + lists run while it
// is in flight (they read neither the histogram nor neighborCount) -> the all-reduce is awaited -> counts (wcsph_slab_step).
static int grid_assign(wcsph_ctx *ctx, bool single = false) {   // single: wcsph_grid_update() of ONE context, not a slab's halves
  Dev &d = ctx->d;
  hipStream_t s = ctx->stream;
  // ctx->cnt is all zero here: k_scan_final clears the counters it consumes.  The error flags are NOT cleared here: a
  // condition stays on the device until k_publish has handed it to the host once (see pull_scalars)
  ctx->pr_valid = false;   // rows are re-sorted
  // the liquid bucket histogram and the region flags are zero here: their consumers (k_bucket_sum, k_ncount_box) clear
  // them -- unless the previous update did not run to its end
  if (ctx->hist_dirty) {
    HIPCHK(hipMemsetAsync(ctx->region_occ, 0, (size_t)ctx->n_regions * sizeof(int), s));
    if (d.faithful) HIPCHK(hipMemsetAsync(ctx->bcount_l, 0, (size_t)d.table_size * sizeof(int), s));
    if (d.faithful && ctx->hb_bins) HIPCHK(hipMemsetAsync(ctx->hb_bins, 0, (HB_MAXBINS + 1) * sizeof(int), s));
  }
  ctx->hist_dirty = true;
  // One faithful context between two of its own complete updates: the byte table of the bucket occupancy is UPDATED by
  // k_cell_assign (particles that changed their cell: ~5 % of them per step at rest, 10-20 % in the developed flow) instead of
  // being rebuilt by the histogram kernels + k_bucket_sum (50 us alone at C3).  Anything else -- the first update, a slab of a
  // decomposition, an update that did not run to its end, a cap a byte cannot hold with a margin -- rebuilds it.
  // (the default owned column range: a rebuild counts only particles inside [own_lo, own_hi), wcsph_set_owned_columns; the in-place
  // update moves every row's count.  In-place mode checks the cap where particles are, in and next to occupied regions
  // (k_cellval_occ); a rebuild checks every bucket (k_bucket_sum).)
  ctx->hist_inc = single && d.faithful && ctx->hist_valid && d.max_in_grid <= 200 && d.n_owned >= d.nl && ctx->win_x1 < 0 &&
                  d.own_lo <= 0 && d.own_hi >= ctx->bs[0];
  ctx->hist_valid = false;
  if (ctx->hist_inc) { ctx->bcount_int_valid = false; ctx->n_hist_inc++; }
  const int n_rows = ctx->nl_phys > 0 ? ctx->nl_phys : d.nl;   // (a resident slab state has holes: api_slab.h wcsph_state_apply)
  if (n_rows > 0)
    hipLaunchKernelGGL(k_cell_assign, dim3(nblk(n_rows)), dim3(BLOCK), 0, s, d, (const float4 *)d.pos, n_rows, ctx->cnt, ctx->pw, ctx->slot,
                       (int *)nullptr, ctx->bbox_part, d.n_owned,
                       ctx->region_occ, ctx->hist_inc ? reinterpret_cast<unsigned *>(ctx->bcount8) : (unsigned *)nullptr,
                       ctx->nl_phys > 0 ? (const int *)d.oid : (const int *)nullptr);
  if (d.faithful && !ctx->hist_inc) { int rc_ = bucket_hist_launches(ctx, s); if (rc_) return rc_; }   // (a slab: before the all-reduce)
  LAUNCHCHK();
  ctx->grid_stage = 1;
  return 0;
}
// the counting sort: occupied box, exclusive scan of the cell counters, every persistent field to its cell-sorted place
// (grid_sort and grid_lists must not touch ctx->bcount: on a slab the packed bucket histogram is being all-reduced in it while they
// are queued -- wcsph_slab_step, allreduce_words_post / _wait)
static int grid_sort(wcsph_ctx *ctx) {
  Dev &d = ctx->d;
  const int nscan = d.ncell_p + 2;
  hipStream_t s = ctx->stream;
  const int n_rows = ctx->nl_phys > 0 ? ctx->nl_phys : d.nl;
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(BLOCK), 0, s, (const int *)ctx->bbox_part, n_rows > 0 ? nblk(n_rows) : 0, d.bbox);
  int rc = run_scan(ctx, ctx->cnt, nscan, d.cs_l, d.bbox, 1);
  if (rc) return rc;
  if (d.nl > 0) {
    SortBuffers b = {d.pos, d.vel, d.vel_guess, d.omega, d.kappa, d.kappa_v, d.pressure, d.oid,
                     ctx->pos_b, ctx->vel_b, ctx->vel_guess_b, ctx->omega_b, ctx->kappa_b, ctx->kappa_v_b, ctx->pressure_b, ctx->oid_b, d.where};
    hipLaunchKernelGGL(k_scatter_liquid, dim3(nblk(n_rows)), dim3(BLOCK), 0, s, d, b, (const int *)d.cs_l, (const int *)ctx->pw,
                       (const int *)ctx->slot, n_rows);
    std::swap(d.pos, ctx->pos_b); std::swap(d.vel, ctx->vel_b); std::swap(d.vel_guess, ctx->vel_guess_b);
    std::swap(d.omega, ctx->omega_b); std::swap(d.kappa, ctx->kappa_b); std::swap(d.kappa_v, ctx->kappa_v_b);
    std::swap(d.oid, ctx->oid_b); std::swap(d.pressure, ctx->pressure_b);
  }
  LAUNCHCHK();
  ctx->nl_phys = 0;   // the sorted arrays are compact
  ctx->grid_stage = 2;
  return 0;
}
// second half: raw neighbour-list length (HashGrid.neighborCount) from the GLOBAL bucket occupancy, compact lists
// faithful mode: box = false: bucket sums and gridCount[hash(cell)] of the occupied regions; box = true: neighborCount, the
// 5x5x5 box sum of gridCount[hash(cell)] around every occupied cell, one LDS-tiled launch (grid.h)
static void neighbor_count_launches(wcsph_ctx *ctx, hipStream_t s, bool box) {
  Dev &d = ctx->d;
  const int nb = ctx->n_regions;
  if (!box) {
    if (!ctx->hist_inc) {   // (an incremental update has left the byte table current: grid_assign)
      hipLaunchKernelGGL(k_bucket_sum, dim3(nblk((d.table_size + 3) / 4)), dim3(BLOCK), 0, s, d, (const int *)ctx->bcount_s, ctx->bcount_l, ctx->bcount, ctx->bcount8, d.flags);
      ctx->bcount_int_valid = true;
    }
    if (d.nl > 0) {
      if (d.max_in_grid <= 255)
        hipLaunchKernelGGL(k_cellval_occ<unsigned char>, dim3(nb), dim3(BLOCK), 0, s, d, (const unsigned char *)ctx->bcount8, (const int *)ctx->region_occ, ctx->cellval);
      else
        hipLaunchKernelGGL(k_cellval_occ<int>, dim3(nb), dim3(BLOCK), 0, s, d, (const int *)ctx->bcount, (const int *)ctx->region_occ, ctx->cellval);
    }
  } else if (d.nl > 0) {
    hipLaunchKernelGGL(k_ncount_box, dim3(nb), dim3(BLOCK), 0, s, d, (const int *)ctx->cellval, (const int *)ctx->region_occ);
  }
}
// raw neighbour-list length (HashGrid.neighborCount)
static int grid_counts(wcsph_ctx *ctx) {
  Dev &d = ctx->d;
  if (d.faithful) { neighbor_count_launches(ctx, ctx->stream, false); neighbor_count_launches(ctx, ctx->stream, true); }
  else if (d.nl > 0) {   // clean: the same box sums over the cells' own occupancy (no hash table)
    hipLaunchKernelGGL(k_cellval_clean, dim3(ctx->n_regions), dim3(BLOCK), 0, ctx->stream, d, (const int *)ctx->region_occ, ctx->cellval);
    hipLaunchKernelGGL(k_ncount_box, dim3(ctx->n_regions), dim3(BLOCK), 0, ctx->stream, d, (const int *)ctx->cellval, (const int *)ctx->region_occ);
  }
  LAUNCHCHK();
  return 0;
}
// the compact per-step lists (they read the sorted positions and the cell table, nothing of the bucket histogram)
static int grid_lists(wcsph_ctx *ctx) {
  Dev &d = ctx->d;
  if (d.nl > 0) hipLaunchKernelGGL(k_build_lists, dim3(nblk8(d.nl)), dim3(BLOCK), 0, ctx->stream, d, 1);
  LAUNCHCHK();
  ctx->grid_stage = 3;
  return 0;
}
static int grid_finish(wcsph_ctx *ctx, bool single) {
  ctx->d.rev ^= 1;
  ctx->hist_dirty = false;   // k_bucket_sum / k_ncount_box have left the histogram and the region flags zeroed
  ctx->hist_valid = single && ctx->d.faithful && ctx->d.nl > 0;   // the byte table matches the cells the sort has just written to pos.w: the next update may update it in place
  ctx->hist_inc = false;
  ctx->grid_valid = true;
  ctx->bk_valid = false;
  ctx->grid_stage = 0;
  return WCSPH_OK;
}
int wcsph_grid_update(wcsph_ctx *ctx) {
  NEED_SETUP();
  int rc = grid_assign(ctx, true);
  PROF_LAP(ctx, "grid:cells");
  if (!rc) rc = grid_sort(ctx);
  PROF_LAP(ctx, "grid:scan+sort");
  if (!rc) rc = grid_counts(ctx);
  PROF_LAP(ctx, "grid:neighborCount");
  if (!rc) rc = grid_lists(ctx);
  PROF_LAP(ctx, "k_build_lists");
  if (!rc) rc = grid_finish(ctx, true);
  return rc;
}
// slab decomposition, faithful neighbour mode: the drivers all-reduce the liquid bucket histogram between the halves
// begin: cells + this rank's histogram; mid: sort + lists (to be queued while the all-reduce is in flight; optional -- end runs
// whatever has not run yet); end: neighborCount from the global histogram
int wcsph_grid_update_begin(wcsph_ctx *ctx) { NEED_SETUP(); ctx->grid_valid = false; return grid_assign(ctx); }
int wcsph_grid_update_mid(wcsph_ctx *ctx) {
  NEED_SETUP();
  if (ctx->grid_stage != 1) return ctx->fail(WCSPH_EARG, "wcsph_grid_update_mid() follows wcsph_grid_update_begin()");
  int rc = grid_sort(ctx);
  return rc ? rc : grid_lists(ctx);
}
int wcsph_grid_update_end(wcsph_ctx *ctx) {
  NEED_SETUP();
  int rc = 0;
  if (ctx->grid_stage == 0) return ctx->fail(WCSPH_EARG, "wcsph_grid_update_end() without wcsph_grid_update_begin()");
  if (ctx->grid_stage == 1 && (rc = grid_sort(ctx))) return rc;
  if ((rc = grid_counts(ctx))) return rc;
  if (ctx->grid_stage == 2 && (rc = grid_lists(ctx))) return rc;
  return grid_finish(ctx, false);
}
int wcsph_bucket_hist_pack(wcsph_ctx *ctx, void **d_ptr, int64_t *n_words) {   // liquid histogram, two 16-bit counts per word
  NEED_SETUP();
  if (!d_ptr || !n_words) return WCSPH_EARG;
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "the bucket histogram exists in faithful neighbour mode only");
  const int T = ctx->d.table_size, W = (T + 1) / 2;
  // per-rank clamp max_in_grid + 1: enough to trip the cap after the sum, and world * (max_in_grid + 1) stays below 2^16
  // for every sane cap (a carry into the neighbouring bucket's half needs max_in_grid > 8190 at 8 ranks)
  const int clamp = ctx->d.max_in_grid + 1 < 0xffff ? ctx->d.max_in_grid + 1 : 0xffff;
  hipLaunchKernelGGL(k_hist_pack16, dim3(nblk(W)), dim3(BLOCK), 0, ctx->stream, (const int *)ctx->bcount_l, T, ctx->bcount, clamp);   // bcount is rebuilt by k_bucket_sum
  LAUNCHCHK();
  *d_ptr = ctx->bcount; *n_words = W;
  return WCSPH_OK;
}
int wcsph_bucket_hist_unpack(wcsph_ctx *ctx) {
  NEED_SETUP();
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "the bucket histogram exists in faithful neighbour mode only");
  const int T = ctx->d.table_size, W = (T + 1) / 2;
  // bcount holds the summed words, bcount_l receives the global counts
  hipLaunchKernelGGL(k_hist_unpack16, dim3(nblk(W)), dim3(BLOCK), 0, ctx->stream, (const int *)ctx->bcount, T, ctx->bcount_l);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_bucket_hist(wcsph_ctx *ctx, int which, void **d_ptr, int64_t *n) {
  NEED_SETUP();
  if (!d_ptr || !n || which < 0 || which > 1) return WCSPH_EARG;
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "the bucket histogram exists in faithful neighbour mode only");
  *d_ptr = which ? ctx->bcount_s : ctx->bcount_l;
  *n = ctx->d.table_size;
  return WCSPH_OK;
}

// ---- the reference's bucket table, on demand (bucket.h) ----------------------------------------------------
static int bucket_build(wcsph_ctx *ctx) {
  Dev &d = ctx->d;
  if (!d.faithful) return ctx->fail(WCSPH_EARG, "the reference bucket table (HashGrid.grid) exists in faithful neighbour mode only");
  if (ctx->win_x1 >= 0) return ctx->fail(WCSPH_EARG, "the reference bucket table is not available on a slab window (it spans all ranks)");
  if (ctx->bk_valid) return 0;
  const int T = d.table_size;
  hipStream_t s = ctx->stream;
  if (!ctx->bk.start) {
    DALLOC(ctx->bk.start, (size_t)T + 2); DALLOC(ctx->bk.cursor, (size_t)T + 1);
    DALLOC(ctx->bk.item, (size_t)ctx->capacity + d.ns + 1); DALLOC(ctx->bk.swhere, (size_t)d.ns + 1);
    if (d.ns > 0) hipLaunchKernelGGL(k_bk_swhere, dim3(nblk(d.ns)), dim3(BLOCK), 0, s, d, ctx->bk);
  }
  int rc = ensure_bcount_int(ctx);
  if (!rc) rc = run_scan(ctx, ctx->bcount, T + 1, ctx->bk.start, nullptr, 0);   // bcount = gridCount of this step (grid update)
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->bk.cursor, ctx->bk.start, (size_t)T * sizeof(int), hipMemcpyDeviceToDevice, s));
  if (d.nl > 0) hipLaunchKernelGGL(k_bk_scatter_liquid, dim3(nblk(d.nl)), dim3(BLOCK), 0, s, d, ctx->bk);
  if (d.ns > 0) hipLaunchKernelGGL(k_bk_scatter_solid, dim3(nblk(d.ns)), dim3(BLOCK), 0, s, d, ctx->bk);
  hipLaunchKernelGGL(k_bk_sort, dim3(nblk(T)), dim3(BLOCK), 0, s, d, ctx->bk);
  LAUNCHCHK();
  ctx->bk_valid = true;
  ctx->bk_rows_valid = false;
  return 0;
}
// + what the whole-row walks of the literal compute_tension use (bucket.h raw_row_s)
static int bucket_build_rows(wcsph_ctx *ctx) {
  int rc = bucket_build(ctx);
  if (rc || ctx->bk_rows_valid) return rc;
  Dev &d = ctx->d;
  if (!ctx->bk.coff) DALLOC(ctx->bk.coff, (size_t)d.ncell + SCAN_TILE + 2);
  hipLaunchKernelGGL(k_bk_cellcount, dim3(nblk(d.ncell + 1)), dim3(BLOCK), 0, ctx->stream, d, ctx->bk);
  rc = run_scan(ctx, ctx->bk.coff, d.ncell + 1, ctx->bk.coff, nullptr, 0);   // in place: a thread reads its eight counts before it writes them
  if (rc) return rc;
  // The item table is sized from the scanned total: every cell within 2 of the liquid's box holds a copy of its
  // bucket, i.e. about (cells of the grown box) x (particles per bucket) items -- once the liquid's box is the whole domain
  // that is the cell count, whatever the particle count.  One 4-byte read per build (the literal compute_tension only).
  int total = 0;
  HIPCHK(hipMemcpyAsync(&total, ctx->bk.coff + d.ncell, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (total < 0) return ctx->fail(WCSPH_EBUCKET, "raw-row item table of the literal compute_tension: more than 2^31 items");
  if (!ctx->bk.citem || total > ctx->bk.citem_cap) {
    if (ctx->bk.citem) {
      for (auto &a : ctx->allocs) if (a == (void *)ctx->bk.citem) a = nullptr;   // hipFree(nullptr) in wcsph_destroy is a no-op
      HIPCHK(hipFree(ctx->bk.citem));
      ctx->bk.citem = nullptr;
    }
    const size_t want = (size_t)total + (size_t)total / 4 + 1024;
    ctx->bk.citem_cap = (int)std::min<size_t>(want, (size_t)0x7ffffff0);
    DALLOC(ctx->bk.citem, (size_t)ctx->bk.citem_cap);
  }
  hipLaunchKernelGGL(k_bk_cellfill, dim3(nblk(d.ncell)), dim3(BLOCK), 0, ctx->stream, d, ctx->bk);
  LAUNCHCHK();
  ctx->bk_rows_valid = true;
  return 0;
}
int wcsph_hashgrid_count_to_host(wcsph_ctx *ctx, int32_t *dst, int64_t n) {   // HashGrid.gridCount
  NEED_GRID();
  if (!dst || n != ctx->d.table_size) return ctx->fail(WCSPH_EARG, "wcsph_hashgrid_count_to_host: n must be the hash table size");
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "HashGrid.gridCount is the reference table: faithful neighbour mode only");
  { int rc_ = ensure_bcount_int(ctx); if (rc_) return rc_; }
  HIPCHK(hipMemcpyAsync(dst, ctx->bcount, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return WCSPH_OK;
}
int wcsph_hashgrid_rows_to_host(wcsph_ctx *ctx, int32_t h0, int32_t n, int32_t cap, int32_t *rows) {   // HashGrid.grid[h0 : h0+n, :cap]
  NEED_GRID();
  if (h0 < 0 || n < 0 || (long)h0 + n > ctx->d.table_size || cap < 1 || !rows) return WCSPH_EARG;
  int rc = bucket_build(ctx);
  if (rc) return rc;
  if (n == 0) return WCSPH_OK;
  int *dr = nullptr;
  HIPCHK(hipMalloc((void **)&dr, (size_t)n * cap * 4));
  hipLaunchKernelGGL(k_bk_grid_rows, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->bk, h0, n, cap, dr);
  hipError_t e1 = hipMemcpyAsync(rows, dr, (size_t)n * cap * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(dr);
  if (e1 != hipSuccess || e2 != hipSuccess) return ctx->fail(WCSPH_EHIP, "hashgrid_rows_to_host copy");
  return WCSPH_OK;
}


}  // extern "C"
