// api_core.h -- the context behind the C ABI (include/wcsph.h): device view, streams, solver state, the mail box through which
// reduction totals, error flags and the device-side stopping decisions reach the host, and the launch macros.
#pragma once
struct NativeComm;   // api_comm.h
struct wcsph_ctx {
  wcsph_params p;
  Dev d;          // DFSPH view (CubicKernel with h = HashGrid.searchR)
  hipStream_t stream = nullptr, own_stream = nullptr;
  // faithful mode, one context: the reference bucket histogram and neighborCount (k_bucket_hist .. k_ncount_box) run on a
  // side stream beside the counting sort and the list build, which need neither (wcsph_grid_update)
  std::string err;
  bool is_setup = false, grid_valid = false;
  // double buffers of the persistent fields (the *_b set is the scatter target)
  float4 *pos_b = nullptr, *vel_b = nullptr, *vel_guess_b = nullptr, *omega_b = nullptr;
  float *kappa_b = nullptr, *kappa_v_b = nullptr, *pressure_b = nullptr;
  int *oid_b = nullptr;
  // grid scratch
  int *cnt = nullptr, *pw = nullptr, *slot = nullptr, *bsum = nullptr;
  int *hb_bins = nullptr, *hb_buf = nullptr; int hb_nbins = 0, hb_cap = 0;   // binned histogram (grid.h k_hash_bin): range counters [+ flag], range lists
  int *bcount = nullptr, *bcount_s = nullptr, *bcount_l = nullptr, *cellval = nullptr;   // gridCount = _s (static solids) + _l (liquid, per step)
  int *bbox_part = nullptr;
  int n_alias = 0, scan_blocks = 0, n_regions = 0;
  unsigned char *bcount8 = nullptr;
  int *region_occ = nullptr;      // neighborCount regions that hold particles: flagged by k_cell_assign, cleared by k_ncount_box
  int nl_phys = 0;                // > 0: the liquid arrays hold this many rows, some of them dead (a resident slab state between wcsph_state_apply and the sort that drops them)
  int grid_stage = 0;             // split grid update (api_grid.h): 1 cells assigned, 2 sorted, 3 lists built; 0 none in progress
  bool hist_valid = false;       // bcount8 == static + liquid histogram of the cells in pos.w (left by a complete update of one context): the next one may update it in place
  bool hist_inc = false;         // the update in progress does so
  int n_dt_mismatch = 0;         // speculative steps whose device-formed deltaT differed from the host rule (WCSPH_S_DT_RULE_MISMATCHES)
  int n_hist_inc = 0;            // how many updates did (WCSPH_S_TABLE_UPDATES_IN_PLACE)
  bool bcount_int_valid = false; // the int table (gridCount) matches the byte table (k_bucket_sum wrote both; k_bucket_expand on demand)
  int warm_gen = 0;              // generation number of flags[FLAG_WARM] (api_dfsph.h update_vel_warm_start)
  int hb_gen = 0;                // generation number of the binned histogram's overflow flag (grid.h k_hash_bin)
  bool hist_dirty = false;        // a grid update began and did not reach the kernels that zero bcount_l / region_occ
  BucketTab bk = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the reference's hash table itself (bucket.h), built on demand
  bool bk_valid = false, bk_rows_valid = false;
  // the classification of the NEXT step's migration / ghost rebuild, taken in the tail of wcsph_slab_step (api_slab.h): its counts
  // reach the host with the step's last mail instead of costing a host wait of their own
  struct TailClassify { bool on = false, valid = false; int lo = 0, hi = 0, ghost = 0, has_l = 0, has_r = 0, cap = 0;
                        long long *d_idx = nullptr; int *d_counts = nullptr; int *h_counts = nullptr;
                        // native data plane: the two slab neighbours' counts, exchanged by RCCL behind the classification and mailed with
                        // it (h_counts[8..13) left, [16..21) right): the next migration needs no header round trip (api_migrate.h)
                        int *d_nbr = nullptr; bool nbr_valid = false; bool nbr_queued = false;
                        unsigned long long pending_seq = 0; } tc;   // pending_seq != 0: the plain path's tail, its mail not collected yet
  // migration + ghost rebuild of a resident slab state inside the library (api_migrate.h): buffers owned here, grown on demand
  struct Migrate { float *send = nullptr, *recv = nullptr; size_t cap_send = 0, cap_recv = 0;      // rows of STATE_W floats
                   int *halo_send = nullptr, *halo_ghost = nullptr; size_t cap_hs = 0, cap_hg = 0; // the halo lists the migration leaves behind
                   int *work = nullptr; size_t cap_work = 0;
                   const long long *d_idx = nullptr; int cap = 0, c[4] = {0, 0, 0, 0};             // of the begin() in progress: near_l, near_r, go_l, go_r
                   bool begun = false;
                   float shift_l = 0.f, shift_r = 0.f; int ghost_gid_base = -1; } mg;               // periodic seam (wcsph_slab_migrate_set_seam)
  NativeComm *comm = nullptr;     // the native RCCL data plane of a slab decomposition (wcsph_comm_init), or none
  bool pr_valid = false;          // Dev::pr holds (pos, rho) of every row (dfsph_kernels.h k_pack_pr)
  // staging for original-order transfers
  float *stage = nullptr;  // [max(count*3, nl*9)] floats
  std::vector<float> solid_host;  // original-order solid positions
  double *h_acc_raw = nullptr;    // pinned [8][ACC_STRIPES]
  double h_acc[8] = {0};          // stripe totals of the last pull_scalars()
  unsigned long long mail_seq = 0;
  int h_pred = 0;                 // the device's stopping decision that came with the last collected mail (k_publish)
  int *pred_dev = nullptr;        // [4] predicate words of the speculative launches (Dev::pred points at one of them)
  bool in_step_spec = false;      // inside wcsph_dfsph_step's divergence solve: the continuation may be queued speculatively
  bool speculate = true;          // the fused step queues its common-case continuation behind the solver decisions (k_publish)
  unsigned long long spec_visc_seq = 0;   // != 0: solve_divergence queued the opening of compute_nonpressure_force behind its decision; its mail
  unsigned long long spec_cfl_seq = 0;    // != 0: ... and the end of it + the CFL reduction behind the viscosity decision; its mail
  unsigned long long spec_press_seq = 0;  // != 0: ... and update_vel + the pressure solve up to its first decision; its mail
  size_t spec_prof_mark = 0;      // prof_used before the speculative pressure iterations were queued
  bool pr_resume = false;         // solve_pressure continues a solve whose first two iterations have run (speculatively)
  bool spec_dt = false;           // the mail of the CFL reduction carries the deltaT the device formed (FLAG_DT)
  float *dt_dev = nullptr;        // deltaT as k_cfl_store formed it, for the kernels queued behind it (Dev::dtp)
  McDev mc = {};                  // MarchingCubeGrid (lazily set up)
  int *mc_tables = nullptr;       // device: edge[256] + tri[256][MC_TRI_ROW]
  size_t mc_nodes = 0;
  static constexpr bool vort_fuse = true;   // compute_nonpressure_force: the vorticity sums ride on a walk that gathers vel_j anyway
  // in-situ timing of the fused pressure_iter launches (wcsph_profile_begin/end): event pairs on the context's stream
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  bool prof_on = false;
  // lap profiling (wcsph_profile_laps_begin / _end): one event behind every tagged launch of the step; the time between two
  // consecutive events goes to the tag of the later one (stream order: the launch itself + whatever idle time preceded it)
  std::vector<hipEvent_t> lap_ev;
  std::vector<const char *> lap_tag;
  std::vector<unsigned char> lap_waited;   // the host waited for the device (a mail, a scalar pull) since the previous lap
  bool lap_on = false, lap_wait_flag = false;
  const char *lap_scope = nullptr;          // != nullptr: laps are booked to this name instead of the kernel's (the two solver iterations share k_vel_from_k<0>)
  bool in_step = false, cfl_done = false;   // wcsph_dfsph_step in progress; CFL maxima already taken
  bool warm_fused = false;        // wcsph_dfsph_step -> solve_divergence: the warm-start density-change pass is already done
  bool fuse_elem = false;         // wcsph_dfsph_step: elementwise neighbours of the launch sequence run fused (set per step)
  bool nonp_begun = false, warm_scaled = false;
  bool vort_in_div = false;       // the last divergence iteration of this step also took the vorticity sums (k_drho_div_iter_vort)
  int *h_flags = nullptr;         // pinned [4]
  // host-side solver state (ref: dfsph.py:36-41 module globals)
  float dt = 0.0f;
  float rho_err = 0.0f;
  SurfDev surf = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f};
  float avg_density_err = 0.0f, cg_delta = 0.0f, cg_delta_old = 0.0f, cg_delta_zero = 0.0f, vel_max0 = 0.0f;
  int vs_iter = 0, dv_iter = 0, pr_iter = 0;
  int fix_dv = 0, fix_pr = 0, fix_vs = 0;  // > 0: run exactly that many iterations (wcsph_dfsph_step_fixed)
  int bs[3] = {0, 0, 0};
  const int *halo_send = nullptr, *halo_ghost = nullptr; int n_halo_send = 0, n_halo_ghost = 0;   // wcsph_slab_set_halo
  int win_x0 = 0, win_x1 = -1;  // cell-x window (slab decomposition); x1 < 0 => whole box
  int capacity = 0;
  double grid_r = 0.0;
  std::vector<void *> allocs;

  int fail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    err = buf;
    return code;
  }
};

#define HIPCHK(x)                                                    \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) return ctx->fail(WCSPH_EHIP, #x, e_);      \
  } while (0)
#define LAUNCHCHK()                                                  \
  do {                                                               \
    hipError_t e_ = hipGetLastError();                               \
    if (e_ != hipSuccess) return ctx->fail(WCSPH_EHIP, "kernel launch", e_); \
  } while (0)
// lap profiling: see wcsph_ctx::lap_ev
static inline void prof_lap(wcsph_ctx *ctx, const char *tag) {
  if (ctx->lap_tag.size() >= ctx->lap_ev.size()) return;
  if (hipEventRecord(ctx->lap_ev[ctx->lap_tag.size()], ctx->stream) != hipSuccess) return;
  ctx->lap_tag.push_back(ctx->lap_scope ? ctx->lap_scope : tag);
  ctx->lap_waited.push_back(ctx->lap_wait_flag ? 1 : 0);
  ctx->lap_wait_flag = false;
}
#define PROF_LAP(ctx, tag) do { if ((ctx)->lap_on) prof_lap(ctx, tag); } while (0)
#define NEED_SETUP()                                                                  \
  if (!ctx) return WCSPH_EARG;                                                        \
  if (!ctx->is_setup) return ctx->fail(WCSPH_EARG, "wcsph_setup() has not been called")
// a resident slab state between wcsph_state_apply and the grid update that sorts it in has dead rows and last step's order:
// only the grid update may run then (every accessor that iterates d.nl or where[] would read wrong or dead rows)
#define NEED_COMPACT()                                                                \
  NEED_SETUP();                                                                       \
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "the state has pending changes (wcsph_state_apply) that no grid update has sorted in yet: only wcsph_grid_update / wcsph_slab_step may run now")
#define NEED_GRID()                                                                   \
  NEED_SETUP();                                                                       \
  if (!ctx->grid_valid) return ctx->fail(WCSPH_EARG, "wcsph_grid_update() has not been called since setup")

static inline int nblk(int n) { return (n + BLOCK - 1) / BLOCK; }
// grids of the neighbour passes are a multiple of 8 blocks (XCD remap in sorted_index())
static inline int nblk8(int n) { int b = nblk(n); return ((b + 7) / 8) * 8; }

template <class T>
static int dalloc(wcsph_ctx *ctx, T **p, size_t n) {
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
  HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(T), ctx->stream));
  ctx->allocs.push_back((void *)*p);
  return 0;
}
#define DALLOC(ptr, n) do { int rc_ = dalloc(ctx, &(ptr), (size_t)(n)); if (rc_) return rc_; } while (0)

static int check_flags(wcsph_ctx *ctx) {
  int f = ctx->h_flags[FLAG_ERR];
  // a bucket beyond the cap: the byte table may be clamped (k_bucket_sum at 255) or saturated (k_cell_assign) -- it is REBUILT by the
  // next grid update instead of being updated in place, and by every one after it for as long as the condition is reported
  if (f & ERRBIT_BUCKET) ctx->hist_valid = false;
  if (f & ERRBIT_BUCKET) return ctx->fail(WCSPH_EBUCKET, "more than maxInGrid particles in one hash bucket (ref: HashGrid.py:72-74)");
  if (f & ERRBIT_NEIGH) return ctx->fail(WCSPH_ENEIGH, "more than maxNeighbour neighbour-list entries (ref: HashGrid.py:100-103)");
  if (f & ERRBIT_LISTCAP) return ctx->fail(WCSPH_ENEIGH, "compact neighbour list capacity (wcsph_params.list_cap, default 256 slots of one kind) exceeded");
  if (f & ERRBIT_ROWITEMS) return ctx->fail(WCSPH_EBUCKET, "raw-row item table of the literal compute_tension too small (it is sized from its scanned total before it is filled: this should not happen)");
  if (f & ERRBIT_NONFINITE) return ctx->fail(WCSPH_ENONFINITE, "non-finite particle position (ref: dfsph.py:645-647)");
  return 0;
}
// Device -> host hand-over of the reduction slots and error flags at the solvers' decision points (4-5 per step).
// One tiny kernel sums the stripes and writes the totals, the flags and then a sequence number straight into pinned
// host memory; the host polls the sequence word.  A copy + blocking stream wait costs ~25 us of idle GPU per
// decision, this ~10.
// Error flags (bucket / list overflow, non-finite position) are STICKY ON THE DEVICE UNTIL THEY HAVE BEEN HANDED TO THE
// HOST ONCE: no grid update clears them, k_publish clears the word it has just published.  A step that never waits for
// a reduction (wcsph_sesph_step, a bare update_grid()) therefore reports at the next synchronisation point -- the next
// wcsph_sync(), solver decision or field transfer -- instead of losing the condition to the following grid update.
// Speculation (wcsph_dfsph_step): the kernel that publishes a solver's error sum also takes the solver's stopping decision
// ON THE DEVICE -- the same IEEE double expressions the host loops evaluate (dfsph.py:141-144, 160-163) -- and leaves it in
// a predicate word; the kernels that follow in the common case (the solver stops) are already in the stream behind it,
// gated by that word (Dev::pred), so the GPU runs on while the mail travels to the host.  The host takes the decision
// FROM THE MAIL (one source of truth) and, when the solver has to go on, finds the gated kernels to have done nothing.
enum { PRED_NONE = 0, PRED_PRESS = 1, PRED_DIV = 2, PRED_VISC = 3, PRED_PASS = 4 };
// gate: the publish itself is speculative -- when *gate == 0 it publishes nothing (the host, which knows from the mail that
// set the gate, does not wait for it) and CLOSES its own predicate, so that a chain of gated segments stops as a whole
struct PredSpec { int kind; double thr, n; int *out; const int *gate; };
#define MAIL_RING 8   /* publishes that may be in flight before the host reads the first of them */
struct HostMail { double acc[8]; int flags[4]; unsigned long long seq; int pred; int pad_; };
__device__ __forceinline__ void publish_body(const double *acc, int *flags, HostMail *m, unsigned long long seq, const PredSpec &ps) {
  const int t = threadIdx.x;
  if (ps.gate && *ps.gate == 0) { if (t == 0 && ps.out) *ps.out = 0; return; }
  // one wave: lane t holds stripe t of every slot (eight independent loads in flight), the totals are formed by the same
  // shuffle tree as acc_total() on the device -- host and device see the same bits -- and land in lane 0
  static_assert(ACC_STRIPES == WAVE, "one stripe per lane");
  double tot[8];
#pragma unroll
  for (int sl = 0; sl < 8; sl++) tot[sl] = acc[sl * ACC_STRIPES + t];
#pragma unroll
  for (int sl = 0; sl < 8; sl++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot[sl] += __shfl_down(tot[sl], o, WAVE);
  if (t == 0) {
#pragma unroll
    for (int sl = 0; sl < 8; sl++) m->acc[sl] = tot[sl];
    int stop = 0;
    const double e = (double)(float)tot[ACC_ERR];              // avg_density_err is a float32 field
    if (ps.kind == PRED_PRESS) stop = !(e / ps.n > ps.thr);   // dfsph.py:160-161: while err > 0.001, err = avg_density_err / N
    else if (ps.kind == PRED_DIV) stop = !(e > ps.thr);       // dfsph.py:141: while avg_density_err > err
    else if (ps.kind == PRED_VISC) {                          // dfsph.py:96-97 after the first compute_viscosity_force
      const double dzero = (double)(float)tot[ACC_DZERO], delta = (double)(float)tot[ACC_DNEW];   // cg_delta_zero, cg_delta: float32 fields
      stop = (delta <= ps.thr * dzero) || (dzero < ps.n);
    } else if (ps.kind == PRED_PASS) stop = 1;
    m->pred = stop;
    if (ps.out) *ps.out = stop;
  } else if (t >= 8 && t < 12) {
    // (a device-scope load: k_cfl_store_publish's own lane 0 has just written two of these words)
    m->flags[t - 8] = __hip_atomic_load(&flags[t - 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t - 8 == FLAG_ERR) flags[FLAG_ERR] = 0;   // handed over: reported exactly once
  }
  __syncthreads();
  if (t == 0) { __threadfence_system(); __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__global__ void k_publish(const double *acc, int *flags, HostMail *m, unsigned long long seq, PredSpec ps) {
  publish_body(acc, flags, m, seq, ps);
}
// the CFL reduction's one-wave tail (k_cfl_store) and the mail that carries its result, in one dispatch
__global__ void k_cfl_store_publish(Dev d, DtRule r, HostMail *m, unsigned long long seq, PredSpec ps) {
  if (!(d.pred && *d.pred == 0)) cfl_store_body(d, r);
  __syncthreads();
  publish_body(d.acc, d.flags, m, seq, ps);
}
static int publish_async(wcsph_ctx *ctx, PredSpec ps = {PRED_NONE, 0.0, 1.0, nullptr, nullptr}) {
  const unsigned long long seq = ++ctx->mail_seq;
  HostMail *m = reinterpret_cast<HostMail *>(ctx->h_acc_raw) + seq % MAIL_RING;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, (const double *)ctx->d.acc, ctx->d.flags, m, seq, ps);
  LAUNCHCHK();
  return 0;
}
static int cfl_store_publish_async(wcsph_ctx *ctx, const DtRule &r, PredSpec ps) {   // ctx->d.pred gates the CFL part, ps.gate the mail
  const unsigned long long seq = ++ctx->mail_seq;
  HostMail *m = reinterpret_cast<HostMail *>(ctx->h_acc_raw) + seq % MAIL_RING;
  hipLaunchKernelGGL(k_cfl_store_publish, dim3(1), dim3(64), 0, ctx->stream, ctx->d, r, m, seq, ps);
  LAUNCHCHK();
  return 0;
}
// waits for the mail of publish number `seq` (0: the last publish_async()); check_acc: also refuse non-finite reduction totals
static int collect_mail(wcsph_ctx *ctx, bool check_acc, unsigned long long seq = 0) {
  if (seq == 0) seq = ctx->mail_seq;
  HostMail *m = reinterpret_cast<HostMail *>(ctx->h_acc_raw) + seq % MAIL_RING;
  volatile unsigned long long *flag = &m->seq;
  if (ctx->lap_on && *flag != seq) ctx->lap_wait_flag = true;   // the host is about to spin: the next lap starts behind a host wait
  for (unsigned long spins = 0; *flag != seq; spins++) {
    if ((spins & 0xfffff) == 0xfffff) {   // every ~1 M polls: is the stream still alive?
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return ctx->fail(WCSPH_EHIP, hipGetErrorString(q));
      if (q == hipSuccess && *flag != seq) return ctx->fail(WCSPH_EHIP, "pull_scalars: stream drained without the mail arriving");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int s = 0; s < 8; s++) ctx->h_acc[s] = m->acc[s];
  for (int s = 0; s < 4; s++) ctx->h_flags[s] = m->flags[s];
  ctx->h_pred = m->pred;
  int rc = check_flags(ctx);
  if (rc || !check_acc) return rc;
  // a NaN error sum makes every `err > tol` test of the reference's host loops false, i.e. "converged" (dfsph.py:141-163)
  for (int s = 0; s < 8; s++)
    if (!std::isfinite(ctx->h_acc[s])) return ctx->fail(WCSPH_ENONFINITE, "non-finite reduction result (solver error sum / PCG dot product)");
  { float vm; memcpy(&vm, &ctx->h_flags[FLAG_VMAX], 4); if (!std::isfinite(vm)) return ctx->fail(WCSPH_ENONFINITE, "non-finite CFL velocity maximum"); }
  return 0;
}
static int pull_scalars(wcsph_ctx *ctx) {
  int rc = publish_async(ctx);
  if (rc) return rc;
  return collect_mail(ctx, true);
}
static int zero_acc(wcsph_ctx *ctx, int slot) {
  HIPCHK(hipMemsetAsync(ctx->d.acc + slot * ACC_STRIPES, 0, ACC_STRIPES * sizeof(double), ctx->stream));
  return 0;
}
