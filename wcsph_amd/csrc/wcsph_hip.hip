// wcsph_hip.hip -- C ABI (include/wcsph.h) over the gfx950 kernels in grid.h / dfsph_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see __graft_entry__.build()).
#include "../../include/wcsph.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "device.h"
#include "grid.h"
#include "dfsph_kernels.h"
#include "solver_extra.h"
#include "surface.h"
#include "sampler.h"
#include "mcgrid.h"
#include "canvas.h"
#include "bucket.h"

#ifndef WCSPH_SPECULATE
#define WCSPH_SPECULATE 1
#endif
struct wcsph_ctx {
  wcsph_params p;
  Dev d;          // DFSPH view (CubicKernel with h = HashGrid.searchR)
  hipStream_t stream = nullptr, own_stream = nullptr;
  // faithful mode, one context: the reference bucket histogram and neighborCount (k_bucket_hist .. k_ncount_box) run on a
  // side stream beside the counting sort and the list build, which need neither (wcsph_grid_update)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_assign = nullptr, ev_scan = nullptr, ev_side = nullptr;
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
  bool hist_dirty = false;        // a grid update began and did not reach the kernels that zero bcount_l / region_occ
  BucketTab bk = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the reference's hash table itself (bucket.h), built on demand
  bool bk_valid = false, bk_rows_valid = false;
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
  bool speculate = WCSPH_SPECULATE != 0;   // -DWCSPH_SPECULATE=0: the rounds 1-3 control flow (one host wait per decision), for A/B builds
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
#define NEED_SETUP()                                                                  \
  if (!ctx) return WCSPH_EARG;                                                        \
  if (!ctx->is_setup) return ctx->fail(WCSPH_EARG, "wcsph_setup() has not been called")
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
  if (f & ERRBIT_BUCKET) return ctx->fail(WCSPH_EBUCKET, "more than maxInGrid particles in one hash bucket (ref: HashGrid.py:72-74)");
  if (f & ERRBIT_NEIGH) return ctx->fail(WCSPH_ENEIGH, "more than maxNeighbour neighbour-list entries (ref: HashGrid.py:100-103)");
  if (f & ERRBIT_LISTCAP) return ctx->fail(WCSPH_ENEIGH, "compact neighbour list capacity (wcsph_params.list_cap, default 256 slots of one kind) exceeded");
  if (f & ERRBIT_ROWITEMS) return ctx->fail(WCSPH_EBUCKET, "raw-row item table of the literal compute_tension full (more than 2 bucket entries per grid cell on average)");
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
__global__ void k_publish(const double *acc, int *flags, HostMail *m, unsigned long long seq, PredSpec ps) {
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
    m->flags[t - 8] = flags[t - 8];
    if (t - 8 == FLAG_ERR) flags[FLAG_ERR] = 0;   // handed over: reported exactly once
  }
  __syncthreads();
  if (t == 0) { __threadfence_system(); __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static int publish_async(wcsph_ctx *ctx, PredSpec ps = {PRED_NONE, 0.0, 1.0, nullptr, nullptr}) {
  const unsigned long long seq = ++ctx->mail_seq;
  HostMail *m = reinterpret_cast<HostMail *>(ctx->h_acc_raw) + seq % MAIL_RING;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, (const double *)ctx->d.acc, ctx->d.flags, m, seq, ps);
  LAUNCHCHK();
  return 0;
}
// waits for the mail of publish number `seq` (0: the last publish_async()); check_acc: also refuse non-finite reduction totals
static int collect_mail(wcsph_ctx *ctx, bool check_acc, unsigned long long seq = 0) {
  if (seq == 0) seq = ctx->mail_seq;
  HostMail *m = reinterpret_cast<HostMail *>(ctx->h_acc_raw) + seq % MAIL_RING;
  volatile unsigned long long *flag = &m->seq;
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
  HIPCHK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_assign, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_scan, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
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
  DALLOC(d.minv0, nl); DALLOC(d.minv1, nl); DALLOC(d.minv2, nl);
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
  DALLOC(d.acc, 8 * ACC_STRIPES); DALLOC(d.flags, 4 + ACC_STRIPES);   // [4..]: striped vmax
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
  if (ctx->side_stream) hipStreamSynchronize(ctx->side_stream);
  for (void *p : ctx->allocs) hipFree(p);
  if (ctx->h_acc_raw) hipHostFree(ctx->h_acc_raw);
  for (hipEvent_t e : ctx->prof_ev) hipEventDestroy(e);
  if (ctx->h_flags) hipHostFree(ctx->h_flags);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
  for (hipEvent_t e : {ctx->ev_assign, ctx->ev_scan, ctx->ev_side}) if (e) hipEventDestroy(e);
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
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->bsum, nb);
  hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BLOCK), 0, ctx->stream, ctx->d, in, n, (const int *)ctx->bsum, out, bbox, clear);
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
  DALLOC(ctx->cnt, nscan); DALLOC(d.cs_l, nscan); DALLOC(d.cs_s, nscan); DALLOC(ctx->bsum, ctx->scan_blocks + 1);
  DALLOC(d.cell_info, d.ncell_p);
  DALLOC(d.bbox, 8); DALLOC(ctx->bbox_part, 6 * (size_t)(nblk(ctx->capacity) + 1));
  if (d.faithful) {
    DALLOC(ctx->bcount, d.table_size + 1); DALLOC(ctx->bcount_s, d.table_size + 1); DALLOC(ctx->bcount_l, d.table_size + 1);
    ctx->hb_nbins = (d.table_size + (1 << HB_SHIFT) - 1) >> HB_SHIFT;
    if (ctx->hb_nbins <= HB_MAXBINS) {   // (larger tables keep the atomics)
      // a range holds table_size / nbins buckets of a pseudo-random hash: twice its share of the rows + a margin
      ctx->hb_cap = (int)(2 * ((size_t)ctx->capacity / ctx->hb_nbins + 1) + 8192);
#ifdef HB_TEST_CAP   /* test build: lists that overflow at once, i.e. the atomics fall-back on every update */
      ctx->hb_cap = HB_TEST_CAP;
#endif
      DALLOC(ctx->hb_bins, HB_MAXBINS + 1); DALLOC(ctx->hb_buf, (size_t)ctx->hb_nbins * ctx->hb_cap);
    }   // [T] stays 0: the scan's end sentinel
    DALLOC(ctx->cellval, d.ncell); DALLOC(ctx->bcount8, d.table_size + 1);
    {
      const int n0 = d.perm ? d.ny : d.nx, n1 = d.perm ? d.nx : d.ny;
      ctx->n_regions = ((n0 + BC0 - 1) / BC0) * ((n1 + BC1 - 1) / BC1) * ((d.nz + BCZ - 1) / BCZ);
      DALLOC(ctx->region_occ, (size_t)ctx->n_regions);
    }
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

// ---- HashGrid.update_grid -------------------------------------------------------------------
// first half: cells, counting sort, this context's share of the reference bucket histogram (owned liquid particles)
// this step's liquid bucket histogram of the owned rows (ref: HashGrid.py:67-76), from the packed cells k_cell_assign left:
// binned per workgroup and counted in LDS (grid.h k_hash_bin / k_bin_hist), atomics when the table is too large for that
#ifndef WCSPH_BINNED_HIST
#define WCSPH_BINNED_HIST 1   /* -DWCSPH_BINNED_HIST=0: one device atomic per particle (the A/B build) */
#endif
static int bucket_hist_launches(wcsph_ctx *ctx, hipStream_t s2) {
  Dev &d = ctx->d;
  const int nh = d.nl < d.n_owned ? d.nl : d.n_owned;
  if (nh > 0 && WCSPH_BINNED_HIST && ctx->hb_bins) {
    HIPCHK(hipMemsetAsync(ctx->hb_bins + HB_FLAG, 0, sizeof(int), s2));
    hipLaunchKernelGGL(k_hash_bin, dim3((nh + BLOCK * HB_ROWS - 1) / (BLOCK * HB_ROWS)), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh,
                       ctx->hb_nbins, ctx->hb_cap, ctx->hb_bins, ctx->hb_buf);
    hipLaunchKernelGGL(k_bin_hist, dim3(ctx->hb_nbins), dim3(1024), 0, s2, d, ctx->hb_cap, ctx->hb_bins, (const int *)ctx->hb_buf, ctx->bcount_l);
    hipLaunchKernelGGL(k_bucket_hist_gated, dim3(nblk(nh)), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh, ctx->bcount_l, ctx->hb_nbins, ctx->hb_bins);
  } else if (nh > 0) hipLaunchKernelGGL(k_bucket_hist, dim3(nblk(nh)), dim3(BLOCK), 0, s2, d, (const int *)ctx->pw, nh, ctx->bcount_l);
  return 0;
}
// fork (wcsph_grid_update of ONE faithful context): the bucket histogram is not taken inside k_cell_assign -- its 2 M random
// device-scope atomics are 71 of that kernel's 88 us at C3 -- but by k_bucket_hist on the side stream, followed there by the
// bucket sums, gridCount[hash(cell)] and (once the scan is done) the neighborCount box sums, while this stream scans,
// sorts and builds the lists: none of those reads the histogram or neighborCount.  grid_end() joins.  What this buys is
// the holes of the one-block kernels (k_bbox_final, k_scan_bsums: ~21 us of an otherwise idle GPU per step), 1-1.5 % of the
// step at C3 -- full-chip kernels of two streams time-slice, they do not add up (forking behind the sort, beside the list
// build, gains nothing: profiles/r03_experiments.md (21)).  On a slab decomposition the histogram is all-reduced between
// the two halves and stays where it was.
static void neighbor_count_launches(wcsph_ctx *ctx, hipStream_t s, bool box);
static int grid_begin(wcsph_ctx *ctx, bool fork = false) {
  Dev &d = ctx->d;
  const int nscan = d.ncell_p + 2;
  hipStream_t s = ctx->stream;
  // ctx->cnt is all zero here: k_scan_final clears the counters it consumes.  The error flags are NOT cleared here: a
  // condition stays on the device until k_publish has handed it to the host once (see pull_scalars)
  ctx->pr_valid = false;   // rows are re-sorted
  // the liquid bucket histogram and the region flags are zero here: their consumers (k_bucket_sum, k_ncount_box) clear
  // them -- unless the previous update did not run to its end
  if (d.faithful && ctx->hist_dirty) {
    HIPCHK(hipMemsetAsync(ctx->bcount_l, 0, (size_t)d.table_size * sizeof(int), s));
    HIPCHK(hipMemsetAsync(ctx->region_occ, 0, (size_t)ctx->n_regions * sizeof(int), s));
    if (ctx->hb_bins) HIPCHK(hipMemsetAsync(ctx->hb_bins, 0, (HB_MAXBINS + 1) * sizeof(int), s));
  }
  ctx->hist_dirty = true;
  if (d.nl > 0)
    hipLaunchKernelGGL(k_cell_assign, dim3(nblk(d.nl)), dim3(BLOCK), 0, s, d, (const float4 *)d.pos, d.nl, ctx->cnt, ctx->pw, ctx->slot,
                       (int *)nullptr, ctx->bbox_part, d.n_owned,
                       d.faithful ? ctx->region_occ : (int *)nullptr);
  if (d.faithful && !fork) { int rc_ = bucket_hist_launches(ctx, s); if (rc_) return rc_; }   // slab halves: on this stream, before the all-reduce
  if (fork) {
    hipStream_t s2 = ctx->side_stream;
    HIPCHK(hipEventRecord(ctx->ev_assign, s));
    HIPCHK(hipStreamWaitEvent(s2, ctx->ev_assign, 0));
    { int rc_ = bucket_hist_launches(ctx, s2); if (rc_) return rc_; }
    neighbor_count_launches(ctx, s2, false);
  }
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(BLOCK), 0, s, (const int *)ctx->bbox_part, d.nl > 0 ? nblk(d.nl) : 0, d.bbox);
  int rc = run_scan(ctx, ctx->cnt, nscan, d.cs_l, d.bbox, 1);
  if (rc) return rc;
  if (fork) {   // the box sums read the cell table and the occupied box
    HIPCHK(hipEventRecord(ctx->ev_scan, s));
    HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_scan, 0));
    neighbor_count_launches(ctx, ctx->side_stream, true);
    HIPCHK(hipEventRecord(ctx->ev_side, ctx->side_stream));
  }

  if (d.nl > 0) {
    SortBuffers b = {d.pos, d.vel, d.vel_guess, d.omega, d.kappa, d.kappa_v, d.pressure, d.oid,
                     ctx->pos_b, ctx->vel_b, ctx->vel_guess_b, ctx->omega_b, ctx->kappa_b, ctx->kappa_v_b, ctx->pressure_b, ctx->oid_b, d.where};
    hipLaunchKernelGGL(k_scatter_liquid, dim3(nblk(d.nl)), dim3(BLOCK), 0, s, d, b, (const int *)d.cs_l, (const int *)ctx->pw,
                       (const int *)ctx->slot);
    std::swap(d.pos, ctx->pos_b); std::swap(d.vel, ctx->vel_b); std::swap(d.vel_guess, ctx->vel_guess_b);
    std::swap(d.omega, ctx->omega_b); std::swap(d.kappa, ctx->kappa_b); std::swap(d.kappa_v, ctx->kappa_v_b);
    std::swap(d.oid, ctx->oid_b); std::swap(d.pressure, ctx->pressure_b);
  }
  LAUNCHCHK();
  return 0;
}
// second half: raw neighbour-list length (HashGrid.neighborCount) from the GLOBAL bucket occupancy, compact lists
// faithful mode: box = false: bucket sums and gridCount[hash(cell)] of the occupied regions; box = true: neighborCount, the
// 5x5x5 box sum of gridCount[hash(cell)] around every occupied cell, one LDS-tiled launch (grid.h)
static void neighbor_count_launches(wcsph_ctx *ctx, hipStream_t s, bool box) {
  Dev &d = ctx->d;
  const int nb = ctx->n_regions;
  if (!box) {
    hipLaunchKernelGGL(k_bucket_sum, dim3(nblk(d.table_size)), dim3(BLOCK), 0, s, d, (const int *)ctx->bcount_s, ctx->bcount_l, ctx->bcount, ctx->bcount8, d.flags);
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
static int grid_end(wcsph_ctx *ctx, bool fork = false) {
  Dev &d = ctx->d;
  hipStream_t s = ctx->stream;
  if (d.faithful) {
    if (!fork) { neighbor_count_launches(ctx, s, false); neighbor_count_launches(ctx, s, true); }
  } else if (d.nl > 0) {
    hipLaunchKernelGGL(k_ncount_clean, dim3(nblk(d.nl)), dim3(BLOCK), 0, s, d);
  }
  if (d.nl > 0) hipLaunchKernelGGL(k_build_lists, dim3(nblk8(d.nl)), dim3(BLOCK), 0, s, d, 1);
  if (fork) HIPCHK(hipStreamWaitEvent(s, ctx->ev_side, 0));   // join: whatever follows may read neighborCount and the flags
  ctx->d.rev ^= 1;
  LAUNCHCHK();
  ctx->hist_dirty = false;   // k_bucket_sum / k_ncount_box have left the histogram and the region flags zeroed
  ctx->grid_valid = true;
  ctx->bk_valid = false;
  return WCSPH_OK;
}
int wcsph_grid_update(wcsph_ctx *ctx) {
  NEED_SETUP();
#ifndef WCSPH_GRID_FORK
#define WCSPH_GRID_FORK 1   /* -DWCSPH_GRID_FORK=0: everything on the context's stream (the A/B build) */
#endif
  const bool fork = WCSPH_GRID_FORK && ctx->d.faithful && ctx->d.nl > 0;
  int rc = grid_begin(ctx, fork);
  if (rc) return rc;
  return grid_end(ctx, fork);
}
// slab decomposition, faithful neighbour mode: the drivers all-reduce the liquid bucket histogram between the halves
int wcsph_grid_update_begin(wcsph_ctx *ctx) { NEED_SETUP(); ctx->grid_valid = false; return grid_begin(ctx); }
int wcsph_grid_update_end(wcsph_ctx *ctx) { NEED_SETUP(); return grid_end(ctx); }
int wcsph_bucket_hist_pack(wcsph_ctx *ctx, void **d_ptr, int64_t *n_words) {   // liquid histogram, two 16-bit counts per word
  NEED_SETUP();
  if (!d_ptr || !n_words) return WCSPH_EARG;
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "the bucket histogram exists in faithful neighbour mode only");
  const int T = ctx->d.table_size, W = (T + 1) / 2;
  // per-rank clamp max_in_grid + 1: enough to trip the cap after the sum, and world * (max_in_grid + 1) stays below 2^16
  // for every sane cap (a carry into the neighbouring bucket's half needs max_in_grid > 8190 at 8 ranks)
  const int clamp = ctx->d.max_in_grid + 1 < 0xffff ? ctx->d.max_in_grid + 1 : 0xffff;
  hipLaunchKernelGGL(k_hist_pack16, dim3(nblk(W)), dim3(BLOCK), 0, ctx->stream, (const int *)ctx->bcount_l, T, ctx->bcount, clamp);   // bcount is rebuilt by grid_end
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
  int rc = run_scan(ctx, ctx->bcount, T + 1, ctx->bk.start, nullptr, 0);   // bcount = gridCount of this step (grid update)
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
  if (!ctx->bk.coff) {
    // every cell within 2 of the liquid's box holds a copy of its bucket: ~one item per cell while the table has about as
    // many buckets as there are particles; twice the cell count is the cap (ERRBIT_ROWITEMS beyond)
    ctx->bk.citem_cap = (int)std::min<size_t>((size_t)2 * d.ncell + 1024, (size_t)0x7ffffff0);
    DALLOC(ctx->bk.coff, (size_t)d.ncell + SCAN_TILE + 2); DALLOC(ctx->bk.citem, (size_t)ctx->bk.citem_cap);
  }
  hipLaunchKernelGGL(k_bk_cellcount, dim3(nblk(d.ncell + 1)), dim3(BLOCK), 0, ctx->stream, d, ctx->bk);
  rc = run_scan(ctx, ctx->bk.coff, d.ncell + 1, ctx->bk.coff, nullptr, 0);   // in place: a thread reads its eight counts before it writes them
  if (rc) return rc;
  hipLaunchKernelGGL(k_bk_cellfill, dim3(nblk(d.ncell)), dim3(BLOCK), 0, ctx->stream, d, ctx->bk);
  LAUNCHCHK();
  ctx->bk_rows_valid = true;
  return 0;
}
int wcsph_hashgrid_count_to_host(wcsph_ctx *ctx, int32_t *dst, int64_t n) {   // HashGrid.gridCount
  NEED_GRID();
  if (!dst || n != ctx->d.table_size) return ctx->fail(WCSPH_EARG, "wcsph_hashgrid_count_to_host: n must be the hash table size");
  if (!ctx->d.faithful) return ctx->fail(WCSPH_EARG, "HashGrid.gridCount is the reference table: faithful neighbour mode only");
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
  NEED_SETUP();
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
    else {  // cg_Minv / G rows -> [n][3][3]
      std::vector<float4> r0(nl), r1(nl), r2(nl);
      std::vector<int> oid(nl);
      const bool isG = fd.kind == 10;
      HIPCHK(hipMemcpyAsync(r0.data(), isG ? ctx->surf.g0 : d.minv0, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(r1.data(), isG ? ctx->surf.g1 : d.minv1, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(r2.data(), isG ? ctx->surf.g2 : d.minv2, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
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
  NEED_SETUP();
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
  if (field == WCSPH_F_POS) ctx->grid_valid = false;  // cells are stale until the next grid update
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
  NEED_SETUP();
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
  NEED_SETUP();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "field_rows: writable vec3 / scalar fields only");
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || !src) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  const int per = fd.kind == 3 ? 3 : 1, stride = fd.kind == 3 ? 4 : 1;
  HIPCHK(hipMemcpyAsync(ctx->stage, src, (size_t)n * per * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rows_in, dim3(nblk(n * per)), dim3(BLOCK), 0, ctx->stream, (float *)fd.ptr, stride, per, (const int *)ctx->d.where, i0, n, (const float *)ctx->stage);
  LAUNCHCHK();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (field == WCSPH_F_POS) ctx->grid_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}

int wcsph_field_gather_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, void *d_dst) {
  NEED_SETUP();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "gather_rows: vec3 / scalar fields only");
  if (n <= 0) return WCSPH_OK;
  if (fd.kind == 3) hipLaunchKernelGGL(k_gather_rows3, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (const float4 *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (float *)d_dst);
  else hipLaunchKernelGGL(k_gather_rows1, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (const float *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (float *)d_dst);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_field_scatter_rows(wcsph_ctx *ctx, int field, const int32_t *d_idx, int32_t n, const void *d_src) {
  NEED_SETUP();
  FieldDesc fd = field_desc(ctx, field);
  if (fd.kind != 3 && fd.kind != 1) return ctx->fail(WCSPH_EARG, "scatter_rows: vec3 / scalar fields only");
  if (n <= 0) return WCSPH_OK;
  if (fd.kind == 3) hipLaunchKernelGGL(k_scatter_rows3, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (float4 *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (const float *)d_src);
  else hipLaunchKernelGGL(k_scatter_rows1, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, (float *)fd.ptr, (const int *)ctx->d.where, d_idx, n, (const float *)d_src);
  LAUNCHCHK();
  if (field == WCSPH_F_POS) ctx->grid_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}
int wcsph_set_owned(wcsph_ctx *ctx, int32_t n_owned) {
  if (!ctx || n_owned < 0 || n_owned > ctx->d.nl) return WCSPH_EARG;
  ctx->d.n_owned = n_owned;
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

// ---- dfsph.py kernels -------------------------------------------------------------------------
#define PASS(kern, ...)                                                                           \
  do {                                                                                            \
    if (ctx->d.nl > 0) hipLaunchKernelGGL(kern, dim3(nblk8(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, ##__VA_ARGS__); \
    ctx->d.rev ^= 1;   /* the next neighbour pass walks the particles the other way round (device.h sorted_index) */ \
    LAUNCHCHK();                                                                                  \
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
  PASS(k_vel_from_k<0>, (const float4 *)ctx->d.pk, ctx->dt, ctx->d.kappa_v);   // also clears ACC_ERR
  if (ctx->fuse_elem && ctx->vort_fuse && ctx->pr_valid) {
    PASS(k_drho_div_iter_vort, ctx->dt);
    ctx->vort_in_div = true;
  } else PASS((k_drho_div<DRHO_ITER, true>), ctx->dt);
  return 0;
}
static int pressure_iter_fused(wcsph_ctx *ctx) {
  if (ctx->d.n_owned < ctx->d.nl) return pressure_iter_async(ctx);
  const bool prof = ctx->prof_on && ctx->prof_used + 2 <= ctx->prof_ev.size();
  if (prof) HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
  PASS(k_vel_from_k<0>, (const float4 *)ctx->d.pk, ctx->dt, ctx->d.kappa);     // also clears ACC_ERR
  PASS((k_drho_press<DRHO_ITER, true>), ctx->dt);
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

int wcsph_dfsph_warmstart_pressure(wcsph_ctx *ctx) {
  NEED_GRID();
  if (ctx->warm_scaled) ctx->warm_scaled = false;   // wcsph_dfsph_step: done by k_update_vel_warm_scale
  else ELEM(k_warm_press_scale, ctx->dt);
  PASS(k_warm_press_vel, ctx->dt);
  return WCSPH_OK;
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
  if (ctx->spec_dt) {   // the kernels queued behind the CFL reduction ran with the device's deltaT: it has to be this one, bit for bit
    ctx->spec_dt = false;
    float ddt; memcpy(&ddt, &ctx->h_flags[FLAG_DT], 4);
    if (memcmp(&ddt, &ctx->dt, 4) != 0) { ctx->spec_press_seq = 0; return ctx->fail(WCSPH_EHIP, "deltaT formed on the device differs from the host rule (dfsph.py:113-129)"); }
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
        hipLaunchKernelGGL(k_cfl_store, dim3(1), dim3(WAVE), 0, ctx->stream, ctx->d,
                           DtRule{ctx->dt_dev, ctx->dt, it, (double)ctx->p.eps, ctx->p.cfl_radius, ctx->p.user_max_t, ctx->p.user_min_t});
        ctx->d.pred = nullptr;
        if ((rc = publish_async(ctx, {PRED_PASS, 0.0, 1.0, nullptr, pB}))) return rc;
        seqC = ctx->mail_seq;
        // ... and the pressure solve up to its first decision (update_vel, warm start, begin, two iterations: dfsph.py:150-161
        // forces two), still gated by B, taking deltaT from the device; its end gated by its own decision (mail D)
        int *pD = ctx->pred_dev + 0;
        ctx->spec_prof_mark = ctx->prof_used;   // in-situ timing: the event pairs of iterations that turn out gated off are dropped
        ctx->d.pred = pB; ctx->d.dtp = ctx->dt_dev;
        ELEM(k_update_vel_warm_scale, ctx->dt);
        PASS(k_warm_press_vel, ctx->dt);
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
  } else if (fuse) { ELEM(k_update_vel_warm_scale, ctx->dt); ctx->warm_scaled = true; }
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

// ---- ParticleData surface reconstruction (post-processing) ------------------------------------------
static int surf_alloc(wcsph_ctx *ctx) {
  if (ctx->surf.color) return 0;
  const size_t n = (size_t)ctx->capacity;
  DALLOC(ctx->surf.color, n); DALLOC(ctx->surf.color_grad, n); DALLOC(ctx->surf.pos_avr, n);
  DALLOC(ctx->surf.g0, n); DALLOC(ctx->surf.g1, n); DALLOC(ctx->surf.g2, n);
  return 0;
}
int wcsph_surface_color_map(wcsph_ctx *ctx) {
  NEED_GRID();
  int rc = surf_alloc(ctx);
  if (rc) return rc;
  Dev v = ctx->d;
  v.list_r2 = v.h2;   // Cubic kernel support
  if (v.nl > 0) {
    hipLaunchKernelGGL(k_surf_color, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
    hipLaunchKernelGGL(k_surf_color_grad, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
  }
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_surface_anisotropic_kernel(wcsph_ctx *ctx, float mc_searchR) {
  NEED_GRID();
  if (!(mc_searchR > 0.0f)) return ctx->fail(WCSPH_EARG, "mc_searchR must be positive");
  int rc = surf_alloc(ctx);
  if (rc) return rc;
  Dev v = ctx->d;
  ctx->surf.R2 = mc_searchR * 2.0f;                 // ParticleData.py:295 support of weight_func
  v.list_r2 = ctx->surf.R2 * ctx->surf.R2;          // entries of the walked cells beyond it weigh exactly zero
  if (v.nl > 0 && v.faithful && ctx->win_x1 < 0) {
    // weight_func's support (0.18) exceeds the walk radius: over the reference's raw rows (csrc/bucket.h)
    if ((rc = bucket_build(ctx))) return rc;
    hipLaunchKernelGGL(k_surf_pos_avr_raw, dim3(nblk(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->bk, ctx->surf);
    hipLaunchKernelGGL(k_surf_aniso_raw, dim3(nblk(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->bk, ctx->surf);
  } else if (v.nl > 0) {
    hipLaunchKernelGGL(k_surf_pos_avr, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
    hipLaunchKernelGGL(k_surf_aniso, dim3(nblk8(v.nl)), dim3(BLOCK), 0, ctx->stream, v, ctx->surf);
  }
  LAUNCHCHK();
  return WCSPH_OK;
}

// ---- Canvas.py (headless) ---------------------------------------------------------------------------------------
int wcsph_canvas_draw(wcsph_ctx *ctx, const float *proj_view, int32_t size_x, int32_t size_y, const float *color_sphere,
                      const float *color_point, float *img) {
  NEED_SETUP();
  if (!proj_view || size_x < 1 || size_y < 1 || size_x > 16384 || size_y > 16384 || !color_sphere || !color_point || !img) return WCSPH_EARG;
  CanvasDev c;
  memcpy(c.vp, proj_view, sizeof c.vp);
  c.sx = size_x; c.sy = size_y;
  const size_t npix = (size_t)size_x * size_y;
  float *dimg = nullptr;
  c.pix = nullptr;
  if (hipMalloc((void **)&c.pix, npix * 8) != hipSuccess || hipMalloc((void **)&dimg, npix * 12) != hipSuccess) {
    if (c.pix) hipFree(c.pix);
    return ctx->fail(WCSPH_EHIP, "wcsph_canvas_draw: allocation");
  }
  const unsigned nb = (unsigned)((npix + BLOCK - 1) / BLOCK);
  auto pack = [](const float *v) { auto q = [](float x) { x = x < 0 ? 0 : (x > 1 ? 1 : x); return (unsigned)(x * 1023.0f + 0.5f); }; return (q(v[0]) << 20) | (q(v[1]) << 10) | q(v[2]); };
  hipLaunchKernelGGL(k_canvas_clear, dim3(nb), dim3(BLOCK), 0, ctx->stream, c);
  if (ctx->d.nl > 0) hipLaunchKernelGGL(k_canvas_liquid, dim3(nblk(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, c, pack(color_sphere), pack(color_point));
  if (ctx->d.ns > 0) hipLaunchKernelGGL(k_canvas_solid, dim3(nblk(ctx->d.ns)), dim3(BLOCK), 0, ctx->stream, ctx->d, c, pack(color_point));
  hipLaunchKernelGGL(k_canvas_resolve, dim3(nb), dim3(BLOCK), 0, ctx->stream, c, dimg);
  hipError_t e1 = hipMemcpyAsync(img, dimg, npix * 12, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(c.pix); hipFree(dimg);
  if (e1 != hipSuccess || e2 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_canvas_draw");
  return WCSPH_OK;
}

// ---- MarchingCubeGrid.py ----------------------------------------------------------------------------------
int wcsph_mc_tables(int32_t *edge_table, int32_t *tri_table) {
  if (!edge_table || !tri_table) return WCSPH_EARG;
  mc_build_tables(edge_table, tri_table);
  return WCSPH_OK;
}
int wcsph_mc_setup(wcsph_ctx *ctx, double grid_r, double search_r, const float *maxb, const float *minb, int32_t dims[3]) {
  NEED_SETUP();
  if (!(grid_r > 0.0) || !(search_r > 0.0) || !maxb || !minb || !dims) return WCSPH_EARG;
  McDev &m = ctx->mc;
  long n = 1;
  for (int a = 0; a < 3; a++) {   // ref: MarchingCubeGrid.py:67-69 (python doubles, int() truncation)
    dims[a] = (int32_t)(((double)maxb[a] - (double)minb[a]) / grid_r + 1.0);
    if (dims[a] < 2) return ctx->fail(WCSPH_EARG, "wcsph_mc_setup: empty box");
    n *= dims[a];
  }
  if (n > (1L << 31) - 1) return ctx->fail(WCSPH_EARG, "wcsph_mc_setup: more than 2^31 grid nodes");
  m.nx = dims[0]; m.ny = dims[1]; m.nz = dims[2];
  m.minx = minb[0]; m.miny = minb[1]; m.minz = minb[2];
  m.grid_r = (float)grid_r;
  m.h = (float)search_r; m.inv_h = 1.0f / m.h;
  m.mk_h3 = (float)((8.0 / M_PI) / (search_r * search_r * search_r));   // kernels/CubicKernel.py:13-15
  m.rho_min = ctx->p.liqiudMass * m.mk_h3;                               // MarchingCubeGrid.py:178
  if ((size_t)n > ctx->mc_nodes) { DALLOC(m.value, (size_t)n); ctx->mc_nodes = (size_t)n; }
  if (!ctx->mc_tables) {
    std::vector<int> t(256 + 256 * MC_TRI_ROW);
    if (mc_build_tables(t.data(), t.data() + 256) > 5) return ctx->fail(WCSPH_EARG, "marching-cubes table: a case with more than 5 triangles");
    DALLOC(ctx->mc_tables, t.size());
    DALLOC(m.vertex_count, 1);
    HIPCHK(hipMemcpyAsync(ctx->mc_tables, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // t lives on this stack
  }
  m.edge_table = ctx->mc_tables; m.tri_table = ctx->mc_tables + 256;
  return WCSPH_OK;
}
#define NEED_MC() do { NEED_GRID(); if (!ctx->mc.value) return ctx->fail(WCSPH_EARG, "call wcsph_mc_setup() first"); } while (0)
int wcsph_mc_surface_value(wcsph_ctx *ctx) {   // ref: MarchingCubeGrid.py:159-184 (rho must be current: compute_density)
  NEED_MC();
  const long n = (long)ctx->mc.nx * ctx->mc.ny * ctx->mc.nz;
  hipLaunchKernelGGL(k_mc_value, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->mc);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_mc_value_to_host(wcsph_ctx *ctx, float *out, int64_t n) {
  NEED_MC();
  if (!out || n != (int64_t)ctx->mc.nx * ctx->mc.ny * ctx->mc.nz) return ctx->fail(WCSPH_EARG, "wcsph_mc_value_to_host: size mismatch");
  HIPCHK(hipMemcpyAsync(out, ctx->mc.value, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return WCSPH_OK;
}
int wcsph_mc_marching_cube(wcsph_ctx *ctx, float isolevel, float *tri_xyz, int32_t cap_vertices, int32_t *n_vertices) {
  NEED_MC();
  if (cap_vertices < 0 || !n_vertices || (cap_vertices > 0 && !tri_xyz)) return WCSPH_EARG;
  McDev m = ctx->mc;
  m.isolevel = isolevel;
  m.cap = cap_vertices;
  float *dtri = nullptr;
  HIPCHK(hipMalloc((void **)&dtri, (size_t)(cap_vertices > 0 ? cap_vertices : 1) * 3 * sizeof(float)));
  m.tri = dtri;
  hipError_t e0 = hipMemsetAsync(m.vertex_count, 0, sizeof(int), ctx->stream);
  const long n = (long)m.nx * m.ny * m.nz;
  hipLaunchKernelGGL(k_mc_march, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, m);
  int count = 0;
  hipError_t e1 = hipMemcpyAsync(&count, m.vertex_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipError_t e3 = hipSuccess;
  const int stored = count < cap_vertices ? count : (cap_vertices / 3) * 3;
  if (e0 == hipSuccess && e1 == hipSuccess && e2 == hipSuccess && stored > 0)
    e3 = hipMemcpy(tri_xyz, dtri, (size_t)stored * 3 * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(dtri);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_mc_marching_cube");
  *n_vertices = count;   // > cap_vertices: the mesh was truncated (the reference prints "exceed max tri")
  return WCSPH_OK;
}

// ---- explicit neighbour list (debug / API parity) ----------------------------------------------
__global__ void k_neighbor_lists(Dev d, int i0, int n, int cap, int *counts, int *lists) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int o = i0 + t;
  int i = d.where[o];
  int *row = lists + (size_t)t * cap;
  for (int k = 0; k < cap; k++) row[k] = -1;
  counts[t] = d.ncount[i];
  float4 pi = d.pos[i];
  const int cw = __float_as_int(pi.w);
  if (cw < 0) return;
  const int cx = cw & 1023, cy = (cw >> 10) & 1023, cz = cw >> 20;
  const int info = d.cell_info[cell_lin(d, cx, cy, cz)];
  const uint16_t *e = (info >> 1) ? d.alias + (size_t)((info >> 1) - 1) * ALIAS_STRIDE : nullptr;
  int w = 0;
  for (int k = 0; k < 125; k++) {
    int x = cx + k / 25 - 2, y = cy + (k / 5) % 5 - 2, z = cz + k % 5 - 2;
    if ((unsigned)x >= (unsigned)d.nx || (unsigned)y >= (unsigned)d.ny || (unsigned)z >= (unsigned)d.nz) continue;
    int mult = 1;
    if (e) for (int q = 0; q < e[0]; q++) if ((e[2 + q] & 0xff) == k) mult += e[2 + q] >> 8;
    int c = cell_lin(d, x, y, z);
    for (int rep = 0; rep < mult; rep++) {
      for (int j = d.cs_l[c]; j < d.cs_l[c + 1]; j++) if (j != i) { if (w < cap) row[w] = d.oid[j]; w++; }
      for (int j = d.cs_s[c]; j < d.cs_s[c + 1]; j++) { if (w < cap) row[w] = d.nl + __float_as_int(d.spos[j].w); w++; }
    }
  }
}
int wcsph_neighbors_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists) {
  NEED_GRID();
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || cap < 1 || !counts || !lists) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  const bool exact = ctx->d.faithful && ctx->win_x1 < 0;   // the reference row itself, far aliased entries included (bucket.h)
  if (exact) { int rc = bucket_build(ctx); if (rc) return rc; }
  int *dc = nullptr, *dl = nullptr;
  HIPCHK(hipMalloc((void **)&dc, (size_t)n * 4));
  HIPCHK(hipMalloc((void **)&dl, (size_t)n * cap * 4));
  if (exact) hipLaunchKernelGGL(k_bk_rows, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, ctx->bk, i0, n, cap, dc, dl);
  else hipLaunchKernelGGL(k_neighbor_lists, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, i0, n, cap, dc, dl);
  hipError_t e1 = hipMemcpyAsync(counts, dc, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(lists, dl, (size_t)n * cap * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipStreamSynchronize(ctx->stream);
  hipFree(dc); hipFree(dl);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "neighbors_to_host copy");
  return WCSPH_OK;
}

// the compact per-step lists the passes iterate, decoded from the 16-bit delta streams (diagnostics / tests)
__global__ void k_compact_lists(Dev d, int i0, int n, int cap, int *counts, int *lists) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = d.where[i0 + t];
  int *row = lists + (size_t)t * cap;
  const int head = d.nhead[i];
  const int cl = head & 0xffff, cs = (int)((unsigned)head >> 16);
  const size_t st = (size_t)d.nbr_stride;
  const unsigned short *L = reinterpret_cast<const unsigned short *>(d.nbl + i), *S = reinterpret_cast<const unsigned short *>(d.nbs + i);
  int w = 0, hops = 0, run = i;
  for (int k = 0; k < cl; k++) {
    const int dd = (int)(short)L[(size_t)(k >> 3) * st * 8 + (k & 7)];
    run += dd;
    if (!NB_IS_ENTRY(dd)) { hops++; continue; }
    if (w < cap) row[w] = d.oid[run];
    w++;
  }
  counts[3 * t] = w;
  run = cs > 0 ? d.nsbase[i] : 0;
  for (int k = 0; k < cs; k++) {
    const int dd = (int)(short)S[(size_t)(k >> 3) * st * 8 + (k & 7)];
    run += dd;
    if (!NB_IS_ENTRY(dd)) { hops++; continue; }
    if (w < cap) row[w] = d.nl + __float_as_int(d.spos[run].w);
    w++;
  }
  counts[3 * t + 1] = w - counts[3 * t];
  counts[3 * t + 2] = hops;
  for (int k = w; k < cap; k++) row[k] = -1;
}
int wcsph_compact_lists_to_host(wcsph_ctx *ctx, int32_t i0, int32_t n, int32_t cap, int32_t *counts, int32_t *lists) {
  NEED_GRID();
  if (i0 < 0 || n < 0 || i0 + n > ctx->d.nl || cap < 1 || !counts || !lists) return WCSPH_EARG;
  if (n == 0) return WCSPH_OK;
  int *dc = nullptr, *dl = nullptr;
  HIPCHK(hipMalloc((void **)&dc, (size_t)n * 12));
  HIPCHK(hipMalloc((void **)&dl, (size_t)n * cap * 4));
  hipLaunchKernelGGL(k_compact_lists, dim3(nblk(n)), dim3(BLOCK), 0, ctx->stream, ctx->d, i0, n, cap, dc, dl);
  hipError_t e1 = hipMemcpyAsync(counts, dc, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(lists, dl, (size_t)n * cap * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipStreamSynchronize(ctx->stream);
  hipFree(dc); hipFree(dl);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(WCSPH_EHIP, "compact_lists_to_host copy");
  return WCSPH_OK;
}

__global__ void __launch_bounds__(BLOCK) k_list_stats(Dev d, unsigned long long *hist) {
  const int i = sorted_index(d);
  const bool own = i < d.nl && d.oid[i] < d.n_owned;
  const int head = own ? d.nhead[i] : 0;
  const int nbl = min(((head & 0xffff) + 7) >> 3, 63), nbs = min((int)(((unsigned)head >> 16) + 7) >> 3, 63);
  if (own) { atomicAdd(&hist[nbl], 1ull); atomicAdd(&hist[64 + nbs], 1ull); }
  int ml = nbl, ms = nbs;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ml = max(ml, __shfl_xor(ml, o, WAVE)); ms = max(ms, __shfl_xor(ms, o, WAVE)); }
  int T = 0;   // the block count at least 3/4 of the wave's lanes reach
  for (int t = 1; t <= 63; t++) { if (__popcll(__ballot(nbl >= t)) < (WAVE * 3) / 4) break; T = t; }
  if ((threadIdx.x & (WAVE - 1)) == 0 && __ballot(own)) { atomicAdd(&hist[128 + ml], 1ull); atomicAdd(&hist[192 + ms], 1ull); atomicAdd(&hist[256 + min(T, 63)], 1ull); }
}
int wcsph_list_stats(wcsph_ctx *ctx, int64_t hist[5][64]) {
  NEED_GRID();
  if (!hist) return WCSPH_EARG;
  unsigned long long *dh = nullptr;
  HIPCHK(hipMalloc((void **)&dh, 5 * 64 * 8));
  hipError_t e0 = hipMemsetAsync(dh, 0, 5 * 64 * 8, ctx->stream);
  if (ctx->d.nl > 0) hipLaunchKernelGGL(k_list_stats, dim3(nblk8(ctx->d.nl)), dim3(BLOCK), 0, ctx->stream, ctx->d, dh);
  hipError_t e1 = hipMemcpyAsync(hist, dh, 5 * 64 * 8, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(dh);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) return ctx->fail(WCSPH_EHIP, "wcsph_list_stats");
  return WCSPH_OK;
}

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
  h->n = nf; h->W = 0;
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

int wcsph_halo_pack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, void *d_out) {
  NEED_SETUP();
  HaloFields h;
  int rc = halo_fields(ctx, n_fields, fields, &h);
  if (rc) return rc;
  if (n <= 0) return WCSPH_OK;
  hipLaunchKernelGGL(k_halo_pack, dim3(nblk(n * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, d_idx, n, (float *)d_out);
  LAUNCHCHK();
  return WCSPH_OK;
}
int wcsph_halo_unpack(wcsph_ctx *ctx, int32_t n_fields, const int32_t *fields, const int32_t *d_idx, int32_t n, const void *d_in) {
  NEED_SETUP();
  HaloFields h;
  int rc = halo_fields(ctx, n_fields, fields, &h);
  if (rc) return rc;
  if (n <= 0) return WCSPH_OK;
  hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(n * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, d_idx, n, (const float *)d_in);
  LAUNCHCHK();
  for (int k = 0; k < n_fields; k++) if (fields[k] == WCSPH_F_POS) ctx->grid_valid = false;
  ctx->pr_valid = false;
  return WCSPH_OK;
}
int wcsph_load_state(wcsph_ctx *ctx, int32_t n_owned, int32_t n_total, const float *d_rows) {
  NEED_SETUP();
  if (n_owned < 0 || n_total < n_owned || n_total > ctx->capacity) return ctx->fail(WCSPH_EARG, "load_state: counts exceed the context capacity");
  ctx->d.nl = n_total; ctx->d.n_owned = n_owned;
  if (n_total > 0) {
    if (!d_rows) return WCSPH_EARG;
    hipLaunchKernelGGL(k_load_state, dim3(nblk(n_total)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_rows, n_total, 0);
    LAUNCHCHK();
  }
  ctx->grid_valid = false;
  return WCSPH_OK;
}
int wcsph_load_state_split(wcsph_ctx *ctx, int32_t n_owned, const float *d_owned_rows, int32_t n_ghost, const float *d_ghost_rows) {
  NEED_SETUP();
  if (n_owned < 0 || n_ghost < 0 || (long)n_owned + n_ghost > ctx->capacity) return ctx->fail(WCSPH_EARG, "load_state: counts exceed the context capacity");
  if ((n_owned > 0 && !d_owned_rows) || (n_ghost > 0 && !d_ghost_rows)) return WCSPH_EARG;
  ctx->d.nl = n_owned + n_ghost; ctx->d.n_owned = n_owned;
  if (n_owned > 0) hipLaunchKernelGGL(k_load_state, dim3(nblk(n_owned)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_owned_rows, n_owned, 0);
  if (n_ghost > 0) hipLaunchKernelGGL(k_load_state, dim3(nblk(n_ghost)), dim3(BLOCK), 0, ctx->stream, ctx->d, d_ghost_rows, n_ghost, n_owned);
  LAUNCHCHK();
  ctx->grid_valid = false;
  return WCSPH_OK;
}
int wcsph_store_state(wcsph_ctx *ctx, float *d_rows) {
  NEED_SETUP();
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
    case WCSPH_PH_WARM_PRESS_SCALE: ELEM(k_warm_press_scale, dt); break;
    case WCSPH_PH_WARM_PRESS_VEL: PASS(k_warm_press_vel, dt); break;
    case WCSPH_PH_DRHO_PRESS_BEGIN: HOT_DRHO_PRESS(DRHO_BEGIN, dt); break;
    case WCSPH_PH_MAKE_K_PRESS: ELEM(k_make_k, ctx->d.kappa, 1.0f); break;
    case WCSPH_PH_DRHO_PRESS_ITER: if (ctx->d.part_mode != 2 && (rc = zero_acc(ctx, ACC_ERR))) return rc; HOT_DRHO_PRESS(DRHO_ITER, dt); break;
    case WCSPH_PH_END_PRESS: ELEM(k_end_press, dt); break;
    case WCSPH_PH_CLEAR: ELEM(k_clear_nonpressure); break;
    // tension_fixed = 0: the reference loop literally, over its raw bucket rows (bucket.h).  On a slab window the table
    // spans all ranks (far aliased entries name particles of other slabs): bucket_build refuses with a message.
    case WCSPH_PH_TENSION_NORMAL:
      if (ctx->p.tension_fixed == 0) { if ((rc = bucket_build_rows(ctx))) return rc; PASS(k_tension_normal_literal, ctx->bk); }
      else RPASS(k_tension_normal);
      break;
    case WCSPH_PH_TENSION_ACCEL:
      if (ctx->p.tension_fixed == 0) { if ((rc = bucket_build_rows(ctx))) return rc; ELEM(k_tension_accel_literal, ctx->bk); }
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
    case WCSPH_PH_UPDATE_VEL_WARM_SCALE: ELEM(k_update_vel_warm_scale, dt); break;
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
  return WCSPH_OK;
}
}  // extern "C" (templates below)
// ghost rows' gather records: (pos, 0) -- their payload (.w) arrives with the halo refreshes, their positions do not
// change during the step
__global__ void k_ghost_records(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || d.oid[i] < d.n_owned) return;
  const float4 p = d.pos[i];
  d.pk[i] = make_float4(p.x, p.y, p.z, 0.f);
  d.pr[i] = make_float4(p.x, p.y, p.z, 0.f);
}
namespace {
struct SlabRun {
  wcsph_ctx *ctx;
  const wcsph_slab_callbacks *cb;
  bool multi, overlap;
  int ilo, ihi, n_xchg = 0, n_wait = 0;
  HaloFields h;
  void hf_begin() { h.n = 0; h.W = 0; }
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
    if (ctx->n_halo_send > 0) {
      hipLaunchKernelGGL(k_halo_pack, dim3(nblk(ctx->n_halo_send * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, ctx->halo_send, ctx->n_halo_send, (float *)snd);
      LAUNCHCHK();
    }
    if ((rc = cb->halo_post(cb->user, h.W))) return ctx->fail(rc, "slab callback halo_post failed");
    n_xchg++;
    const bool split = overlap && has_pass;
    if (split) { ctx->d.part_mode = 1; ctx->d.part_x0 = ilo; ctx->d.part_x1 = ihi > ilo ? ihi : ilo; if ((rc = launch())) return rc; }
    if ((rc = cb->halo_wait(cb->user))) return ctx->fail(rc, "slab callback halo_wait failed");
    if (ctx->n_halo_ghost > 0) {
      hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ctx->n_halo_ghost * h.W)), dim3(BLOCK), 0, ctx->stream, h, (const int *)ctx->d.where, ctx->halo_ghost, ctx->n_halo_ghost, (const float *)rcv);
      if (also) hipLaunchKernelGGL(k_halo_unpack, dim3(nblk(ctx->n_halo_ghost * h.W)), dim3(BLOCK), 0, ctx->stream, *also, (const int *)ctx->d.where, ctx->halo_ghost, ctx->n_halo_ghost, (const float *)rcv);
      LAUNCHCHK();
    }
    if (!has_pass) return 0;
    if (split) ctx->d.part_mode = 2;
    rc = launch();
    ctx->d.part_mode = 0;
    return rc;
  }
  int allreduce(int slot) {   // the stripes of a reduction slot summed over the ranks, in place on the device
    int rc;
    if (multi && (rc = cb->allreduce_acc(cb->user, slot))) return ctx->fail(rc, "slab callback allreduce_acc failed");
    return 0;
  }
  int fetch() { n_wait++; return pull_scalars(ctx); }   // one host wait: every slot total + the error flags
};
}  // namespace
extern "C" {
#define SRC(x) do { if ((rc = (x))) { ctx->d.part_mode = 0; return rc; } } while (0)
#define SPASS_(kern, ...) [&]() -> int { PASS(kern, ##__VA_ARGS__); return 0; }
int wcsph_slab_step(wcsph_ctx *ctx, const wcsph_slab_callbacks *cb, int64_t n_global, int32_t interior_lo, int32_t interior_hi,
                    int32_t flags, int32_t stats[2]) {
  NEED_SETUP();
  if (n_global < 1) return WCSPH_EARG;
  SlabRun R{ctx, cb, cb && cb->world > 1, (flags & 1) != 0 && cb && cb->world > 1, interior_lo, interior_hi};
  if (R.multi && (!cb->halo_buffers || !cb->halo_post || !cb->halo_wait || !cb->allreduce_acc || !cb->allreduce_max || !cb->allreduce_words)) return WCSPH_EARG;
  Dev &d = ctx->d;
  int rc;
  const double N = (double)n_global;
  const bool tension = (flags & 2) && !(d.tension == 0.0f && d.tension_b == 0.0f);
  if (tension && ctx->p.tension_fixed == 0 && R.multi) return ctx->fail(WCSPH_EARG, "the literal compute_tension reads the reference's raw bucket rows, which span all slabs: tension_fixed = 1 on a decomposition");
  auto no_pass = []() -> int { return 0; };
  // ---- particle_data.hash_grid.update_grid(): with the reference's GLOBAL bucket table in faithful mode (HashGrid.py:108-114)
  if (d.faithful && R.multi) {
    void *words = nullptr; int64_t nw = 0;
    SRC(wcsph_grid_update_begin(ctx));
    SRC(wcsph_bucket_hist_pack(ctx, &words, &nw));
    if ((rc = cb->allreduce_words(cb->user, words, nw))) return ctx->fail(rc, "slab callback allreduce_words failed");
    SRC(wcsph_bucket_hist_unpack(ctx));
    SRC(wcsph_grid_update_end(ctx));
  } else SRC(wcsph_grid_update(ctx));
  if (d.nl == 0) { if (stats) stats[0] = stats[1] = 0; return WCSPH_OK; }
  if (d.n_owned < d.nl) ELEM(k_ghost_records);
  // ---- compute_density(); compute_dfsph_coff(); and the first pass of solve_vel_divergence() (dfsph.py:131-146): they read
  // pos_j / vel_j only, which the ghost rows brought with them
  PASS(k_density_coff_warm, ctx->dt);
  ctx->dv_iter = 0;
  // rho_j (field and gather record) and the rescaled kappa_v_j (the payload of the warm-start velocity pass) in ONE message
  R.hf_begin(); R.hf_add(d.rho, 1, 1); R.hf_add((const float *)d.pk + 3, 1, 4);
  {
    HaloFields also = R.h;                    // column 0 once more: -> pr.w
    also.ptr[0] = (const float *)d.pr + 3; also.stride[0] = 4;
    SRC(R.xp(false, no_pass, &also));
  }
  ctx->pr_valid = true;                       // (pos, rho) records of every row: own rows by the density pass, ghosts just now
  SRC(SPASS_(k_vel_from_k<1>, (const float4 *)d.pk, ctx->dt)());      // warm-start velocities (dfsph.py:420-439); ghosts' kappa_v arrived above
  double err = -0.1;
  const float dt = ctx->dt;
  R.hf_begin(); R.hf_add(d.vel, 3, 4);
  SRC(R.xp(true, SPASS_((k_drho_div<DRHO_BEGIN, true>), dt)));
  const bool vort_div = d.n_owned >= 0;       // the vorticity sums ride on the last divergence iteration (k_drho_div_iter_vort)
  while ((double)ctx->avg_density_err > err && ctx->dv_iter < 10) {
    R.hf_begin(); R.hf_add((const float *)d.pk + 3, 1, 4);           // k_j = b_j alpha_j, written by the density-change pass
    SRC(R.xp(true, SPASS_(k_vel_from_k<0>, (const float4 *)d.pk, dt, d.kappa_v)));
    R.hf_begin(); R.hf_add(d.vel, 3, 4);
    if (vort_div) SRC(R.xp(true, SPASS_(k_drho_div_iter_vort, dt)));
    else SRC(R.xp(true, SPASS_((k_drho_div<DRHO_ITER, true>), dt)));
    SRC(R.allreduce(ACC_ERR));
    SRC(R.fetch());
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    err = 0.001 * N / (double)dt;
    ctx->dv_iter++;
  }
  // ---- compute_nonpressure_force() (dfsph.py:84-103)
  ELEM(k_end_div_begin_nonpressure, dt, vort_div ? 1 : 0);
  if (tension) {
    RPASS(k_tension_normal);
    R.hf_begin(); R.hf_add(d.normal, 3, 4);
    SRC(R.xp(true, SPASS_(k_tension_accel)));
  }
  R.hf_begin(); R.hf_add(d.vel_guess, 3, 4);
  if (vort_div) SRC(R.xp(true, SPASS_(k_visc_minv_residual, dt)));
  else SRC(R.xp(true, SPASS_(k_visc_minv_residual_vort, dt)));
  SRC(R.allreduce(ACC_DZERO));                  // the first k_visc_update takes delta_zero from the device
  R.hf_begin(); R.hf_add(d.cg_dir, 3, 4);
  SRC(R.xp(true, SPASS_(k_visc_Ad, dt)));
  SRC(R.allreduce(ACC_DAD));
  ELEM(k_visc_update, 0.0f, 1);
  SRC(R.allreduce(ACC_DNEW));
  SRC(R.fetch());
  ctx->cg_delta_zero = (float)ctx->h_acc[ACC_DZERO];
  ctx->cg_delta_old = ctx->cg_delta_zero;
  ctx->cg_delta = (float)ctx->h_acc[ACC_DNEW];
  ctx->vs_iter = 1;
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
  // end_viscosity + the integration loop of compute_vorticity + the per-particle part of the CFL reduction
  const bool cfl_here = !(d.faithful && d.nl <= 1);
  if (cfl_here) { ELEM(k_finish_nonpressure, dt); ctx->cfl_done = true; }
  else { ELEM(k_visc_end, dt); ELEM(k_vorticity_integrate, dt); }
  // ---- optimize_time_step() (dfsph.py:107-129) on the GLOBAL velocity maximum
  double vmax;
  if (R.multi && cb->allreduce_max_bits) {
    // the maximum over the ranks is formed on the device (a MAX all-reduce of the bit pattern, queued behind k_cfl_store) and
    // read with the one wait this decision needs anyway.  Every rank makes the call, also one whose reference launches nothing
    // (dfsph.py:108-111 with a single liquid particle: vel_max[0] keeps its 0)
    if (d.faithful && d.nl <= 1) HIPCHK(hipMemsetAsync(d.flags + FLAG_VMAX, 0, sizeof(int), ctx->stream));
    else {
      if (ctx->cfl_done) ctx->cfl_done = false;   // k_finish_nonpressure already took the per-particle maxima
      else ELEM(k_cfl, ctx->dt);
      hipLaunchKernelGGL(k_cfl_store, dim3(1), dim3(WAVE), 0, ctx->stream, ctx->d, DtRule{nullptr, 0.f, 0, 0.0, 0.0, 0.0, 0.0});
      LAUNCHCHK();
    }
    if ((rc = cb->allreduce_max_bits(cb->user, d.flags + FLAG_VMAX))) return ctx->fail(rc, "slab callback allreduce_max_bits failed");
    SRC(pull_scalars(ctx));
    const int bits = ctx->h_flags[FLAG_VMAX];
    memcpy(&ctx->vel_max0, &bits, 4);
    R.n_wait++;
    vmax = (double)ctx->vel_max0;
  } else {
    SRC(wcsph_dfsph_cfl_reduce(ctx));
    R.n_wait++;
    vmax = (double)ctx->vel_max0;
    if (R.multi && (rc = cb->allreduce_max(cb->user, &vmax))) return ctx->fail(rc, "slab callback allreduce_max failed");
  }
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
  const float dtn = ctx->dt;
  // ---- update_vel(); solve_pressure() (dfsph.py:150-164; the warm start changes no velocity in the reference, Q7)
  ELEM(k_update_vel_warm_scale, dtn);
  PASS(k_warm_press_vel, dtn);
  ctx->pr_iter = 0;
  err = 0.0;
  R.hf_begin(); R.hf_add(d.vel, 3, 4);
  SRC(R.xp(true, SPASS_((k_drho_press<DRHO_BEGIN, true>), dtn)));
  while ((err > 0.001 || ctx->pr_iter < 2) && ctx->pr_iter < 100) {
    const bool prof = ctx->prof_on && ctx->prof_used + 2 <= ctx->prof_ev.size();
    if (prof) HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
    R.hf_begin(); R.hf_add((const float *)d.pk + 3, 1, 4);
    SRC(R.xp(true, SPASS_(k_vel_from_k<0>, (const float4 *)d.pk, dtn, d.kappa)));
    R.hf_begin(); R.hf_add(d.vel, 3, 4);
    SRC(R.xp(true, SPASS_((k_drho_press<DRHO_ITER, true>), dtn)));
    if (prof) { HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], ctx->stream)); ctx->prof_used += 2; }
    ctx->pr_iter++;
    if (ctx->pr_iter < 2) { err = 1.0; continue; }   // `pr_iter < 2` forces the second iteration whatever the first reports
    SRC(R.allreduce(ACC_ERR));
    SRC(R.fetch());
    ctx->avg_density_err = (float)ctx->h_acc[ACC_ERR];
    err = (double)ctx->avg_density_err / N;
  }
  ELEM(k_end_press_update_pos, dtn);
  if (stats) { stats[0] = R.n_xchg; stats[1] = R.n_wait; }
  return WCSPH_OK;
}
#undef SPASS_
#undef SRC

// ---- boundary sampler --------------------------------------------------------------------------------
int wcsph_sample_surface(const float *tri_xyz, int32_t n_tri, double R, uint32_t seed, int32_t trials,
                         float *out_xyz, int32_t cap, int32_t *n_out) {
  if (!tri_xyz || n_tri < 1 || !(R > 0.0) || trials < 1 || !out_xyz || cap < 0 || !n_out) return WCSPH_EARG;
  // ref: boundry.py:128-165 -- face areas / normals (float32 arrays), bounding box, candidate count
  std::vector<float> nrm((size_t)n_tri * 3), area(n_tri);
  double total = 0.0, maxa = 0.0;
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int t = 0; t < n_tri; t++) {
    const float *a = tri_xyz + 9 * (size_t)t, *b = a + 3, *c = a + 6;
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], fminf(a[k], fminf(b[k], c[k]))); mx[k] = fmaxf(mx[k], fmaxf(a[k], fmaxf(b[k], c[k]))); }
    const double d1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, d2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
    const double n[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
    const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    area[t] = (float)(len * 0.5);
    for (int k = 0; k < 3; k++) nrm[3 * (size_t)t + k] = len > 0 ? (float)(n[k] / len) : 0.f;
    total += (double)area[t];
    if ((double)area[t] > maxa) maxa = (double)area[t];
  }
  if (!(maxa > 0.0)) return WCSPH_EARG;
  const double pi_ref = 3.1415926;   // boundry.py:54
  const long ncand_l = (long)(40.0 * (total / (pi_ref * R * R)));
  if (ncand_l < 1 || ncand_l > 400000000L) return WCSPH_EARG;
  PsDev s;
  memset(&s, 0, sizeof s);
  s.n_tri = n_tri; s.n_cand = (int)ncand_l; s.seed = seed; s.radius = (float)R; s.max_area = (float)maxa;
  const double grid_r = R / sqrt(3.0);   // boundry.py:21
  s.inv_grid = (float)(1.0 / grid_r);
  s.minx = mn[0]; s.miny = mn[1]; s.minz = mn[2];
  const long dims[3] = {(long)(((double)mx[0] - mn[0]) / grid_r) + 3, (long)(((double)mx[1] - mn[1]) / grid_r) + 3, (long)(((double)mx[2] - mn[2]) / grid_r) + 3};
  if (dims[0] * dims[1] * dims[2] > 1500000000L) return WCSPH_EARG;
  s.nx = (int)dims[0]; s.ny = (int)dims[1]; s.nz = (int)dims[2]; s.ncell = s.nx * s.ny * s.nz;
  std::vector<void *> mem;
  hipError_t herr = hipSuccess;
  auto A = [&](size_t bytes) -> void * { void *p = nullptr; if (herr == hipSuccess) herr = hipMalloc(&p, bytes ? bytes : 4); if (p) { mem.push_back(p); (void)hipMemset(p, 0, bytes ? bytes : 4); } return p; };
  float *d_tri = (float *)A((size_t)n_tri * 36), *d_nrm = (float *)A((size_t)n_tri * 12), *d_area = (float *)A((size_t)n_tri * 4);
  s.pos = (float *)A((size_t)s.n_cand * 12); s.tri_id = (int *)A((size_t)s.n_cand * 4); s.cell = (int *)A((size_t)s.n_cand * 4);
  s.cnt = (int *)A(((size_t)s.ncell + 1) * 4); s.start = (int *)A(((size_t)s.ncell + 1) * 4); s.order = (int *)A((size_t)s.n_cand * 4);
  s.samp_cnt = (int *)A((size_t)s.ncell * 4); s.samp = (int *)A((size_t)s.ncell * PS_MAX_PER_CELL * 4); s.accepted = (int *)A((size_t)s.n_cand * 4);
  s.occ_start = (int *)A(28 * 4);
  int rc = WCSPH_OK;
  auto done = [&](int code) { for (void *p : mem) (void)hipFree(p); return code; };
  if (herr != hipSuccess) return done(WCSPH_EHIP);
  s.tri = d_tri; s.nrm = d_nrm; s.area = d_area;
  (void)hipMemcpy(d_tri, tri_xyz, (size_t)n_tri * 36, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_nrm, nrm.data(), (size_t)n_tri * 12, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_area, area.data(), (size_t)n_tri * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ps_init, dim3(nblk(s.n_cand)), dim3(BLOCK), 0, 0, s);
  // cell table on the host (an offline tool: one pass over the dense cell array)
  std::vector<int> cnt((size_t)s.ncell + 1), start((size_t)s.ncell + 1);
  if (hipMemcpy(cnt.data(), s.cnt, ((size_t)s.ncell + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return done(WCSPH_EHIP);
  std::vector<int> occ[27];
  int run = 0;
  for (int c = 0; c < s.ncell; c++) {
    start[c] = run; run += cnt[c];
    if (cnt[c] > 0) { const int cz = c % s.nz, cy = (c / s.nz) % s.ny, cx = c / (s.nz * s.ny); occ[cx % 3 + 3 * (cy % 3) + 9 * (cz % 3)].push_back(c); }   // boundry.py:265
  }
  start[s.ncell] = run;
  std::vector<int> occ_all, occ_start(28, 0);
  for (int g = 0; g < 27; g++) { occ_start[g] = (int)occ_all.size(); occ_all.insert(occ_all.end(), occ[g].begin(), occ[g].end()); }
  occ_start[27] = (int)occ_all.size();
  s.occ = (int *)A(occ_all.size() * 4);
  if (herr != hipSuccess) return done(WCSPH_EHIP);
  (void)hipMemcpy(s.start, start.data(), ((size_t)s.ncell + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(s.occ, occ_all.data(), occ_all.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(s.occ_start, occ_start.data(), 28 * 4, hipMemcpyHostToDevice);
  (void)hipMemset(s.cnt, 0, ((size_t)s.ncell + 1) * 4);
  hipLaunchKernelGGL(k_ps_place, dim3(nblk(s.n_cand)), dim3(BLOCK), 0, 0, s);
  hipLaunchKernelGGL(k_ps_sort_cells, dim3(nblk(s.ncell)), dim3(BLOCK), 0, 0, s);
  for (int trial = 0; trial < trials; trial++)          // boundry.py:424-458: trial_total rounds over the 27 phase groups
    for (int pg = 0; pg < 27; pg++) {
      const int n = occ_start[pg + 1] - occ_start[pg];
      if (n > 0) hipLaunchKernelGGL(k_ps_trial, dim3(nblk(n)), dim3(BLOCK), 0, 0, s, pg, trial);
    }
  std::vector<int> acc(s.n_cand);
  std::vector<float> pos((size_t)s.n_cand * 3);
  if (hipMemcpy(acc.data(), s.accepted, (size_t)s.n_cand * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(pos.data(), s.pos, (size_t)s.n_cand * 12, hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess)
    return done(WCSPH_EHIP);
  int n = 0;
  for (int i = 0; i < s.n_cand; i++)
    if (acc[i]) { if (n < cap) { out_xyz[3 * n] = pos[3 * (size_t)i]; out_xyz[3 * n + 1] = pos[3 * (size_t)i + 1]; out_xyz[3 * n + 2] = pos[3 * (size_t)i + 2]; } n++; }
  *n_out = n;
  return done(rc);
}

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
