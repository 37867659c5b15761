// api_migrate.h -- migration of owners across the cuts and the ghost rebuild of a slab whose state is resident in the library,
// INSIDE the library (round 6; no reference counterpart: dfsph.py:606-617 is single-device).
//
// Until now the driver (wcsph_amd/slab.py _redistribute_resident) gathered the travelling rows, exchanged a header and the rows
// over torch.distributed, computed the swap plan in ~20 small torch launches and called wcsph_state_apply: 0.21-0.23 ms per step
// between the step's last mail and the next step's first launch, with the GPU idle behind it, and one host wait for the header
// (profiles/r06z_ring_real.json).  Here the same bookkeeping is five launches on the context's stream with every size known on
// the host beforehand:
//   wcsph_slab_migrate_begin   the rows that travel -> the library's send buffer [go_left | near_left | go_right | near_right]
//   (exchange)                 the caller's communicator, or -- wcsph_slab_migrate -- one grouped ncclSend / ncclRecv of exact
//                              sizes: the neighbours' counts came with the step's last mail (api_slab.h tail_classify)
//   wcsph_slab_migrate_end     swap plan (one workgroup), arrivals + new ghost rows written into the slots of what left, the
//                              halo lists of the coming step built in the library's own buffers and installed
// The plan is SlabSim._swap_plan's, entry for entry (the same bijection: tests compare the two): a table of n owned rows loses
// the k rows `holes` = go_left ++ go_right and gains a arrivals.  a >= k: the first k arrivals take the holes in list order, the
// rest is appended at n...  a < k: the arrivals take the first a holes, the table shrinks by L = k - a, and the rows of its last
// L labels that stay (an arrival that took a hole up there included) swap LABELS with the open holes below the new end -- r-th
// staying tail label <-> r-th open hole below the end in list order; no row data moves.
#pragma once

struct MigLists { const long long *nl, *nr, *gl, *gr; int n_nl, n_nr, n_ml, n_mr; };
__device__ __forceinline__ int mig_hole(const MigLists &m, int q) { return (int)(q < m.n_ml ? m.gl[q] : m.gr[q - m.n_ml]); }

__device__ __forceinline__ void state_row_out(const Dev &d, int o, float *r) {
  const int i = d.where[o];
  const float4 p = d.pos[i], v = d.vel[i], w = d.omega[i], g = d.vel_guess[i];
  r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = v.x; r[4] = v.y; r[5] = v.z; r[6] = w.x; r[7] = w.y; r[8] = w.z;
  r[9] = g.x; r[10] = g.y; r[11] = g.z; r[12] = d.kappa[i]; r[13] = d.kappa_v[i];
  r[14] = __int_as_float(d.gid ? d.gid[o] : o);
}
// out = [go_left | near_left | go_right | near_right] rows of STATE_W floats.  A row is 60 bytes: the 256 rows of a block go through
// LDS so that the writes are contiguous (as k_store_state; written straight from the lanes it is a 60-byte-stride store)
__global__ void __launch_bounds__(BLOCK) k_mig_gather(Dev d, MigLists m, float *out) {
  __shared__ float sh[BLOCK * STATE_W];
  const int total = m.n_ml + m.n_nl + m.n_mr + m.n_nr;
  const int k0 = blockIdx.x * BLOCK, cnt = min(BLOCK, total - k0);
  if (cnt <= 0) return;
  if ((int)threadIdx.x < cnt) {
    int k = k0 + threadIdx.x;
    long long o;
    if (k < m.n_ml) o = m.gl[k];
    else if ((k -= m.n_ml) < m.n_nl) o = m.nl[k];
    else if ((k -= m.n_nl) < m.n_mr) o = m.gr[k];
    else o = m.nr[k - m.n_mr];
    state_row_out(d, (int)o, sh + threadIdx.x * STATE_W);
  }
  __syncthreads();
  float *dst = out + (size_t)k0 * STATE_W;
  for (int q = threadIdx.x; q < cnt * STATE_W; q += BLOCK) dst[q] = sh[q];
}
// a < k: the plan of the shrinking table.  ONE workgroup (the lists are a few thousand entries); work = hole_tail[L] | T[L] | dest[L].
// dest[j] = the label that tail label n_new + j has afterwards (itself when it is an open hole: it falls off the end).
constexpr int MIG_PLAN_THREADS = 1024;
__global__ void __launch_bounds__(MIG_PLAN_THREADS) k_mig_plan(Dev d, MigLists m, int n, int a, int *work) {
  const int k = m.n_ml + m.n_mr, L = k - a, n_new = n - L, tid = threadIdx.x, NT = MIG_PLAN_THREADS;
  int *hole_tail = work, *T = work + L, *dest = work + 2 * (size_t)L;
  for (int j = tid; j < L; j += NT) hole_tail[j] = 0;
  __syncthreads();
  for (int r = tid; r < L; r += NT) { const int h = mig_hole(m, a + r); if (h >= n_new) hole_tail[h - n_new] = 1; }
  __syncthreads();
  const int seg = (L + NT - 1) / NT, s0 = min(tid * seg, L), s1 = min(s0 + seg, L);
  int ck = 0, cb = 0;
  for (int j = s0; j < s1; j++) { ck += hole_tail[j] ? 0 : 1; cb += mig_hole(m, a + j) < n_new ? 1 : 0; }
  __shared__ int sk[MIG_PLAN_THREADS], sb[MIG_PLAN_THREADS];
  sk[tid] = ck; sb[tid] = cb;
  __syncthreads();
  for (int o = 1; o < NT; o <<= 1) {          // inclusive scan of both
    const int vk = tid >= o ? sk[tid - o] : 0, vb = tid >= o ? sb[tid - o] : 0;
    __syncthreads();
    sk[tid] += vk; sb[tid] += vb;
    __syncthreads();
  }
  int ok = sk[tid] - ck, ob = sb[tid] - cb;
  for (int r = s0; r < s1; r++) { const int h = mig_hole(m, a + r); if (h < n_new) T[ob++] = h; }
  __syncthreads();
  for (int j = s0; j < s1; j++) {
    if (hole_tail[j]) { dest[j] = n_new + j; continue; }
    const int t = n_new + j, h = T[ok++];
    dest[j] = h;
    // the two labels swap rows (k_state_relabel): label h gets the staying row, label t the dead one
    const int st = d.where[t], sh = d.where[h];
    d.where[h] = st; d.oid[st] = h;
    d.where[t] = sh; d.oid[sh] = t;
    if (d.gid) d.gid[h] = d.gid[t];
  }
}
struct MigPlan { int n, a, a_l, n_new, L; const int *dest; };   // dest == nullptr: a >= k
__device__ __forceinline__ int mig_tr(const MigPlan &p, int x) { return (p.dest && x >= p.n_new) ? p.dest[x - p.n_new] : x; }
__device__ __forceinline__ int mig_pos(const MigLists &m, const MigPlan &p, int q) {   // the label arrival q gets
  const int k = m.n_ml + m.n_mr;
  if (!p.dest) return q < k ? mig_hole(m, q) : p.n + q - k;
  return mig_tr(p, mig_hole(m, q));
}
// the halo lists of the coming step: send = [near_left' | arrivals from the left | near_right' | arrivals from the right] (what this
// rank's neighbours hold ghosts of, in the order they hold them), ghost = n_new, n_new + 1, ...
__global__ void __launch_bounds__(BLOCK) k_mig_lists(MigLists m, MigPlan p, int *send_idx, int n_send, int *ghost_idx, int n_ghost) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_ghost) ghost_idx[t] = p.n_new + t;
  if (t >= n_send) return;
  const int a_r = p.a - p.a_l;
  int v, u = t;
  if (u < m.n_nl) v = mig_tr(p, (int)m.nl[u]);
  else if ((u -= m.n_nl) < p.a_l) v = mig_pos(m, p, u);
  else if ((u -= p.a_l) < m.n_nr) v = mig_tr(p, (int)m.nr[u]);
  else { u -= m.n_nr; v = u < a_r ? mig_pos(m, p, p.a_l + u) : -1; }
  send_idx[t] = v;
}
// what arrived and the new ghost rows into the state: arrival q -> label mig_pos(q); ghost g -> label n_new + g, in the order
// [from_left beyond its a_l arrivals | go_left (kept: it sits just beyond the cut now) | from_right beyond a_r | go_right].
// A label below nl_old has a slot (a row that left, an old ghost); beyond it the row gets a slot of its own (k_state_write_rows).
struct MigRows { const float *from_l, *from_r, *sent; int n_from_l, n_from_r, a_r, to_r0; float shift_l, shift_r; int gid_base; };
__global__ void __launch_bounds__(BLOCK) k_mig_write(Dev d, MigLists m, MigPlan p, MigRows w, int n_ghost, int nl_old) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.a + n_ghost) return;
  const float *r; float shift = 0.f; int o, gid_new = -1;
  if (t < p.a) {
    o = mig_pos(m, p, t);
    if (t < p.a_l) { r = w.from_l + (size_t)t * STATE_W; shift = w.shift_l; }
    else { r = w.from_r + (size_t)(t - p.a_l) * STATE_W; shift = w.shift_r; }
  } else {
    int g = t - p.a;
    o = p.n_new + g;
    const int gl1 = w.n_from_l - p.a_l, gr1 = w.n_from_r - w.a_r;
    if (g < gl1) { r = w.from_l + (size_t)(p.a_l + g) * STATE_W; shift = w.shift_l; if (w.gid_base >= 0) gid_new = w.gid_base + g; }
    else if ((g -= gl1) < m.n_ml) r = w.sent + (size_t)g * STATE_W;
    else if ((g -= m.n_ml) < gr1) { r = w.from_r + (size_t)(w.a_r + g) * STATE_W; shift = w.shift_r; if (w.gid_base >= 0) gid_new = w.gid_base + gl1 + g; }
    else r = w.sent + (size_t)(w.to_r0 + g - gr1) * STATE_W;
  }
  int i = o;
  if (o < nl_old) i = d.where[o]; else { d.where[o] = o; d.oid[o] = o; }
  d.pos[i] = make_float4(r[0] + shift, r[1], r[2], __int_as_float(-1));
  d.vel[i] = make_float4(r[3], r[4], r[5], 0);
  d.omega[i] = make_float4(r[6], r[7], r[8], 0);
  d.vel_guess[i] = make_float4(r[9], r[10], r[11], 0);
  d.kappa[i] = r[12];
  d.kappa_v[i] = r[13];
  if (d.gid) d.gid[o] = gid_new >= 0 ? gid_new : __float_as_int(r[14]);
}

template <class T>
static int mig_reserve(wcsph_ctx *ctx, T **p, size_t *cap, size_t need) {
  if (need <= *cap && *p) return 0;
  if (*p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(*p)); *p = nullptr; *cap = 0; }
  const size_t c = need + need / 4 + 4096;
  HIPCHK(hipMalloc((void **)p, c * sizeof(T)));
  *cap = c;
  return 0;
}
static MigLists mig_lists(const wcsph_ctx::Migrate &g) {
  return MigLists{g.d_idx, g.d_idx + (size_t)g.cap, g.d_idx + 2 * (size_t)g.cap, g.d_idx + 3 * (size_t)g.cap, g.c[0], g.c[1], g.c[2], g.c[3]};
}

extern "C" {

// a periodic seam (a ring of slabs; the one-GPU test of the migration, tools/ring_real.py) for wcsph_slab_migrate, whose exchange
// the caller does not see: what arrives from the left / right has shift_left / shift_right added to x; ghost_gid_base >= 0: the ghost COPIES get global ids of their own from there on (on a
// ring of one slab a copy would otherwise share its id with the row it copies).  Defaults 0, 0, -1.
int wcsph_slab_migrate_set_seam(wcsph_ctx *ctx, float shift_left, float shift_right, int32_t ghost_gid_base) {
  if (!ctx) return WCSPH_EARG;
  ctx->mg.shift_l = shift_left; ctx->mg.shift_r = shift_right; ctx->mg.ghost_gid_base = ghost_gid_base;
  return WCSPH_OK;
}
// counts[5] / d_idx (4 lists of `cap` int64 each): a classification of the present state (wcsph_state_classify or the step's tail).
// Gathers the rows that travel; *d_to_left = [go_left | near_left] (counts[2] + counts[0] rows), *d_to_right = [go_right | near_right]
int wcsph_slab_migrate_begin(wcsph_ctx *ctx, const int64_t *d_idx, int32_t cap, const int32_t counts[5], void **d_to_left, void **d_to_right) {
  NEED_SETUP();
  if (!d_idx || cap <= 0 || !counts || !d_to_left || !d_to_right) return WCSPH_EARG;
  if (ctx->nl_phys > 0) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate_begin: the state has pending changes that no grid update has sorted in yet");
  for (int k = 0; k < 4; k++) if (counts[k] < 0 || counts[k] > cap) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate_begin: a classification list is longer than its buffer");
  if (counts[4]) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate_begin: particles moved more than a ghost layer past a cut in one step (CFL violated)");
  auto &g = ctx->mg;
  g.d_idx = (const long long *)d_idx; g.cap = cap;
  for (int k = 0; k < 4; k++) g.c[k] = counts[k];
  const int total = counts[0] + counts[1] + counts[2] + counts[3];
  int rc = mig_reserve(ctx, &g.send, &g.cap_send, (size_t)total * STATE_W);
  if (rc) return rc;
  if (total > 0) hipLaunchKernelGGL(k_mig_gather, dim3(nblk(total)), dim3(BLOCK), 0, ctx->stream, ctx->d, mig_lists(g), g.send);
  LAUNCHCHK();
  *d_to_left = g.send; *d_to_right = g.send + (size_t)(counts[2] + counts[0]) * STATE_W;
  g.begun = true;
  return WCSPH_OK;
}
// what the neighbours sent: n_from_left rows of which the first a_left change owner to this rank (the rest are its new ghosts),
// the same from the right.  out[5] = {n_owned, rows sent to / ghosts held of the left neighbour per refresh, ... right}:
// {n_owned_new, n_send_left, n_send_right, n_ghost_left, n_ghost_right}.  The halo lists are the library's own
// (wcsph_slab_halo_lists) and are installed (wcsph_slab_set_halo, and wcsph_comm_set_halo_split on a native communicator).
static int migrate_end(wcsph_ctx *ctx, const void *d_from_left, int32_t n_from_left, int32_t a_left,
                       const void *d_from_right, int32_t n_from_right, int32_t a_right, int32_t out[5], bool seam) {
  NEED_SETUP();
  auto &g = ctx->mg;
  if (!g.begun) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate_end without wcsph_slab_migrate_begin");
  if (!out || a_left < 0 || a_right < 0 || a_left > n_from_left || a_right > n_from_right || (n_from_left > 0 && !d_from_left) || (n_from_right > 0 && !d_from_right)) return WCSPH_EARG;
  g.begun = false;
  Dev &d = ctx->d;
  const MigLists m = mig_lists(g);
  const int n = d.n_owned, nl_old = d.nl, k = m.n_ml + m.n_mr, a = a_left + a_right;
  const int n_new = n + a - k;
  const int n_gl = n_from_left - a_left + m.n_ml, n_gr = n_from_right - a_right + m.n_mr, n_ghost = n_gl + n_gr;
  const int n_sl = m.n_nl + a_left, n_sr = m.n_nr + a_right, n_send = n_sl + n_sr;
  if (n_new < 0 || (long)n_new + n_ghost > ctx->capacity) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate_end: counts exceed the context capacity");
  int rc;
  if ((rc = mig_reserve(ctx, &g.halo_send, &g.cap_hs, (size_t)n_send)) || (rc = mig_reserve(ctx, &g.halo_ghost, &g.cap_hg, (size_t)n_ghost))) return rc;
  MigPlan p{n, a, a_left, n_new, 0, nullptr};
  hipStream_t s = ctx->stream;
  if (a < k) {
    const int L = k - a;
    if ((rc = mig_reserve(ctx, &g.work, &g.cap_work, 3 * (size_t)L))) return rc;
    hipLaunchKernelGGL(k_mig_plan, dim3(1), dim3(MIG_PLAN_THREADS), 0, s, d, m, n, a, g.work);
    p.L = L; p.dest = g.work + 2 * (size_t)L;
  }
  const int nlist = n_send > n_ghost ? n_send : n_ghost;
  if (nlist > 0) hipLaunchKernelGGL(k_mig_lists, dim3(nblk(nlist)), dim3(BLOCK), 0, s, m, p, g.halo_send, n_send, g.halo_ghost, n_ghost);
  MigRows w{(const float *)d_from_left, (const float *)d_from_right, g.send, n_from_left, n_from_right, a_right, m.n_ml + m.n_nl,
            seam ? g.shift_l : 0.f, seam ? g.shift_r : 0.f, seam ? g.ghost_gid_base : -1};
  if (a + n_ghost > 0) hipLaunchKernelGGL(k_mig_write, dim3(nblk(a + n_ghost)), dim3(BLOCK), 0, s, d, m, p, w, n_ghost, nl_old);
  LAUNCHCHK();
  d.n_owned = n_new; d.nl = n_new + n_ghost;
  ctx->nl_phys = nl_old > d.nl ? nl_old : d.nl;   // rows [d.nl, nl_old) by original index are dead until the sort drops them
  ctx->grid_valid = ctx->hist_valid = false;
  ctx->pr_valid = false;
  ctx->halo_send = g.halo_send; ctx->n_halo_send = n_send; ctx->halo_ghost = g.halo_ghost; ctx->n_halo_ghost = n_ghost;
  if (ctx->comm) {
    ctx->comm->n_send_left = n_sl; ctx->comm->n_ghost_left = n_gl;
    if ((rc = comm_reserve(ctx->comm, (size_t)n_send * 4, (size_t)n_ghost * 4))) return rc;
  }
  out[0] = n_new; out[1] = n_sl; out[2] = n_sr; out[3] = n_gl; out[4] = n_gr;
  return WCSPH_OK;
}
int wcsph_slab_migrate_end(wcsph_ctx *ctx, const void *d_from_left, int32_t n_from_left, int32_t a_left,
                           const void *d_from_right, int32_t n_from_right, int32_t a_right, int32_t out[5]) {
  return migrate_end(ctx, d_from_left, n_from_left, a_left, d_from_right, n_from_right, a_right, out, false);
}
// the halo lists wcsph_slab_migrate_end left installed (device pointers, int32; valid until the next migration)
int wcsph_slab_halo_lists(wcsph_ctx *ctx, void **d_send_idx, int32_t *n_send, void **d_ghost_idx, int32_t *n_ghost) {
  NEED_SETUP();
  if (!d_send_idx || !n_send || !d_ghost_idx || !n_ghost) return WCSPH_EARG;
  *d_send_idx = (void *)ctx->halo_send; *n_send = ctx->n_halo_send; *d_ghost_idx = (void *)ctx->halo_ghost; *n_ghost = ctx->n_halo_ghost;
  return WCSPH_OK;
}
// begin + exchange + end on the library's own communicator (wcsph_comm_init), nothing of it on the host: the message sizes follow
// from this rank's counts and the neighbours' (nbr_left[5] / nbr_right[5]: wcsph_slab_tail_neighbour_counts -- or whatever
// channel the caller has; a missing neighbour: zeros).  One grouped ncclSend / ncclRecv on the context's stream.
int wcsph_slab_migrate(wcsph_ctx *ctx, const int64_t *d_idx, int32_t cap, const int32_t counts[5], const int32_t nbr_left[5], const int32_t nbr_right[5], int32_t out[5]) {
  NEED_SETUP();
  if (!ctx->comm) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate: no native communicator (wcsph_comm_init); use wcsph_slab_migrate_begin / _end around the caller's own exchange");
  if (!nbr_left || !nbr_right || !out || !counts) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if ((c->left < 0 && (counts[0] || counts[2])) || (c->right < 0 && (counts[1] || counts[3])))
    return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate: rows classified towards a neighbour this communicator does not have");
  // from the left neighbour: what it sends to ITS right = [its go_right | its near_right]; from the right: [its go_left | its near_left]
  const int a_l = c->left >= 0 ? nbr_left[3] : 0, n_fl = c->left >= 0 ? nbr_left[3] + nbr_left[1] : 0;
  const int a_r = c->right >= 0 ? nbr_right[2] : 0, n_fr = c->right >= 0 ? nbr_right[2] + nbr_right[0] : 0;
  if (a_l < 0 || a_r < 0 || n_fl < a_l || n_fr < a_r || nbr_left[4] || nbr_right[4]) return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate: implausible neighbour counts");
  void *to_l = nullptr, *to_r = nullptr;
  int rc = wcsph_slab_migrate_begin(ctx, d_idx, cap, counts, &to_l, &to_r);
  if (rc) return rc;
  auto &g = ctx->mg;
  if ((rc = mig_reserve(ctx, &g.recv, &g.cap_recv, (size_t)(n_fl + n_fr) * STATE_W))) { g.begun = false; return rc; }
  const size_t sl = (size_t)(counts[2] + counts[0]) * STATE_W, sr = (size_t)(counts[3] + counts[1]) * STATE_W;
  if (c->left == c->rank && c->right == c->rank && ((size_t)n_fr * STATE_W != sl || (size_t)n_fl * STATE_W != sr)) {
    g.begun = false;
    return ctx->fail(WCSPH_EARG, "wcsph_slab_migrate on a loop-back communicator: what goes out to the left must be as long as what comes in from the right");
  }
  if (c->halo_pending && (rc = ncb_wait(c))) { g.begun = false; return rc; }
  rc = comm_exchange_sized(c, g.send, sl, sr, g.recv, (size_t)n_fl * STATE_W, (size_t)n_fr * STATE_W, ctx->stream);
  if (rc) { g.begun = false; return rc; }
  return migrate_end(ctx, g.recv, n_fl, a_l, g.recv + (size_t)n_fl * STATE_W, n_fr, a_r, out, true);
}

}  // extern "C"
