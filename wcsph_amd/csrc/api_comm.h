// api_comm.h -- the native data plane of the slab decomposition (wcsph_comm_init / wcsph_halo_exchange): RCCL calls
// issued by the library itself, no host code between the library and the wire.  New work: the reference is single-device
// (dfsph.py:606-617).  RCCL is resolved at run time (dlopen of librccl.so.1: the one a torch process has already loaded, else
// ROCm's), so the library has no link dependency on it, single-GPU users never load it -- and building the library needs no
// RCCL headers: the nine entry points are declared here (ABI of rccl.h, NCCL 2.x: the types below are stable across releases).
// The communicator behind wcsph_slab_step is then this file's implementation of wcsph_slab_callbacks (include/wcsph.h).
//
// Streams (round 6): a halo refresh runs on the communicator's OWN stream (`halo`), fenced by events against the context's
// stream: pack (context) -> ev_packed -> grouped ncclSend / ncclRecv (halo) -> ev_landed -> unpack (context).  The pass on the
// slab interior that wcsph_slab_step queues between post and wait therefore runs WHILE the rows travel (until round 5 the
// exchange sat on the context's stream and serialised with it).  The all-reduces of the reduction slots stay on the context's
// stream (their result is the next thing that stream needs); the bucket histogram's all-reduce has its own side stream.
#pragma once
#include <dlfcn.h>
#include <mutex>

// ---- the part of rccl.h this file uses (types and values as in /opt/rocm/include/rccl/rccl.h) -------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;     // ncclSuccess = 0
typedef int ncclDataType_t;   // ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8
typedef int ncclRedOp_t;      // ncclSum = 0, ncclMax = 2
enum { ncclSuccess = 0, ncclInt32 = 2, ncclFloat = 7, ncclDouble = 8, ncclSum = 0, ncclMax = 2 };
static_assert(sizeof(ncclUniqueId) == WCSPH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  std::string why;            // why the library is not usable (empty when it is)
};
// resolved once per process (std::call_once: two contexts may initialise from two threads); a missing symbol closes the
// handle again and the reason stays readable for every later caller
static RcclApi *rccl_api(std::string *why) {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // the copy already in the process (torch's), if any
    for (int k = 0; !h && k < 3; k++) h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!h) { const char *e = dlerror(); api.why = std::string("dlopen(librccl.so.1): ") + (e ? e : "not found"); return; }
#define RCCL_SYM(f) if (api.why.empty() && !(api.f = (decltype(api.f))dlsym(h, "nccl" #f))) api.why = "librccl lacks nccl" #f;
    RCCL_SYM(GetUniqueId) RCCL_SYM(CommInitRank) RCCL_SYM(CommDestroy) RCCL_SYM(GetErrorString) RCCL_SYM(AllReduce)
    RCCL_SYM(Send) RCCL_SYM(Recv) RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd)
#undef RCCL_SYM
    if (!api.why.empty()) { dlclose(h); return; }
    api.h = h;
  });
  if (!api.h) { if (why) *why = api.why; return nullptr; }
  return &api;
}
// the reason wcsph_comm_unique_id failed (it has no context to leave a message in): wcsph_comm_last_error()
static std::mutex g_comm_err_mu;
static std::string g_comm_err;

struct NativeComm {
  RcclApi *api = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, left = -1, right = -1;
  int n_send_left = 0, n_ghost_left = 0;   // how the lists of wcsph_slab_set_halo split between the neighbours (left first)
  struct Buf { float *send = nullptr, *recv = nullptr; size_t cap_send = 0, cap_recv = 0; };
  Buf buf;
  hipStream_t halo = nullptr;              // the halo refreshes' sends / receives run here, beside the pass on the slab interior
  hipEvent_t ev_packed = nullptr, ev_landed = nullptr, ev_chain = nullptr;
  bool halo_pending = false;               // an exchange is in flight on `halo` that the context's stream has not been made to wait for
  // wcsph_comm_set_mode: 0 = refreshes inline on the context's stream; 1 = on the halo stream beside the interior pass, the
  // context's stream waits for them before the unpack; 2 = exchange, unpack AND the pass on the rows next to the cuts on the halo
  // stream beside the interior pass (api_slab.h SlabRun::xp), one join at the end.  Which is fastest depends on what an exchange
  // costs on the machine (profiles/r06_overlap.md: on ONE GPU, where it is a 16 us copy, mode 0 without the split): bench.py
  // --gpus N measures the candidates during warm-up and keeps the fastest.
  int mode = 1;
  hipStream_t side = nullptr;              // the bucket histogram's all-reduce runs here beside the sort and the list build
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  bool words_pending = false;
  double *d_scalar = nullptr;
  long n_exchanges = 0, n_allreduces = 0;
  wcsph_slab_callbacks cb = {};
  wcsph_ctx *ctx = nullptr;
};

#define NCCLCHK(call)                                                                                          \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) { c->ctx->fail(WCSPH_EHIP, (std::string(#call ": ") + c->api->GetErrorString(r_)).c_str()); return WCSPH_EHIP; } \
  } while (0)
#define CHIPCHK(call)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) { c->ctx->fail(WCSPH_EHIP, #call, e_); return WCSPH_EHIP; }                          \
  } while (0)

// the split of the halo lists against the lists themselves (wcsph_slab_set_halo may have shrunk them since
// wcsph_comm_set_halo_split; (n_send - n_send_left) * W would wrap around as size_t), and on a loop-back communicator what goes
// out to the left must be as long as what comes in from the right
static int comm_check_split(NativeComm *c, int n_send, int n_ghost) {
  if (c->n_send_left < 0 || c->n_ghost_left < 0 || c->n_send_left > n_send || c->n_ghost_left > n_ghost)
    return c->ctx->fail(WCSPH_EARG, "halo exchange: wcsph_comm_set_halo_split() names more rows to the left than the halo lists hold (call it after every wcsph_slab_set_halo)");
  if (c->left == c->rank && c->right == c->rank && (c->n_send_left != n_ghost - c->n_ghost_left || n_send - c->n_send_left != c->n_ghost_left))
    return c->ctx->fail(WCSPH_EARG, "halo exchange on a loop-back communicator: the rows sent to the left must be as many as the rows received from the right (and vice versa)");
  return 0;
}
// one grouped exchange with the slab neighbours on stream s: rows [0, n_send_left) of d_send go left, the rest right; rows
// [0, n_ghost_left) of d_recv come from the left, the rest from the right.  The receives are posted right-first: on a loop-back
// communicator (left == right == this rank: the one-GPU test of the data plane) what goes out to the left comes in from the
// right, as on a ring.
static int comm_exchange_sized(NativeComm *c, const float *d_send, size_t sl, size_t sr, float *d_recv, size_t gl, size_t gr, hipStream_t s) {
  NCCLCHK(c->api->GroupStart());
  if (c->left >= 0 && sl) NCCLCHK(c->api->Send(d_send, sl, ncclFloat, c->left, c->comm, s));
  if (c->right >= 0 && sr) NCCLCHK(c->api->Send(d_send + sl, sr, ncclFloat, c->right, c->comm, s));
  if (c->right >= 0 && gr) NCCLCHK(c->api->Recv(d_recv + gl, gr, ncclFloat, c->right, c->comm, s));
  if (c->left >= 0 && gl) NCCLCHK(c->api->Recv(d_recv, gl, ncclFloat, c->left, c->comm, s));
  NCCLCHK(c->api->GroupEnd());
  c->n_exchanges++;
  return 0;
}
static int comm_exchange(NativeComm *c, int W, const float *d_send, float *d_recv, int n_send, int n_ghost, hipStream_t s) {
  int rc = comm_check_split(c, n_send, n_ghost);
  if (rc) return rc;
  return comm_exchange_sized(c, d_send, (size_t)c->n_send_left * W, (size_t)(n_send - c->n_send_left) * W,
                             d_recv, (size_t)c->n_ghost_left * W, (size_t)(n_ghost - c->n_ghost_left) * W, s);
}
// the five classification counts of this rank's next migration (api_slab.h tail_classify) to both slab neighbours, theirs into
// d_nbr[0..5) (left) and d_nbr[8..13) (right); a missing neighbour's block stays as it is (zeroed at allocation)
static int comm_counts_exchange(NativeComm *c, const int *d_mine, int *d_nbr, hipStream_t s) {
  NCCLCHK(c->api->GroupStart());
  if (c->left >= 0) NCCLCHK(c->api->Send(d_mine, 5, ncclInt32, c->left, c->comm, s));
  if (c->right >= 0) NCCLCHK(c->api->Send(d_mine, 5, ncclInt32, c->right, c->comm, s));
  if (c->right >= 0) NCCLCHK(c->api->Recv(d_nbr + 8, 5, ncclInt32, c->right, c->comm, s));
  if (c->left >= 0) NCCLCHK(c->api->Recv(d_nbr, 5, ncclInt32, c->left, c->comm, s));
  NCCLCHK(c->api->GroupEnd());
  return 0;    // (not counted in n_exchanges: that is the halo refreshes and the row migrations)
}
// The halo stream has the HIGHEST priority the device offers: the few workgroups of a send / receive kernel should be placed as
// soon as they are launched, not when the thousands of workgroups of the interior pass queued beside them drain.  (Measured, round
// 6, profiles/r06_overlap.md: on one GPU RCCL's kernel stretches to the length of its companion pass at EITHER priority -- a plain
// copy or a generic kernel of the same footprint does not --; the priority is kept because it is what a communication stream
// should ask for, not because it was seen to help.)
static hipError_t create_halo_stream(hipStream_t *s) {
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);   // (numerically lowest = highest priority)
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}
// ---- wcsph_slab_callbacks on RCCL (nothing waits on the host) -----------------------------------------------------------
// ONE send / receive buffer pair for every row width.  It is safe because every refresh is awaited before the next one is
// packed: ev_landed is recorded behind the WHOLE group (sends included), halo_wait makes the context's stream wait for it, and
// the next refresh's pack kernel is queued on that stream afterwards -- so pack k+1 follows send k, and receive k+1 (behind
// ev_packed k+1) follows unpack k.  Sized for rows of 4 floats with a quarter of slack when the halo lists are set
// (wcsph_comm_set_halo_split), so that no allocation happens between the exchanges of a step.
static int comm_reserve(NativeComm *c, size_t ns, size_t ng) {
  NativeComm::Buf &b = c->buf;
  if (ns > b.cap_send || !b.send) {
    if (b.send) { CHIPCHK(hipStreamSynchronize(c->ctx->stream)); CHIPCHK(hipStreamSynchronize(c->halo)); CHIPCHK(hipFree(b.send)); b.send = nullptr; }
    b.cap_send = ns + ns / 4 + 4096;
    CHIPCHK(hipMalloc((void **)&b.send, b.cap_send * sizeof(float)));
  }
  if (ng > b.cap_recv || !b.recv) {
    if (b.recv) { CHIPCHK(hipStreamSynchronize(c->ctx->stream)); CHIPCHK(hipStreamSynchronize(c->halo)); CHIPCHK(hipFree(b.recv)); b.recv = nullptr; }
    b.cap_recv = ng + ng / 4 + 4096;
    CHIPCHK(hipMalloc((void **)&b.recv, b.cap_recv * sizeof(float)));
  }
  return 0;
}
static int ncb_wait(void *u);
static int ncb_buffers(void *u, int32_t W, void **snd, void **rcv) {
  NativeComm *c = (NativeComm *)u;
  int rc;
  if (c->halo_pending && (rc = ncb_wait(u))) return rc;   // (a caller that posted without waiting: the buffers are about to be overwritten)
  if ((rc = comm_reserve(c, (size_t)c->ctx->n_halo_send * W, (size_t)c->ctx->n_halo_ghost * W))) return rc;
  *snd = c->buf.send; *rcv = c->buf.recv;
  return 0;
}
// post: the sends / receives start on the halo stream as soon as the pack kernel (the last thing on the context's stream) is
// done; whatever the caller queues on the context's stream next runs beside them
static int ncb_post(void *u, int32_t W) {
  NativeComm *c = (NativeComm *)u;
  if (c->ctx->n_halo_send == 0 && c->ctx->n_halo_ghost == 0) { c->n_exchanges++; return 0; }   // nothing travels: no call, no event hop
  // inline: the exchange sits between the pack and the unpack on the context's stream -- no event hop (two of ~12 us each on
  // this platform), nothing beside it: the cheaper form when the pass is not split (profiles/r06_overlap.md)
  if (c->mode == 0) return comm_exchange(c, W, c->buf.send, c->buf.recv, c->ctx->n_halo_send, c->ctx->n_halo_ghost, c->ctx->stream);
#ifdef WCSPH_DEV
  // diagnostic (development builds): the loop-back exchange as two plain copies on the halo stream instead of RCCL's kernel
  if (c->left == c->rank && c->right == c->rank && getenv("WCSPH_COMM_LOOPBACK_MEMCPY")) {
    CHIPCHK(hipEventRecord(c->ev_packed, c->ctx->stream));
    CHIPCHK(hipStreamWaitEvent(c->halo, c->ev_packed, 0));
    const size_t sl = (size_t)c->n_send_left * W, sr = (size_t)(c->ctx->n_halo_send - c->n_send_left) * W, gl = (size_t)c->n_ghost_left * W;
    if (sl) CHIPCHK(hipMemcpyAsync(c->buf.recv + gl, c->buf.send, sl * sizeof(float), hipMemcpyDeviceToDevice, c->halo));
    if (sr) CHIPCHK(hipMemcpyAsync(c->buf.recv, c->buf.send + sl, sr * sizeof(float), hipMemcpyDeviceToDevice, c->halo));
    CHIPCHK(hipEventRecord(c->ev_landed, c->halo));
    c->halo_pending = true; c->n_exchanges++;
    return 0;
  }
#endif
  CHIPCHK(hipEventRecord(c->ev_packed, c->ctx->stream));
  CHIPCHK(hipStreamWaitEvent(c->halo, c->ev_packed, 0));
  int rc = comm_exchange(c, W, c->buf.send, c->buf.recv, c->ctx->n_halo_send, c->ctx->n_halo_ghost, c->halo);
  if (rc) return rc;
  CHIPCHK(hipEventRecord(c->ev_landed, c->halo));
  c->halo_pending = true;
  return 0;
}
// wait: the context's stream goes on (unpack, then the pass on the rows next to the cuts) once the group has completed
static int ncb_wait(void *u) {
  NativeComm *c = (NativeComm *)u;
  if (c->halo_pending) { CHIPCHK(hipStreamWaitEvent(c->ctx->stream, c->ev_landed, 0)); c->halo_pending = false; }
  return 0;
}
static int ncb_acc(void *u, int32_t slot) {
  NativeComm *c = (NativeComm *)u;
  double *p = c->ctx->d.acc + slot * ACC_STRIPES;
  NCCLCHK(c->api->AllReduce(p, p, ACC_STRIPES, ncclDouble, ncclSum, c->comm, c->ctx->stream));
  c->n_allreduces++;
  return 0;
}
static int native_allreduce_slots(NativeComm *c, int slot0, int nslots) {   // a contiguous range of reduction slots in one call
  double *p = c->ctx->d.acc + slot0 * ACC_STRIPES;
  NCCLCHK(c->api->AllReduce(p, p, (size_t)nslots * ACC_STRIPES, ncclDouble, ncclSum, c->comm, c->ctx->stream));
  c->n_allreduces++;
  return 0;
}
static int ncb_max(void *u, double *v) {   // (only without allreduce_max_bits: never on this communicator; kept complete)
  NativeComm *c = (NativeComm *)u;
  CHIPCHK(hipMemcpyAsync(c->d_scalar, v, sizeof(double), hipMemcpyHostToDevice, c->ctx->stream));
  NCCLCHK(c->api->AllReduce(c->d_scalar, c->d_scalar, 1, ncclDouble, ncclMax, c->comm, c->ctx->stream));
  CHIPCHK(hipMemcpyAsync(v, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream));
  CHIPCHK(hipStreamSynchronize(c->ctx->stream));
  c->n_allreduces++;
  return 0;
}
static int ncb_words(void *u, void *d_words, int64_t n) {
  NativeComm *c = (NativeComm *)u;
  NCCLCHK(c->api->AllReduce(d_words, d_words, (size_t)n, ncclInt32, ncclSum, c->comm, c->ctx->stream));
  c->n_allreduces++;
  return 0;
}
static int ncb_max_bits(void *u, void *d_word) {
  NativeComm *c = (NativeComm *)u;
  NCCLCHK(c->api->AllReduce(d_word, d_word, 1, ncclInt32, ncclMax, c->comm, c->ctx->stream));
  c->n_allreduces++;
  return 0;
}
// the packed bucket histogram's all-reduce on the side stream: ordered behind what the context's stream holds now (the packed
// words), in flight while the library queues the scan, the sort and the list build; _wait makes the context's stream wait for it
static int ncb_words_post(void *u, void *d_words, int64_t n) {
  NativeComm *c = (NativeComm *)u;
  CHIPCHK(hipEventRecord(c->ev_ready, c->ctx->stream));
  CHIPCHK(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  NCCLCHK(c->api->AllReduce(d_words, d_words, (size_t)n, ncclInt32, ncclSum, c->comm, c->side));
  CHIPCHK(hipEventRecord(c->ev_done, c->side));
  c->words_pending = true;
  c->n_allreduces++;
  return 0;
}
static int ncb_words_wait(void *u) {
  NativeComm *c = (NativeComm *)u;
  if (c->words_pending) { CHIPCHK(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0)); c->words_pending = false; }
  return 0;
}

extern "C" {

int wcsph_comm_unique_id(void *id_out) {
  if (!id_out) return WCSPH_EARG;
  std::string why;
  RcclApi *api = rccl_api(&why);
  ncclUniqueId id;
  ncclResult_t r = ncclSuccess;
  if (api && (r = api->GetUniqueId(&id)) != ncclSuccess) why = std::string("ncclGetUniqueId: ") + api->GetErrorString(r);
  if (!api || r != ncclSuccess) {
    std::lock_guard<std::mutex> g(g_comm_err_mu);
    g_comm_err = "RCCL is not available: " + why;
    return WCSPH_EHIP;
  }
  memcpy(id_out, &id, sizeof id);
  return WCSPH_OK;
}
// why the last wcsph_comm_unique_id() of this process failed (it has no context to leave a message in); "" if none did
const char *wcsph_comm_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> g(g_comm_err_mu);
  copy = g_comm_err;
  return copy.c_str();
}
int wcsph_comm_destroy(wcsph_ctx *ctx) {
  if (!ctx) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if (!c) return WCSPH_OK;
  hipStreamSynchronize(ctx->stream);
  if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
  if (c->halo) { hipStreamSynchronize(c->halo); hipStreamDestroy(c->halo); }
  if (c->ev_packed) hipEventDestroy(c->ev_packed);
  if (c->ev_landed) hipEventDestroy(c->ev_landed);
  if (c->ev_chain) hipEventDestroy(c->ev_chain);
  if (c->ev_ready) hipEventDestroy(c->ev_ready);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  if (c->buf.send) hipFree(c->buf.send);
  if (c->buf.recv) hipFree(c->buf.recv);
  if (c->d_scalar) hipFree(c->d_scalar);
  if (c->comm) c->api->CommDestroy(c->comm);
  delete c;
  ctx->comm = nullptr;
  return WCSPH_OK;
}
int wcsph_comm_init(wcsph_ctx *ctx, const void *unique_id, int32_t rank, int32_t nranks) {
  if (!ctx || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return WCSPH_EARG;
  if (ctx->comm) return ctx->fail(WCSPH_EARG, "wcsph_comm_init() called twice (wcsph_comm_destroy() first)");
  std::string why;
  RcclApi *api = rccl_api(&why);
  if (!api) return ctx->fail(WCSPH_EHIP, ("RCCL is not available: " + why).c_str());
  NativeComm *c = new NativeComm();
  c->api = api; c->ctx = ctx; c->rank = rank; c->nranks = nranks;
  c->left = rank > 0 ? rank - 1 : -1;
  c->right = rank < nranks - 1 ? rank + 1 : -1;
  ctx->comm = c;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  ncclResult_t r = api->CommInitRank(&c->comm, nranks, id, rank);   // on the calling thread's current device: the context's
  if (r != ncclSuccess) {
    ctx->fail(WCSPH_EHIP, (std::string("ncclCommInitRank: ") + api->GetErrorString(r)).c_str());
    c->comm = nullptr; wcsph_comm_destroy(ctx);
    return WCSPH_EHIP;
  }
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || create_halo_stream(&c->halo) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_landed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess || hipMalloc((void **)&c->d_scalar, 2 * sizeof(double)) != hipSuccess) {
    wcsph_comm_destroy(ctx);
    return ctx->fail(WCSPH_EHIP, "wcsph_comm_init: stream / event / scratch allocation");
  }
  c->cb.user = c; c->cb.world = nranks;
  c->cb.halo_buffers = ncb_buffers; c->cb.halo_post = ncb_post; c->cb.halo_wait = ncb_wait;
  c->cb.allreduce_acc = ncb_acc; c->cb.allreduce_max = ncb_max; c->cb.allreduce_words = ncb_words; c->cb.allreduce_max_bits = ncb_max_bits;
  c->cb.allreduce_words_post = ncb_words_post; c->cb.allreduce_words_wait = ncb_words_wait;
  return WCSPH_OK;
}
int wcsph_comm_set_neighbours(wcsph_ctx *ctx, int32_t left, int32_t right) {
  if (!ctx || !ctx->comm) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if (left >= c->nranks || right >= c->nranks) return ctx->fail(WCSPH_EARG, "wcsph_comm_set_neighbours: no such rank");
  c->left = left < 0 ? -1 : left; c->right = right < 0 ? -1 : right;
  return WCSPH_OK;
}
int wcsph_comm_set_mode(wcsph_ctx *ctx, int32_t mode) {
  if (!ctx || !ctx->comm || mode < 0 || mode > 2) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if (c->halo_pending) { int rc = ncb_wait(c); if (rc) return rc; }
  c->mode = mode;
  return WCSPH_OK;
}
int wcsph_comm_set_halo_split(wcsph_ctx *ctx, int32_t n_send_left, int32_t n_ghost_left) {
  if (!ctx || !ctx->comm || n_send_left < 0 || n_ghost_left < 0) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if (n_send_left > ctx->n_halo_send || n_ghost_left > ctx->n_halo_ghost) return ctx->fail(WCSPH_EARG, "wcsph_comm_set_halo_split: more rows to the left than wcsph_slab_set_halo() listed");
  c->n_send_left = n_send_left; c->n_ghost_left = n_ghost_left;
  return comm_reserve(c, (size_t)ctx->n_halo_send * 4, (size_t)ctx->n_halo_ghost * 4);
}
// a communicator outlives the context it was created on: a rebalance of the decomposition rebuilds the context (new cell window)
// and hands the communicator over
int wcsph_comm_detach(wcsph_ctx *ctx, void **handle) {
  if (!ctx || !handle) return WCSPH_EARG;
  *handle = nullptr;
  NativeComm *c = ctx->comm;
  if (!c) return WCSPH_OK;
  CHIPCHK(hipStreamSynchronize(ctx->stream));
  CHIPCHK(hipStreamSynchronize(c->side));
  CHIPCHK(hipStreamSynchronize(c->halo));
  c->halo_pending = false;
  c->ctx = nullptr; ctx->comm = nullptr;
  *handle = c;
  return WCSPH_OK;
}
int wcsph_comm_attach(wcsph_ctx *ctx, void *handle) {
  if (!ctx || !handle) return WCSPH_EARG;
  if (ctx->comm) return ctx->fail(WCSPH_EARG, "wcsph_comm_attach: the context has a communicator");
  NativeComm *c = (NativeComm *)handle;
  c->ctx = ctx; ctx->comm = c;
  c->n_send_left = c->n_ghost_left = 0;
  return WCSPH_OK;
}
int wcsph_halo_exchange(wcsph_ctx *ctx, int32_t W, const void *d_send, int32_t n_send, void *d_recv, int32_t n_ghost) {
  if (!ctx || !ctx->comm || W < 1 || n_send < 0 || n_ghost < 0) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  if ((n_send > 0 && !d_send) || (n_ghost > 0 && !d_recv)) return WCSPH_EARG;
  // (the stand-alone entry point is stream-ordered on the CONTEXT's stream: the caller's buffers, the caller's ordering)
  return comm_exchange(c, W, (const float *)d_send, (float *)d_recv, n_send, n_ghost, ctx->stream);
}
// dtype_op: 0 = int32 SUM, 1 = float64 SUM, 2 = int32 MAX; in place on the device, on the context's stream
int wcsph_comm_allreduce(wcsph_ctx *ctx, void *d_buf, int64_t n, int32_t dtype_op) {
  if (!ctx || !ctx->comm || !d_buf || n < 0 || dtype_op < 0 || dtype_op > 2) return WCSPH_EARG;
  NativeComm *c = ctx->comm;
  NCCLCHK(c->api->AllReduce(d_buf, d_buf, (size_t)n, dtype_op == 1 ? ncclDouble : ncclInt32, dtype_op == 2 ? ncclMax : ncclSum, c->comm, ctx->stream));
  c->n_allreduces++;
  return WCSPH_OK;
}
int wcsph_comm_stats(wcsph_ctx *ctx, int64_t out[2]) {
  if (!ctx || !out) return WCSPH_EARG;
  out[0] = ctx->comm ? ctx->comm->n_exchanges : 0; out[1] = ctx->comm ? ctx->comm->n_allreduces : 0;
  return WCSPH_OK;
}

}  // extern "C"
#undef NCCLCHK
#undef CHIPCHK
