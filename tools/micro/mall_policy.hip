// Infinity Cache (MALL) allocation probe for gfx950: does a pass over a big "bystander" buffer B evict a 96 MB buffer A
// from the 256 MB memory-side cache, depending on the cache-policy bits of B's loads / stores and on how B was allocated?
// Sequence per row: read A (plain, warm it) -> pass over B (mode) -> read A again, TIMED.  Reference rows: A re-read right
// after itself (resident) and after a plain 1 GB sweep (evicted).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mall_policy tools/micro/mall_policy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_read(const v4u *a, size_t n, unsigned *out) {
  unsigned s = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v4u v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
  if (s == 0x12345678u) out[0] = s;
}
// latency-bound re-read: one wave per CU-slot, each 1 KB chunk's address depends on the previous chunk's data
__global__ void __launch_bounds__(64) k_read_lat(const v4u *a, size_t n, unsigned *out) {
  unsigned s = 0;
  size_t i = blockIdx.x * 64 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 64;
  while (i < n) { const v4u v = a[i]; s += v.y; i += step + (v.x & 2u); }   // memory holds 0x01 bytes: (v.x & 2) == 0
  if (s == 0x12345678u) out[0] = s;
}
enum { LD_PLAIN, LD_NT, LD_SC1, LD_SC0SC1, LD_SC0SC1NT, ST_PLAIN, ST_NT, ST_SC1, ST_SC0SC1NT, NMODE };
static const char *mode_name[NMODE] = {"load plain", "load nt", "load sc1", "load sc0 sc1", "load sc0 sc1 nt", "store plain", "store nt", "store sc1", "store sc0 sc1 nt"};
template <int M>
__global__ void __launch_bounds__(256) k_pass(v4u *b, size_t n, unsigned *out) {
  unsigned s = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    v4u *p = b + i;
    v4u v = {(unsigned)i, 1u, 2u, 3u};
    if (M == LD_PLAIN) v = *p;
    else if (M == LD_NT) v = __builtin_nontemporal_load(p);
    else if (M == LD_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (M == LD_SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (M == LD_SC0SC1NT) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (M == ST_PLAIN) *p = v;
    else if (M == ST_NT) __builtin_nontemporal_store(v, p);
    else if (M == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (M == ST_SC0SC1NT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    s += v.x;
  }
  if (s == 0x12345678u) out[0] = s;
}
template <int M> void pass(v4u *b, size_t n, unsigned *out) { hipLaunchKernelGGL((k_pass<M>), dim3(256 * 16), dim3(256), 0, 0, b, n, out); }
static void run_pass(int m, v4u *b, size_t n, unsigned *out) {
  switch (m) {
    case LD_PLAIN: pass<LD_PLAIN>(b, n, out); break;
    case LD_NT: pass<LD_NT>(b, n, out); break;
    case LD_SC1: pass<LD_SC1>(b, n, out); break;
    case LD_SC0SC1: pass<LD_SC0SC1>(b, n, out); break;
    case LD_SC0SC1NT: pass<LD_SC0SC1NT>(b, n, out); break;
    case ST_PLAIN: pass<ST_PLAIN>(b, n, out); break;
    case ST_NT: pass<ST_NT>(b, n, out); break;
    case ST_SC1: pass<ST_SC1>(b, n, out); break;
    case ST_SC0SC1NT: pass<ST_SC0SC1NT>(b, n, out); break;
  }
}
static int g_lat = 0;
static float time_read(const v4u *a, size_t n, unsigned *out, hipEvent_t e0, hipEvent_t e1) {
  float ms;
  hipEventRecord(e0);
  if (g_lat) hipLaunchKernelGGL(k_read_lat, dim3(1024), dim3(64), 0, 0, a, n, out);
  else hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, a, n, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main(int argc, char **argv) {
  const size_t MB = 1 << 20;
  const size_t sa = (argc > 1 ? atoi(argv[1]) : 96) * MB, sb = (argc > 2 ? atoi(argv[2]) : 224) * MB, sbig = 1024 * MB;
  v4u *A, *B[3], *BIG; unsigned *out;
  CK(hipMalloc(&A, sa)); CK(hipMalloc(&BIG, sbig)); CK(hipMalloc(&out, 64));
  const char *alloc_name[3] = {"hipMalloc", "hipExtMallocWithFlags(Uncached)", "hipExtMallocWithFlags(Finegrained)"};
  CK(hipMalloc(&B[0], sb));
  if (hipExtMallocWithFlags((void **)&B[1], sb, hipDeviceMallocUncached) != hipSuccess) B[1] = nullptr;
  if (hipExtMallocWithFlags((void **)&B[2], sb, hipDeviceMallocFinegrained) != hipSuccess) B[2] = nullptr;
  CK(hipMemset(A, 1, sa)); CK(hipMemset(BIG, 1, sbig));
  for (int k = 0; k < 3; k++) if (B[k]) CK(hipMemset(B[k], 1, sb));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t na = sa / 16, nb = sb / 16, nbig = sbig / 16;
  g_lat = argc > 3 ? atoi(argv[3]) : 0;
  printf("%s re-read.  ", g_lat ? "LATENCY-bound (dependent 1 KB chunks, 1024 waves)" : "bandwidth-bound");
  printf("A = %zu MB (plain reads), bystander B = %zu MB; us for the re-read of A (best of 3)\n", sa / MB, sb / MB);
  float best = 1e9f;
  for (int r = 0; r < 4; r++) { time_read(A, na, out, e0, e1); float t = time_read(A, na, out, e0, e1); if (t < best) best = t; }
  printf("%-64s %8.1f us  (%.2f TB/s)\n", "A right after A (resident)", best, sa / best * 1e-6);
  best = 1e9f;
  for (int r = 0; r < 3; r++) { time_read(A, na, out, e0, e1); pass<LD_PLAIN>(BIG, nbig, out); float t = time_read(A, na, out, e0, e1); if (t < best) best = t; }
  printf("%-64s %8.1f us  (%.2f TB/s)\n", "A after a plain 1 GB sweep (evicted)", best, sa / best * 1e-6);
  for (int al = 0; al < 3; al++) {
    if (!B[al]) { printf("%s: allocation failed\n", alloc_name[al]); continue; }
    for (int m = 0; m < NMODE; m++) {
      best = 1e9f;
      float tb = 0;
      for (int r = 0; r < 3; r++) {
        time_read(A, na, out, e0, e1);
        hipEventRecord(e0); run_pass(m, B[al], nb, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tb, e0, e1);
        float t = time_read(A, na, out, e0, e1);
        if (t < best) best = t;
      }
      char name[128];
      snprintf(name, sizeof name, "B %s, %s", alloc_name[al], mode_name[m]);
      printf("%-64s %8.1f us  (%.2f TB/s)   [B pass %.1f us]\n", name, best, sa / best * 1e-6, tb * 1e3);
    }
  }
  return 0;
}
