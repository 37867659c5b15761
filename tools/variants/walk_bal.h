// tools/variants/walk_bal.h -- EXPERIMENT, not part of the product build (profiles/r03b_balanced_walk_experiment.md).
// The "segmented" form of the neighbour walk asked for by VERDICT r02 item 4: a lane walks only the part of its list that
// 3/4 of its wave shares, the rest becomes work items of the whole workgroup (LDS map + LDS accumulators).
// Needs ANCHORED stream blocks (every block's first delta relative to i / nhead[i].y instead of the previous block's last
// entry: NbWriter::entry resets `run` at block boundaries, the decoders likewise) and a walk_range() that takes slot ranges;
// both are in the same commit.  Anchoring alone costs the developed flow ~4 % (+4 hop slots per block that starts in another
// outer tile plane, 128 k particles away once the liquid covers the floor).
// Parity-green (61 GPU tests) and slower than the plain walk in both states; kept so that the measurement can be repeated:
// paste below walk() in wcsph_amd/csrc/device.h and switch k_vel_from_k / k_drho_press / k_drho_div to it (the kernel
// side of the experiment is in git history: commit "walk_bal experiment").
// ---- balanced walk (the solver-loop passes) -------------------------------------------------------------------------
// A wave runs as many trips as its LONGEST list.  At rest the lists of a wave's 64 particles are equally long; once the
// fluid has broken up a wave mixes bulk, surface and spray particles and most lanes idle through the long lists' tail
// (+75 % wave-level memory instructions at step 1200 of the C3 dam break).  Here a lane walks only the COMMON part of
// its own list -- the first T blocks, T = the block count at least 3/4 of the wave's lanes reach (one T per wave and
// stream kind) -- and the blocks beyond it become work items of the whole 256-thread workgroup: compacted into an LDS
// map by a prefix sum, taken round-robin by all lanes (a lane then decodes ANOTHER particle's block: blocks are anchored,
// see NbWriter), the partial sums of a block added to its particle's accumulators in LDS.  The particle's own lane
// adds its common-part sums to the same accumulators and reads the totals back: "segmented reduction" of the north star.
// A workgroup without tail items (everything at rest) pays one barrier.
//
// o / acc: the kernel's own-particle inputs and its accumulators; the lambdas gl / fl / fs use them BY REFERENCE, so the
// tail swaps another particle's inputs in (load_own(i2)) and zeroed accumulators, runs the same lambdas, and flushes.
// Every thread of the block must call this (barriers), own or not.
#define BAL_MAP_CAP 4096   /* tail items of one workgroup (uint16 each) */
template <int NACC> struct BalShared { float acc[NACC][BLOCK]; unsigned short map[BAL_MAP_CAP]; int wsum[BLOCK / WAVE]; };
__device__ __forceinline__ int bal_common_blocks(int nb) {   // the largest T that at least 3/4 of the wave's lanes reach
  int T = 0;
  for (int t = 1; t <= 32; t++) {
    if (__popcll(__ballot(nb >= t)) < (WAVE * 3) / 4) break;
    T = t;
  }
  return T;
}
template <int U, bool NT, int NACC, class Own, class LO, class GL, class FL, class FS>
__device__ __forceinline__ void walk_bal(const Dev &d, int i, bool own, float4 pi, const float4 *P, Own &o, float (&acc)[NACC], LO &&load_own,
                                         GL &&gl, FL &&fl, FS &&fs) {
  __shared__ BalShared<NACC> sh;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int2 head = own ? d.nhead[i] : make_int2(0, 0);
  const int cl = head.x & 0xffff, cs = (int)((unsigned)head.x >> 16);
  const int nbl = (cl + 7) >> 3, nbs = (cs + 7) >> 3;
  const int T = bal_common_blocks(nbl), TS = bal_common_blocks(nbs);
  const int tl = max(nbl - T, 0), ts = max(nbs - TS, 0), cnt = tl + ts;
  int inc = cnt;
#pragma unroll
  for (int s = 1; s < WAVE; s <<= 1) { const int t = __shfl_up(inc, s, WAVE); if (lane >= s) inc += t; }
  if (lane == WAVE - 1) sh.wsum[wv] = inc;
#pragma unroll
  for (int a = 0; a < NACC; a++) sh.acc[a][tid] = 0.0f;
  __syncthreads();
  int off = inc - cnt, W = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / WAVE; w++) { if (w < wv) off += sh.wsum[w]; W += sh.wsum[w]; }
  const bool bal = W > 0 && W <= BAL_MAP_CAP;   // uniform over the workgroup
  if (bal) {   // item = p | kind << 8 | block << 9
    for (int t = 0; t < tl; t++) sh.map[off + t] = (unsigned short)(tid | ((T + t) << 9));
    for (int t = 0; t < ts; t++) sh.map[off + tl + t] = (unsigned short)(tid | 256 | ((TS + t) << 9));
  }
  // the common part (the whole list when the workgroup has no tail or too long a one)
  if (own) walk_range<U, NT>(d, i, xyz(pi), P, head.y, 0, bal ? min(cl, T << 3) : cl, 0, bal ? min(cs, TS << 3) : cs, gl, fl, fs);
  if (!bal) return;
  if (own) {
#pragma unroll
    for (int a = 0; a < NACC; a++) atomicAdd(&sh.acc[a][tid], acc[a]);
  }
  __syncthreads();
  const int i0 = i - tid;
  for (int w = tid; w < W; w += BLOCK) {
    const unsigned it = sh.map[w];
    const int p2 = it & 255, b = it >> 9, i2 = i0 + p2;
    const int2 h2 = d.nhead[i2];
    const float4 pi2 = d.pos[i2];
    o = load_own(i2);
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0f;
    if (it & 256) walk_range<U, NT>(d, i2, xyz(pi2), P, h2.y, 0, 0, b << 3, min((int)((unsigned)h2.x >> 16), (b + 1) << 3), gl, fl, fs);
    else walk_range<U, NT>(d, i2, xyz(pi2), P, h2.y, b << 3, min(h2.x & 0xffff, (b + 1) << 3), 0, 0, gl, fl, fs);
#pragma unroll
    for (int a = 0; a < NACC; a++) atomicAdd(&sh.acc[a][p2], acc[a]);
  }
  __syncthreads();
  if (own) {
    o = load_own(i);
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = sh.acc[a][tid];
  }
}
