// dfsph_kernels.h -- the @ti.kernel loops of dfsph.py as hand-written gfx950 kernels.
// One thread per cell-sorted liquid particle; every pass iterates the compact per-step neighbour list of the particle
// (16-bit delta streams built by the grid update, device.h walk()) instead of the reference's 8 KiB/particle row.
#pragma once
#include "device.h"

#define WAVES6
// Cache policy of the density pass, the first walk after the list build (profiles/r03_experiments.md (11), (15)): its stream
// loads are plain -- they keep the stream the list build just wrote in the Infinity Cache for the walks that follow -- and its
// (pos, rho) records, which nobody gathers before the last divergence iteration, are stored non-temporally so that they do
// not displace it.  The (pos, k) record of the divergence iteration that also takes the vorticity sums likewise: in three
// steps of four it is the LAST iteration and nobody reads the record.
#define LOAD_I                                        \
  const int i = sorted_index(d);                       \
  const bool live = i < d.nl;                         \
  const float4 pi = live ? d.pos[i] : make_float4(0, 0, 0, __int_as_float(-1)); \
  const bool own = live && owned(d, i) && part_ok(d, pi);   /* one context owning every row (uniform): no oid read */
// for the walks that gather the (pos, rho) records: the particle's own position comes from its record too (pi.w = rho_i
// then) unless a slab decomposition needs the packed cell in pos.w -- 32 MB less read, and kept out of the cache, per pass
#define LOAD_I_PR                                     \
  const int i = sorted_index(d);                       \
  const bool live = i < d.nl;                         \
  const bool whole = d.part_mode == 0;                \
  const float4 pi = live ? (whole ? d.pr[i] : d.pos[i]) : make_float4(0, 0, 0, __int_as_float(-1)); \
  const bool own = live && owned(d, i) && (whole || part_ok(d, pi));
#define RHO_I_PR (whole ? pi.w : ld_once(d.rho + i))

// ref: dfsph.py:248-262 compute_density
__global__ void __launch_bounds__(BLOCK) k_density(Dev d) {
  LOAD_I
  if (!own) return;
  float sl = 0.f, ss = 0.f;
  walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); }, [&](int, float3, float r2, NoPayload) { sl += cubic_W(d, r2); },
          [&](int, float3, float r2) { ss += cubic_W(d, r2); });
  const float rho = d.VL0 * d.rho_L0 * (d.mk_h3 + sl) + d.VS0 * d.rho_S0 * ss;
  d.rho[i] = rho;
  d.pr[i] = make_float4(pi.x, pi.y, pi.z, rho);
}

// ref: dfsph.py:345-372 compute_dfsph_coff
__global__ void __launch_bounds__(BLOCK) k_coff(Dev d) {
  LOAD_I
  if (!own) return;
  float3 sg = f3(0, 0, 0);
  float sq = 0.f;
  walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); },
          [&](int, float3 r, float r2, NoPayload) {
            float g = cubic_gradW_scale(d, r2) * d.VL0;
            sq += g * g * r2;
            sg = sg + r * g;
          },
          [&](int, float3 r, float r2) { sg = sg + r * (cubic_gradW_scale(d, r2) * d.VS0); });
  sq += dot3(sg, sg);
  d.alpha[i] = (sq > d.eps) ? -1.0f / sq : 0.0f;
}

// fused density + alpha (one walk instead of two); used by wcsph_dfsph_step
__global__ void __launch_bounds__(BLOCK) k_density_coff(Dev d) {
  LOAD_I
  if (!own) return;
  float sl = 0.f, ss = 0.f, sq = 0.f;
  float3 sg = f3(0, 0, 0);
  walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); },
          [&](int, float3 r, float r2, NoPayload) {
            sl += cubic_W(d, r2);
            float g = cubic_gradW_scale(d, r2) * d.VL0;
            sq += g * g * r2;
            sg = sg + r * g;
          },
          [&](int, float3 r, float r2) {
            ss += cubic_W(d, r2);
            sg = sg + r * (cubic_gradW_scale(d, r2) * d.VS0);
          });
  const float rho = d.VL0 * d.rho_L0 * (d.mk_h3 + sl) + d.VS0 * d.rho_S0 * ss;
  d.rho[i] = rho;
  d.pr[i] = make_float4(pi.x, pi.y, pi.z, rho);
  sq += dot3(sg, sg);
  d.alpha[i] = (sq > d.eps) ? -1.0f / sq : 0.0f;
}

// wcsph_dfsph_step only: compute_density + compute_dfsph_coff + the first pass of warmstart_divergence
// (update_drho_divergence, dfsph.py:418, with the kappa_v rescale of :416) in ONE walk -- all three read only
// pos_j / vel_j over the same frozen list -- and the (pos, kappa_v) gather record of the warm-start velocity pass.
__global__ void __launch_bounds__(BLOCK) k_density_coff_warm(Dev d, float dt) {
  LOAD_I
  if (!own) return;
  float sl = 0.f, ss = 0.f, sq = 0.f, a = 0.f;
  float3 sg = f3(0, 0, 0);
  const float3 vi = xyz(d.vel[i]);
  walk<8>(d, i, pi, d.pos, [&](int j, float4) { return nb_ld(d.vel, j); },
          [&](int, float3 r, float r2, float4 vj) {
            sl += cubic_W(d, r2);
            const float gw = cubic_gradW_scale(d, r2);
            const float g = gw * d.VL0;
            sq += g * g * r2;
            sg = sg + r * g;
            a += d.VL0 * gw * dot3(vi - xyz(vj), r);
          },
          [&](int, float3 r, float r2) {
            ss += cubic_W(d, r2);
            const float gw = cubic_gradW_scale(d, r2);
            sg = sg + r * (gw * d.VS0);
            a += d.VS0 * gw * dot3(vi, r);
          });
  const float rho = d.VL0 * d.rho_L0 * (d.mk_h3 + sl) + d.VS0 * d.rho_S0 * ss;
  st_once(d.rho + i, rho);
  st_once(d.pr + i, make_float4(pi.x, pi.y, pi.z, rho));
  sq += dot3(sg, sg);
  st_once(d.alpha + i, (sq > d.eps) ? -1.0f / sq : 0.0f);
  const float kv = 0.5f * fmaxf(ld_once(d.kappa_v + i) / dt, -0.5f * d.rho_L0 * d.rho_L0);
  st_once(d.kappa_v + i, kv);
  a = fmaxf(a, 0.0f);
  if (ld_once(d.ncount + i) < 20) a = 0.0f;  // raw list length, Q2
  st_once(d.adv_rho + i, a);
  d.pk[i] = make_float4(pi.x, pi.y, pi.z, kv);
}

// ---- density-change passes ------------------------------------------------------------------
enum { DRHO_PLAIN = 0, DRHO_WARM = 1, DRHO_BEGIN = 2, DRHO_ITER = 3 };

// ref: dfsph.py:374-392 update_drho_divergence, inlined at :418 (WARM), :443 (BEGIN), :475 (ITER)
// FUSEK (fused solver loops only): the epilogue also emits next iteration's (pos, k_i) gather record,
// k_i = b_i * alpha_i (dfsph.py:455-456 / 523-525), so that no separate k_make_k launch is needed;
// kappa += k_i then happens in the prologue of the k_vel_from_k launch that consumes it.
template <int MODE, bool FUSEK = false>
__global__ void __launch_bounds__(BLOCK) WAVES6 k_drho_div(Dev d, float dt) {
  LOAD_I
  float a = 0.f;
  if (own) {
    if (MODE == DRHO_WARM) st_once(d.kappa_v + i, 0.5f * fmaxf(ld_once(d.kappa_v + i) / dt, -0.5f * d.rho_L0 * d.rho_L0));
    const float3 vi = xyz(d.vel[i]);
    walk<8>(d, i, pi, d.pos, [&](int j, float4) { return nb_ld(d.vel, j); },
            [&](int, float3 r, float r2, float4 vj) { a += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(vj), r); },
            [&](int, float3 r, float r2) { a += d.VS0 * cubic_gradW_scale(d, r2) * dot3(vi, r); });
    a = fmaxf(a, 0.0f);
    if (ld_once(d.ncount + i) < 20) a = 0.0f;  // raw list length, Q2
    st_once(d.adv_rho + i, a);
    float al = ld_once(d.alpha + i);
    if (MODE == DRHO_BEGIN) { al = al / dt; st_once(d.alpha + i, al); st_once(d.kappa_v + i, 0.0f); }
    if (FUSEK) d.pk[i] = make_float4(pi.x, pi.y, pi.z, a * al);
  }
  if (MODE == DRHO_ITER) block_accumulate(ACC_SLOT(d, ACC_ERR), a);  // ref: dfsph.py:476 avg_density_err += adv_rho
}

// wcsph_dfsph_step only: k_drho_div<DRHO_ITER, true> + the sums of compute_vorticity (dfsph.py:307-328, see k_vorticity).
// The last divergence iteration already gathers pos_j and vel_j with the velocities compute_vorticity will see (nothing
// between the end of solve_vel_divergence and compute_vorticity changes vel, omega or rho): riding along costs one more
// gather per entry (omega_j; rho_j comes with the (pos, rho) record) instead of two in the viscosity walk.  Every
// iteration recomputes the sums, the last one stands.  dv is parked in cg_s (free until the PCG update) because d_vel is
// cleared after this kernel (k_end_div_begin_nonpressure adds it back).
__global__ void __launch_bounds__(BLOCK) k_drho_div_iter_vort(Dev d, float dt) {
  LOAD_I_PR
  float a = 0.f;
  if (own) {
    const float3 vi = xyz(d.vel[i]), wi = xyz(d.omega[i]);
    const float rhoi = RHO_I_PR;
    float3 dom = f3(0, 0, 0), dv = f3(0, 0, 0);
    const float c1 = -1.0f / dt * d.vort_init * d.visc_omega * d.mass;
    const float c2 = d.vort_coff / rhoi * d.mass;
    const float c3 = d.vort_coff / rhoi * d.vort_init * d.mass;
    struct VW { float4 v, w; };
    walk<4>(d, i, pi, d.pr, [&](int j, float4 p) { VW q; q.v = nb_ld(d.vel, j); q.w = nb_ld(d.omega, j); q.w.w = p.w; return q; },
            [&](int, float3 r, float r2, VW q) {
              const float g = cubic_gradW_scale(d, r2);
              const float3 dvel = vi - xyz(q.v), dw = wi - xyz(q.w);
              a += d.VL0 * g * dot3(dvel, r);
              dom = dom + dw * (c1 * frcp(q.w.w) * cubic_W(d, r2));
              dv = dv + cross3(dw, r) * (c2 * g);
              dom = dom + cross3(dvel, r) * (c3 * g);
            },
            [&](int, float3 r, float r2) {
              const float g = cubic_gradW_scale(d, r2);
              a += d.VS0 * g * dot3(vi, r);
              dv = dv + cross3(wi, r) * (d.vort_coff / rhoi * d.rho_L0 * d.VS0 * g);
              dom = dom + cross3(vi, r) * (d.vort_coff / rhoi * d.vort_init * d.rho_L0 * d.VL0 * g);
            });
    const int nc = ld_once(d.ncount + i);
    a = fmaxf(a, 0.0f);
    if (nc < 20) a = 0.0f;  // raw list length, Q2
    st_once(d.adv_rho + i, a);
    st_once(d.pk + i, make_float4(pi.x, pi.y, pi.z, a * ld_once(d.alpha + i)));
    dom = dom + wi * (-2.0f * d.vort_init * d.vort_coff * (float)nc);
    st_once(d.d_omega + i, f4(dom, 0));
    st_once(d.cg_s + i, f4(dv, 0));
  }
  block_accumulate(ACC_SLOT(d, ACC_ERR), a);  // ref: dfsph.py:476 avg_density_err += adv_rho
}

// ref: dfsph.py:394-412 update_drho_pressure, inlined at :513 (BEGIN), :545 (ITER)
// Fused ITER launches take the geometry -- the particle's own position and the gathered ones -- from the (pos, k) records
// the previous density-change pass wrote and store only the record's k: the solver loop then never touches `pos`, and
// its working set (streams + records + vel + five scalars, ~230 MB at C3) stays inside the 256 MB Infinity Cache across
// the V / D launches of an iteration instead of cycling through it (DESIGN.md 4 "working set").  Gathers read xyz only,
// the epilogue writes w only: no launch reads a word another thread of the same launch writes.
template <int MODE, bool FUSEK = false>
__global__ void __launch_bounds__(BLOCK) WAVES6 k_drho_press(Dev d, float dt) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  constexpr bool REC = FUSEK && MODE == DRHO_ITER;
  const int i = sorted_index(d);
  const bool live = i < d.nl;
  const float4 pi = live ? (REC ? d.pk[i] : d.pos[i]) : make_float4(0, 0, 0, __int_as_float(-1));
  const bool own = live && owned(d, i) && (REC ? (d.part_mode == 0 || part_ok(d, d.pos[i])) : part_ok(d, pi));
  float e = 0.f;
  if (own) {
    const float3 vi = xyz(d.vel[i]);
    float t = 0.f;
    walk<8>(d, i, pi, REC ? d.pk : d.pos, [&](int j, float4) { return nb_ld(d.vel, j); },
            [&](int, float3 r, float r2, float4 vj) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi - xyz(vj), r); },
            [&](int, float3 r, float r2) { t += d.VL0 * cubic_gradW_scale(d, r2) * dot3(vi, r); });
    // REC (the solver loop): rho, alpha, adv_rho belong to the loop's cache-resident working set; elsewhere they are
    // touched once
    float a = fmaxf(1.0f, (REC ? d.rho[i] : ld_once(d.rho + i)) / d.rho_L0 + dt * t);
    if (REC) d.adv_rho[i] = a; else st_once(d.adv_rho + i, a);
    e = a - 1.0f;
    float al = REC ? d.alpha[i] : ld_once(d.alpha + i);
    if (MODE == DRHO_BEGIN) { al = al / dt / dt; d.alpha[i] = al; st_once(d.kappa + i, 0.0f); }
    if (REC) reinterpret_cast<float *>(d.pk + i)[3] = e * al;
    else if (FUSEK) d.pk[i] = make_float4(pi.x, pi.y, pi.z, e * al);
  }
  if (MODE == DRHO_ITER) block_accumulate(ACC_SLOT(d, ACC_ERR), e);  // ref: dfsph.py:546
}

// first half of divergence_iter / pressure_iter, per-particle part (ref: dfsph.py:456-458, 525-527):
// k_i = b_i * alpha_i ; kappa_i += k_i.  kbuf holds k for the neighbour loop that follows.
__global__ void k_make_k(Dev d, float *kap, float bias) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float k = (d.adv_rho[i] - bias) * d.alpha[i];
  d.kbuf[i] = k;
  float4 p = d.pos[i];
  d.pk[i] = make_float4(p.x, p.y, p.z, k);  // (pos, k) record: one 16-byte gather per neighbour in k_vel_from_k
  if (owned(d, i)) kap[i] += k;
}

// (pos.xyz, rho) records for the viscosity / vorticity / tension walks, for every row (ghosts included), when rho did not
// come from this context's own density pass
__global__ void k_pack_pr(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float4 p = d.pos[i];
  d.pr[i] = make_float4(p.x, p.y, p.z, d.rho[i]);
}
// (pos.xyz, kb) records for k_vel_from_k when k comes from an existing array (warm start: kappa_v)
__global__ void k_pack_k(Dev d, const float *kb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float4 p = d.pos[i];
  d.pk[i] = make_float4(p.x, p.y, p.z, kb[i]);
}

// neighbour loop of divergence_iter / pressure_iter / warmstart_divergence_vel
// (ref: dfsph.py:460-471, 529-541, 420-439): v_i += dt * sum (k_i + k_j) V gradW with the eps gates.
// COND 0: always; COND 1: only where adv_rho_i > 0 (divergence warm start, :421)
// kap != nullptr (fused solver loops): kappa_i += k_i here instead of in k_make_k (dfsph.py:458 / 527)
template <int COND>
__global__ void __launch_bounds__(BLOCK) k_vel_from_k(Dev d, const float4 *pk, float dt, float *kap = nullptr) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  if (kap) acc_clear(ACC_SLOT(d, ACC_ERR));   // fused solver loops: the density-change pass launched next sums into it
  // The pass is HBM-bound on its own inputs: the particle's position comes with its (pos, k) record, and
  // pos[i] / oid[i] are read only where ownership is not trivial (slab decomposition) -- 40 B per particle less.
  const int i = sorted_index(d);
  if (i >= d.nl) return;
  if (d.n_owned < d.nl || d.part_mode != 0) {
    if (!(owned(d, i) && part_ok(d, d.pos[i]))) return;
  }
  if (COND == 1 && !(ld_once(d.adv_rho + i) > 0.0f)) return;
  const float4 pi = pk[i];
  const float ki = pi.w;
  if (kap) st_once(kap + i, ld_once(kap + i) + ki);
  float3 acc = f3(0, 0, 0);
  const bool ks = fabsf(ki) > d.eps;
  walk<8>(d, i, pi, pk, [](int, float4 p) { return p.w; },
          [&](int, float3 r, float r2, float kj) {
            float s = ki + kj;
            if (fabsf(s) > d.eps) acc = acc + r * (s * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) {
            if (ks) acc = acc + r * (ki * (d.VS0 / d.VL0) * cubic_gradW_scale(d, r2));
          });
  float4 v = d.vel[i];
  const float f = dt * d.VL0;
  v.x += f * acc.x; v.y += f * acc.y; v.z += f * acc.z;
  d.vel[i] = v;
}

// ref: dfsph.py:487-508 warmstart_pressure, literal (Q7: the gate adv_rho > rho_L0 practically never
// opens; the solid branch reads kappa_v).  kappa is rescaled first (:490).
// gen != 0: also raises flags[FLAG_WARM] = gen where the gate of the velocity pass opens (k_warm_press_vel_flagged)
__global__ void k_warm_press_scale(Dev d, float dt, int gen) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  d.kappa[i] = fmaxf(d.kappa[i] / dt / dt, -0.5f * d.rho_L0 * d.rho_L0);
  if (gen && d.adv_rho[i] > d.rho_L0) d.flags[FLAG_WARM] = gen;
}
__device__ __forceinline__ void warm_press_vel_one(const Dev &d, int i, float4 pi, float dt) {
  if (!(d.adv_rho[i] > d.rho_L0)) return;
  const float ki = d.kappa[i], kvi = d.kappa_v[i];
  float3 acc = f3(0, 0, 0);
  walk<4>(d, i, pi, d.pos, [&](int j, float4) { return nb_ld(d.kappa, j); },
          [&](int, float3 r, float r2, float kj) {
            float s = ki + kj;
            if (fabsf(s) > d.eps) acc = acc + r * (dt * s * d.VL0 * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) {
            if (fabsf(ki) > d.eps) acc = acc + r * (dt * kvi * d.VS0 * cubic_gradW_scale(d, r2));
          });
  float4 v = d.vel[i];
  v.x += acc.x; v.y += acc.y; v.z += acc.z;
  d.vel[i] = v;
}
__global__ void __launch_bounds__(BLOCK) k_warm_press_vel(Dev d, float dt) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  LOAD_I
  if (!own) return;
  warm_press_vel_one(d, i, pi, dt);
}
// wcsph_dfsph_step: k_update_vel_warm_scale, which runs just before with the same adv_rho, raises flags[FLAG_WARM] to the step's
// generation number when ANY particle passes the gate above (Q7: practically never); otherwise the workgroups of this small
// grid-stride launch read one word and leave (the full-size launch costs ~10 us per step for nothing)
__global__ void __launch_bounds__(BLOCK) k_warm_press_vel_flagged(Dev d, float dt, int gen) {
  PRED_GATE(d);
  if (d.flags[FLAG_WARM] != gen) return;
  DT_OVERRIDE(d, dt);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.nl; i += gridDim.x * blockDim.x) {
    const float4 pi = d.pos[i];
    if (owned(d, i) && part_ok(d, pi)) warm_press_vel_one(d, i, pi, dt);
  }
}

// ref: dfsph.py:480-484, 549-552
__global__ void k_end_div(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  d.kappa_v[i] *= dt;
  d.alpha[i] *= dt;
}
__global__ void k_end_press(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  d.kappa[i] *= dt * dt;
}

// ref: dfsph.py:167-178 reset_param
__global__ void k_reset(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  d.vel[i] = make_float4(0, 0, 0, 0);
  d.omega[i] = make_float4(0, 0, 0, 0);
  d.pressure[i] = 0.f; d.kappa_v[i] = 0.f; d.kappa[i] = 0.f;
}
// ref: dfsph.py:333-336 clear_nonpressure
__global__ void k_clear_nonpressure(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  d.d_vel[i] = make_float4(d.gx, d.gy, d.gz, 0);
}

// ---- surface tension (ref: dfsph.py:264-305) ------------------------------------------------
// Fixed semantics (DESIGN.md "Quirks not reproduced"): the reference multiplies the running normal
// by h once per list entry (Q8, order dependent garbage) and applies the curvature term to every raw
// list entry; here n_i = h * sum_j (m/rho_j) gradW and the curvature term runs over the support.
// The cohesion value is discarded exactly as in the reference (Q9).
__global__ void __launch_bounds__(BLOCK) k_tension_normal(Dev d) {
  LOAD_I
  if (!own) return;
  float3 n = f3(0, 0, 0);
  walk<8>(d, i, pi, d.pr, [](int, float4 p) { return p.w; },
          [&](int, float3 r, float r2, float rhoj) { n = n + r * (d.mass * frcp(rhoj) * cubic_gradW_scale(d, r2)); },
          [&](int, float3, float) {});
  d.normal[i] = f4(n * d.h, 0);
}
// ref: kernels/AdhesionKernel.py:20-29
__device__ __forceinline__ float adhesion_W(const Dev &d, float r) {
  float res = 0.f;
  if (r * r <= d.h2 && r > 0.5f * d.h) res = d.adh_mk * powf(fmaxf(-4.0f * r * r / d.h + 6.0f * r - 2.0f * d.h, 0.0f), 0.25f);
  return res;
}
__global__ void __launch_bounds__(BLOCK) k_tension_accel(Dev d) {
  LOAD_I
  if (!own) return;
  const float3 ni = xyz(d.normal[i]);
  const float rhoi = d.rho[i];
  float3 acc = f3(0, 0, 0);
  walk<4>(d, i, pi, d.pos, [&](int j, float4) { float4 n = nb_ld(d.normal, j); n.w = nb_ld(d.rho, j); return n; },
       [&](int, float3, float r2, float4 nj) {
         float kij = 2.0f * d.rho_L0 * frcp(rhoi + nj.w);
         if (r2 <= d.h2) acc = acc + (ni - xyz(nj)) * (-d.tension * kij);
       },
       [&](int j, float3 r, float r2) {
         float3 pj = xyz(d.spos[j]);
         float3 c = pj - f3(0.0f, 0.5f, 0.0f);
         float dis = sqrtf(dot3(c, c));
         if (r2 > d.eps && dis < 0.26f) {  // scene hack of the reference, dfsph.py:300-302
           float rl = sqrtf(r2);
           acc = acc + r * (-d.tension_b * d.rho_S0 * d.VS0 * adhesion_W(d, rl) / rl);
         }
       });
  float4 a = d.d_vel[i];
  a.x += acc.x; a.y += acc.y; a.z += acc.z;
  d.d_vel[i] = a;
}

// ---- implicit viscosity PCG (ref: dfsph.py:181-246, 339-343) --------------------------------
// ref: dfsph.py:181-195 get_viscosity_Ax
__device__ __forceinline__ float3 visc_Ax(const Dev &d, int i, float4 pi, const float4 *x, float dt, float rhoi) {
  const float3 xi = xyz(x[i]);
  const float e = 0.01f * d.h2;
  float3 ret = f3(0, 0, 0);
  const float cl = d.dim_coff * d.visc * d.mass;
  const float cs = d.dim_coff * d.visc_b * d.rho_S0 / rhoi * d.VS0;
  walk<8>(d, i, pi, d.pr, [&](int j, float4 p) { float4 xj = nb_ld(x, j); xj.w = p.w; return xj; },
          [&](int, float3 r, float r2, float4 xj) {
            float s = cl * dot3(xi - xyz(xj), r) * frcp(xj.w * (r2 + e));
            ret = ret + r * (s * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) {
            float s = cs * dot3(xi, r) * frcp(r2 + e);
            ret = ret + r * (s * cubic_gradW_scale(d, r2));
          });
  return xi - ret * (dt / rhoi);
}
// ref: dfsph.py:199-200
__global__ void k_visc_guess(Dev d) {
  acc_clear(ACC_SLOT(d, ACC_DZERO));   // summed into by the residual kernel launched next
  acc_clear(ACC_SLOT(d, ACC_DAD));     // ... and by the first k_visc_Ad
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float4 g = d.vel_guess[i], v = d.vel[i];
  d.vel_guess[i] = make_float4(g.x + v.x, g.y + v.y, g.z + v.z, 0);
}
// ref: dfsph.py:202-215 block-Jacobi preconditioner (I - dt/rho_i * sum c gradW (x) r)^-1
__global__ void __launch_bounds__(BLOCK) k_visc_minv(Dev d, float dt) {
  LOAD_I
  if (!own) return;
  const float rhoi = d.rho[i];
  const float e = 0.01f * d.h2;
  const float cl = d.dim_coff * d.visc * d.mass;
  const float cs = d.dim_coff * d.visc_b * d.rho_S0 / rhoi * d.VS0;
  float m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;  // gradW (x) r = g * r r^T is symmetric
  auto accum = [&](float3 r, float s) {
    m00 += s * r.x * r.x; m01 += s * r.x * r.y; m02 += s * r.x * r.z;
    m11 += s * r.y * r.y; m12 += s * r.y * r.z; m22 += s * r.z * r.z;
  };
  walk<8>(d, i, pi, d.pr, [](int, float4 p) { return p.w; },
          [&](int, float3 r, float r2, float rhoj) { accum(r, cl * frcp(rhoj * (r2 + e)) * cubic_gradW_scale(d, r2)); },
          [&](int, float3 r, float r2) { accum(r, cs * frcp(r2 + e) * cubic_gradW_scale(d, r2)); });
  const float f = dt / rhoi;
  float a = 1.0f - m00 * f, b = -m01 * f, c = -m02 * f, dd = 1.0f - m11 * f, ee = -m12 * f, ff = 1.0f - m22 * f;
  // inverse of the symmetric matrix [[a b c][b dd ee][c ee ff]] by adjugate / determinant
  float A = dd * ff - ee * ee, B = -(b * ff - ee * c), C = b * ee - dd * c;
  float id = 1.0f / (a * A + b * B + c * C);
  float i11 = (a * ff - c * c) * id, i12 = -(a * ee - c * b) * id, i22 = (a * dd - b * b) * id;
  d.minv_a[i] = make_float4(A * id, B * id, C * id, i11);
  d.minv_b[i] = make_float2(i12, i22);
}
__device__ __forceinline__ float3 minv_mul(const Dev &d, int i, float3 v) {
  const float4 ma = ld_once(d.minv_a + i);
  const float2 mb = ld_once(d.minv_b + i);
  const float4 r0 = make_float4(ma.x, ma.y, ma.z, 0), r1 = make_float4(ma.y, ma.w, mb.x, 0), r2 = make_float4(ma.z, mb.x, mb.y, 0);
  return f3(r0.x * v.x + r0.y * v.y + r0.z * v.z, r1.x * v.x + r1.y * v.y + r1.z * v.z, r2.x * v.x + r2.y * v.y + r2.z * v.z);
}
// ref: dfsph.py:217-223: r = v - A(v_guess); d = Minv r; delta_zero = sum r.d
__global__ void __launch_bounds__(BLOCK) k_visc_residual(Dev d, float dt) {
  LOAD_I
  float part = 0.f;
  if (own) {
    float3 ax = visc_Ax(d, i, pi, d.vel_guess, dt, ld_once(d.rho + i));
    float3 r = xyz(d.vel[i]) - ax;
    float3 dir = minv_mul(d, i, r);
    d.cg_r[i] = f4(r, 0);
    d.cg_dir[i] = f4(dir, 0);
    part = dot3(r, dir);
  }
  block_accumulate(ACC_SLOT(d, ACC_DZERO), part);
}
// fused dfsph.py:202-223: one neighbour walk builds the preconditioner block AND A(v_guess) (both read rho_j,
// the second also v_guess_j); used by the fused host path, the split kernels above stay for the per-phase API
__global__ void __launch_bounds__(BLOCK) k_visc_minv_residual(Dev d, float dt) {
  PRED_GATE(d);
  LOAD_I_PR
  float part = 0.f;
  if (own) {
    const float rhoi = RHO_I_PR;
    const float e = 0.01f * d.h2;
    const float cl = d.dim_coff * d.visc * d.mass;
    const float cs = d.dim_coff * d.visc_b * d.rho_S0 / rhoi * d.VS0;
    const float3 xi = xyz(d.vel_guess[i]);
    float m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
    float3 ret = f3(0, 0, 0);
    auto accum = [&](float3 r, float s) {
      m00 += s * r.x * r.x; m01 += s * r.x * r.y; m02 += s * r.x * r.z;
      m11 += s * r.y * r.y; m12 += s * r.y * r.z; m22 += s * r.z * r.z;
    };
    walk<8>(d, i, pi, d.pr, [&](int j, float4 p) { float4 xj = nb_ld(d.vel_guess, j); xj.w = p.w; return xj; },
            [&](int, float3 r, float r2, float4 xj) {
              const float c = cl * cubic_gradW_scale(d, r2) * frcp(xj.w * (r2 + e));
              accum(r, c);
              ret = ret + r * (c * dot3(xi - xyz(xj), r));
            },
            [&](int, float3 r, float r2) {
              const float c = cs * cubic_gradW_scale(d, r2) * frcp(r2 + e);
              accum(r, c);
              ret = ret + r * (c * dot3(xi, r));
            });
    const float f = dt / rhoi;
    float a = 1.0f - m00 * f, b = -m01 * f, c = -m02 * f, dd = 1.0f - m11 * f, ee = -m12 * f, ff = 1.0f - m22 * f;
    float A = dd * ff - ee * ee, B = -(b * ff - ee * c), C = b * ee - dd * c;
    float id = 1.0f / (a * A + b * B + c * C);
    float i11 = (a * ff - c * c) * id, i12 = -(a * ee - c * b) * id, i22 = (a * dd - b * b) * id;
    const float4 r0 = make_float4(A * id, B * id, C * id, 0), r1 = make_float4(B * id, i11, i12, 0), r2 = make_float4(C * id, i12, i22, 0);
    st_once(d.minv_a + i, make_float4(r0.x, r0.y, r0.z, r1.y)); st_once(d.minv_b + i, make_float2(r1.z, r2.z));
    const float3 res = xyz(ld_once(d.vel + i)) - (xi - ret * f);
    const float3 dir = f3(r0.x * res.x + r0.y * res.y + r0.z * res.z, r1.x * res.x + r1.y * res.y + r1.z * res.z,
                          r2.x * res.x + r2.y * res.y + r2.z * res.z);
    st_once(d.cg_r + i, f4(res, 0));
    d.cg_dir[i] = f4(dir, 0);
    part = dot3(res, dir);
  }
  block_accumulate(ACC_SLOT(d, ACC_DZERO), part);
}

// wcsph_dfsph_step only: k_visc_minv_residual + k_vorticity in one walk
__global__ void __launch_bounds__(BLOCK) k_visc_minv_residual_vort(Dev d, float dt) {
  PRED_GATE(d);
  LOAD_I
  float part = 0.f;
  if (own) {
    const float rhoi = d.rho[i];
    const float e = 0.01f * d.h2;
    const float cl = d.dim_coff * d.visc * d.mass;
    const float cs = d.dim_coff * d.visc_b * d.rho_S0 / rhoi * d.VS0;
    const float3 xi = xyz(d.vel_guess[i]);
    float m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
    float3 ret = f3(0, 0, 0);
    auto accum = [&](float3 r, float s) {
      m00 += s * r.x * r.x; m01 += s * r.x * r.y; m02 += s * r.x * r.z;
      m11 += s * r.y * r.y; m12 += s * r.y * r.z; m22 += s * r.z * r.z;
    };
    // compute_vorticity's sums (dfsph.py:307-330, see k_vorticity) ride on the same walk: they read vel_j, omega_j,
    // rho_j, none of which the viscosity solve changes
    const float3 wi = xyz(d.omega[i]), vi = xyz(d.vel[i]);
    float3 dom = f3(0, 0, 0), dv = f3(0, 0, 0);
    const float c1 = -1.0f / dt * d.vort_init * d.visc_omega * d.mass;
    const float c2 = d.vort_coff / rhoi * d.mass;
    const float c3 = d.vort_coff / rhoi * d.vort_init * d.mass;
    struct XVW { float4 x, v, w; };
    walk<4>(d, i, pi, d.pr, [&](int j, float4 p) { XVW q; q.x = nb_ld(d.vel_guess, j); q.x.w = p.w; q.v = nb_ld(d.vel, j); q.w = nb_ld(d.omega, j); return q; },
            [&](int, float3 r, float r2, XVW q) {
              const float g = cubic_gradW_scale(d, r2);
              const float irho = frcp(q.x.w);
              const float c = cl * g * irho * frcp(r2 + e);
              accum(r, c);
              ret = ret + r * (c * dot3(xi - xyz(q.x), r));
              const float3 dw = wi - xyz(q.w);
              dom = dom + dw * (c1 * irho * cubic_W(d, r2));
              dv = dv + cross3(dw, r) * (c2 * g);
              dom = dom + cross3(vi - xyz(q.v), r) * (c3 * g);
            },
            [&](int, float3 r, float r2) {
              const float g = cubic_gradW_scale(d, r2);
              const float c = cs * g * frcp(r2 + e);
              accum(r, c);
              ret = ret + r * (c * dot3(xi, r));
              dv = dv + cross3(wi, r) * (d.vort_coff / rhoi * d.rho_L0 * d.VS0 * g);
              dom = dom + cross3(vi, r) * (d.vort_coff / rhoi * d.vort_init * d.rho_L0 * d.VL0 * g);
            });
    dom = dom + wi * (-2.0f * d.vort_init * d.vort_coff * (float)d.ncount[i]);
    d.d_omega[i] = f4(dom, 0);
    { float4 a4 = d.d_vel[i]; a4.x += dv.x; a4.y += dv.y; a4.z += dv.z; d.d_vel[i] = a4; }
    const float f = dt / rhoi;
    float a = 1.0f - m00 * f, b = -m01 * f, c = -m02 * f, dd = 1.0f - m11 * f, ee = -m12 * f, ff = 1.0f - m22 * f;
    float A = dd * ff - ee * ee, B = -(b * ff - ee * c), C = b * ee - dd * c;
    float id = 1.0f / (a * A + b * B + c * C);
    float i11 = (a * ff - c * c) * id, i12 = -(a * ee - c * b) * id, i22 = (a * dd - b * b) * id;
    const float4 r0 = make_float4(A * id, B * id, C * id, 0), r1 = make_float4(B * id, i11, i12, 0), r2 = make_float4(C * id, i12, i22, 0);
    d.minv_a[i] = make_float4(r0.x, r0.y, r0.z, r1.y); d.minv_b[i] = make_float2(r1.z, r2.z);
    const float3 res = xyz(d.vel[i]) - (xi - ret * f);
    const float3 dir = f3(r0.x * res.x + r0.y * res.y + r0.z * res.z, r1.x * res.x + r1.y * res.y + r1.z * res.z,
                          r2.x * res.x + r2.y * res.y + r2.z * res.z);
    d.cg_r[i] = f4(res, 0);
    d.cg_dir[i] = f4(dir, 0);
    part = dot3(res, dir);
  }
  block_accumulate(ACC_SLOT(d, ACC_DZERO), part);
}

// ref: dfsph.py:227-230: Ad = A(dir); dAd = eps + sum dir.Ad
__global__ void __launch_bounds__(BLOCK) k_visc_Ad(Dev d, float dt) {
  PRED_GATE(d);
  acc_clear(ACC_SLOT(d, ACC_DNEW));    // k_visc_update (next) sums into it; its last reader, k_visc_dir, has finished
  LOAD_I_PR
  float part = 0.f;
  if (own) {
    float3 ad = visc_Ax(d, i, pi, d.cg_dir, dt, RHO_I_PR);
    st_once(d.cg_Ad + i, f4(ad, 0));
    part = dot3(xyz(d.cg_dir[i]), ad);
  }
  block_accumulate(ACC_SLOT(d, ACC_DAD), part);
}
// ref: dfsph.py:233-240: alpha = delta / dAd; v_guess += alpha d; r -= alpha Ad; s = Minv r; delta' = sum r.s
// from_dzero (first iteration of the fused host loop): delta = delta_zero is still on the device (ACC_DZERO), the host
// has not waited for it
__global__ void __launch_bounds__(BLOCK) k_visc_update(Dev d, float delta, int from_dzero = 0) {
  PRED_GATE(d);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float part = 0.f;
  const double dAd = acc_total(ACC_SLOT(d, ACC_DAD));
  if (from_dzero) delta = (float)acc_total(ACC_SLOT(d, ACC_DZERO));
  if (i < d.nl && owned(d, i)) {
    const float alpha = delta / (float)((double)d.eps + dAd);
    float3 dir = xyz(d.cg_dir[i]);
    st_once(d.vel_guess + i, f4(xyz(ld_once(d.vel_guess + i)) + dir * alpha, 0));
    float3 r = xyz(ld_once(d.cg_r + i)) - xyz(ld_once(d.cg_Ad + i)) * alpha;
    float3 s = minv_mul(d, i, r);
    st_once(d.cg_r + i, f4(r, 0));
    st_once(d.cg_s + i, f4(s, 0));
    part = dot3(r, s);
  }
  block_accumulate(ACC_SLOT(d, ACC_DNEW), part);
}
// ref: dfsph.py:243-246: beta = delta' / delta_old; d = s + beta d
__global__ void k_visc_dir(Dev d, float delta_old, int from_dzero = 0) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double dnew = acc_total(ACC_SLOT(d, ACC_DNEW));
  if (from_dzero) delta_old = (float)acc_total(ACC_SLOT(d, ACC_DZERO));
  acc_clear(ACC_SLOT(d, ACC_DAD));     // read by k_visc_update (finished); the next k_visc_Ad sums into it
  if (i >= d.nl || !owned(d, i)) return;
  const float beta = (float)dnew / delta_old;
  d.cg_dir[i] = f4(xyz(ld_once(d.cg_s + i)) + xyz(d.cg_dir[i]) * beta, 0);
}
// ref: dfsph.py:339-343 end_viscosity
__global__ void k_visc_end(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float3 g = xyz(d.vel_guess[i]), v = xyz(d.vel[i]);
  float3 a = xyz(d.d_vel[i]) + (g - v) * (1.0f / dt);
  d.d_vel[i] = f4(a, 0);
  d.vel_guess[i] = f4(g - v, 0);
}

// ---- micropolar vorticity (ref: dfsph.py:307-330) -------------------------------------------
// Q18: solid j => omega_j = vel_j = 0 (the reference indexes out of bounds there).
// Q10: the damping term is added once per raw list entry => * neighborCount.
__global__ void __launch_bounds__(BLOCK) k_vorticity(Dev d, float dt) {
  LOAD_I
  if (!own) return;
  const float3 wi = xyz(d.omega[i]), vi = xyz(d.vel[i]);
  const float rhoi = d.rho[i];
  float3 dom = f3(0, 0, 0), dv = f3(0, 0, 0);
  const float c1 = -1.0f / dt * d.vort_init * d.visc_omega * d.mass;
  const float c2 = d.vort_coff / rhoi * d.mass;
  const float c3 = d.vort_coff / rhoi * d.vort_init * d.mass;
  struct WV { float4 w, v; };
  walk<4>(d, i, pi, d.pr, [&](int j, float4 q) { WV p; p.w = nb_ld(d.omega, j); p.w.w = q.w; p.v = nb_ld(d.vel, j); return p; },
       [&](int, float3 r, float r2, WV pj) {
         float3 dw = wi - xyz(pj.w);
         float g = cubic_gradW_scale(d, r2);
         dom = dom + dw * (c1 * frcp(pj.w.w) * cubic_W(d, r2));
         dv = dv + cross3(dw, r) * (c2 * g);
         dom = dom + cross3(vi - xyz(pj.v), r) * (c3 * g);
       },
       [&](int, float3 r, float r2) {
         float g = cubic_gradW_scale(d, r2);
         dv = dv + cross3(wi, r) * (d.vort_coff / rhoi * d.rho_L0 * d.VS0 * g);
         dom = dom + cross3(vi, r) * (d.vort_coff / rhoi * d.vort_init * d.rho_L0 * d.VL0 * g);
       });
  dom = dom + wi * (-2.0f * d.vort_init * d.vort_coff * (float)d.ncount[i]);
  d.d_omega[i] = f4(dom, 0);
  float4 a = d.d_vel[i];
  a.x += dv.x; a.y += dv.y; a.z += dv.z;
  d.d_vel[i] = a;
}
// second loop of compute_vorticity (ref: dfsph.py:329-330); separate launch: omega_j is read above
__global__ void k_vorticity_integrate(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float4 w = d.omega[i], dw = d.d_omega[i];
  d.omega[i] = make_float4(w.x + dw.x * dt, w.y + dw.y * dt, w.z + dw.z * dt, 0);
}

// ---- CFL (ref: dfsph.py:555-568 + :107-111) -------------------------------------------------
// The reference's strided tree leaves max over original indices [0, 2^floor(log2(N-1))) in slot 0 (Q5).
__global__ void __launch_bounds__(BLOCK) k_cfl(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float vm = 0.f;
  if (i < d.nl) {
    float3 v = xyz(d.vel[i]) + xyz(d.d_vel[i]) * dt;
    float m = fmaxf(dot3(v, v), 0.1f);
    d.vel_max[i] = m;
    int o = d.oid[i];
    if (o < d.n_owned && (d.gid ? d.gid[o] : o) < d.cfl_prefix) vm = m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_down(vm, o, WAVE));
  __shared__ float sh[BLOCK / WAVE];
  if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = vm;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < BLOCK / WAVE; k++) vm = fmaxf(vm, sh[k]);
    // positive floats order like their bit patterns; 64 striped targets (flags[4..67]) instead of one: same-address
    // device-scope atomics are resolved one at a time at the memory side (see ACC_STRIPES)
    if (vm > 0.f) atomicMax(&d.flags[4 + (blockIdx.x & (ACC_STRIPES - 1))], __float_as_int(vm));
  }
}
// one wave: max over the stripes -> flags[FLAG_VMAX] and slot 0 of the reference's vel_max field; stripes cleared
// the host rule of optimize_time_step (dfsph.py:113-129; wcsph_dfsph_optimize_time_step) for the device: the same IEEE
// double expressions in the same order.  it = max(vs_iter, pr_iter) as the host knows it when it queues the launch.
struct DtRule { float *out; float dt_old; int it; double eps, cfl_radius, user_max_t, user_min_t; };
__device__ __forceinline__ float dt_rule(const DtRule &r, float vmax0) {
  float dt = r.dt_old;
  if ((double)vmax0 > r.eps) {
    double time_step = 0.5 * 0.4 * r.cfl_radius * 2.0 / sqrt((double)vmax0);
    if (time_step > r.user_max_t) time_step = r.user_max_t;
    if (time_step < r.user_min_t) time_step = r.user_min_t;
    if (r.it > 10) dt = (float)((double)dt * 0.9);
    else if (r.it < 5) dt = (float)((double)dt * 1.1);
    if ((double)dt > time_step) dt = (float)time_step;
  }
  return dt;
}
__device__ __forceinline__ void cfl_store_body(const Dev &d, const DtRule &r) {   // one wave
  int v = d.flags[4 + threadIdx.x];
  d.flags[4 + threadIdx.x] = 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, WAVE));
  if (threadIdx.x == 0) {
    d.flags[FLAG_VMAX] = v;
    if (d.nl > 0 && d.n_owned > 0) d.vel_max[d.where[0]] = __int_as_float(v);
    if (r.out) { const float dt = dt_rule(r, __int_as_float(v)); *r.out = dt; d.flags[FLAG_DT] = __float_as_int(dt); }
  }
}
__global__ void k_cfl_store(Dev d, DtRule r) {
  PRED_GATE(d);
  cfl_store_body(d, r);
}

// wcsph_dfsph_step only: end_viscosity (dfsph.py:339-343) + the integration loop of compute_vorticity (:329-330) + the
// per-particle part of the CFL reduction (:555-568) in one elementwise launch -- all three touch only particle i and
// nothing reads their outputs from a neighbour before the next launch
__global__ void __launch_bounds__(BLOCK) k_finish_nonpressure(Dev d, float dt) {
  PRED_GATE(d);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float vm = 0.f;
  if (i < d.nl) {
    const int o = ld_once(d.oid + i);
    const float3 v = xyz(ld_once(d.vel + i));
    float3 a = xyz(ld_once(d.d_vel + i));
    if (o < d.n_owned) {
      const float3 g = xyz(ld_once(d.vel_guess + i));
      a = a + (g - v) * (1.0f / dt);
      st_once(d.d_vel + i, f4(a, 0));
      st_once(d.vel_guess + i, f4(g - v, 0));
      const float4 w = ld_once(d.omega + i), dw = ld_once(d.d_omega + i);
      st_once(d.omega + i, make_float4(w.x + dw.x * dt, w.y + dw.y * dt, w.z + dw.z * dt, 0));
    }
    const float3 vn = v + a * dt;
    const float m = fmaxf(dot3(vn, vn), 0.1f);
    st_once(d.vel_max + i, m);
    if (o < d.n_owned && (d.gid ? d.gid[o] : o) < d.cfl_prefix) vm = m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_down(vm, o, WAVE));
  __shared__ float sh[BLOCK / WAVE];
  if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = vm;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < BLOCK / WAVE; k++) vm = fmaxf(vm, sh[k]);
    if (vm > 0.f) atomicMax(&d.flags[4 + (blockIdx.x & (ACC_STRIPES - 1))], __float_as_int(vm));
  }
}

// ref: dfsph.py:572-580
__global__ void k_update_vel(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float4 v = d.vel[i], a = d.d_vel[i];
  d.vel[i] = make_float4(v.x + a.x * dt, v.y + a.y * dt, v.z + a.z * dt, 0);
}
__global__ void k_update_pos(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float4 p = d.pos[i], v = d.vel[i];
  d.pos[i] = make_float4(p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt, p.w);
}

// wcsph_dfsph_step only -- elementwise neighbours in the launch sequence, fused (each touches particle i alone and nothing
// in between reads their outputs):
// end of solve_divergence (dfsph.py:480-484) + clear_nonpressure (:333-336) + the v_guess update of init_viscosity_para (:199-200)
// vort != 0: k_drho_div_iter_vort parked the vorticity acceleration in cg_s
__global__ void k_end_div_begin_nonpressure(Dev d, float dt, int vort) {
  PRED_GATE(d);
  acc_clear(ACC_SLOT(d, ACC_DZERO));   // as k_visc_guess
  acc_clear(ACC_SLOT(d, ACC_DAD));
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  if (owned(d, i)) { st_once(d.kappa_v + i, ld_once(d.kappa_v + i) * dt); st_once(d.alpha + i, ld_once(d.alpha + i) * dt); }
  float4 a = make_float4(d.gx, d.gy, d.gz, 0);
  if (vort) { const float4 dv = ld_once(d.cg_s + i); a.x += dv.x; a.y += dv.y; a.z += dv.z; }
  st_once(d.d_vel + i, a);
  float4 g = ld_once(d.vel_guess + i), v = d.vel[i];
  d.vel_guess[i] = make_float4(g.x + v.x, g.y + v.y, g.z + v.z, 0);   // gathered by the viscosity walk launched next
}
// update_vel (dfsph.py:572-575) + the kappa rescale that opens warmstart_pressure (:490)
// gen != 0: also raises flags[FLAG_WARM] = gen where the warm start's gate opens (k_warm_press_vel_flagged)
__global__ void k_update_vel_warm_scale(Dev d, float dt, int gen) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float4 v = ld_once(d.vel + i), a = ld_once(d.d_vel + i);
  d.vel[i] = make_float4(v.x + a.x * dt, v.y + a.y * dt, v.z + a.z * dt, 0);   // gathered by the pressure solve
  st_once(d.kappa + i, fmaxf(ld_once(d.kappa + i) / dt / dt, -0.5f * d.rho_L0 * d.rho_L0));
  if (gen && d.adv_rho[i] > d.rho_L0) d.flags[FLAG_WARM] = gen;
}
// end of solve_pressure (dfsph.py:549-552) + update_pos (:577-580)
__global__ void k_end_press_update_pos(Dev d, float dt) {
  PRED_GATE(d);
  DT_OVERRIDE(d, dt);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  st_once(d.kappa + i, ld_once(d.kappa + i) * (dt * dt));
  float4 p = ld_once(d.pos + i), v = ld_once(d.vel + i);
  st_once(d.pos + i, make_float4(p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt, p.w));   // the sort reads it once
}

// ---- sesph.py (ref: sesph.py:130-196); Dev.h/ml_h3/mk_h3 hold the script's private kernel constants --
__global__ void k_ss_reset(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  d.vel[i] = make_float4(0, 0, 0, 0);
  d.pressure[i] = 0.f;
}
__global__ void __launch_bounds__(BLOCK) k_ss_density(Dev d) {
  LOAD_I
  if (!own) return;
  float sl = 0.f, ss = 0.f;
  walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); }, [&](int, float3, float r2, NoPayload) { sl += cubic_W(d, r2); },
          [&](int, float3, float r2) { ss += cubic_W(d, r2); });
  d.rho[i] = (d.ss_VL0 * (d.mk_h3 + sl) + d.ss_VS0 * ss) * d.ss_rho0;
}
__global__ void k_ss_pressure(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float rho = fmaxf(d.rho[i], d.ss_rho0);
  d.rho[i] = rho;
  float q = rho / d.ss_rho0, qq = q * q, qqqq = qq * qq;
  d.pressure[i] = d.ss_stiff * (qqqq * qq * q - 1.0f);
}
__global__ void __launch_bounds__(BLOCK) k_ss_force(Dev d) {
  LOAD_I
  if (!own) return;
  const float3 vi = xyz(d.vel[i]);
  const float rhoi = d.rho[i], prsi = d.pressure[i];
  const float e = 0.01f * d.h2;
  const float pi2 = prsi / (rhoi * rhoi);
  float3 a = f3(d.gx, d.gy, d.gz);
  struct VP { float4 v; float p; };
  walk<4>(d, i, pi, d.pos, [&](int j, float4) { VP q; q.v = nb_ld(d.vel, j); q.v.w = nb_ld(d.rho, j); q.p = nb_ld(d.pressure, j); return q; },
       [&](int, float3 r, float r2, VP q) {
         float g = cubic_gradW_scale(d, r2);
         float rhoj = q.v.w;
         float visc = d.ss_dim * d.ss_visc * d.ss_mass * dot3(vi - xyz(q.v), r) * frcp(rhoj * (r2 + e));
         float prs = -d.ss_rho0 * d.ss_VL0 * (pi2 + q.p * frcp(rhoj * rhoj));
         a = a + r * ((visc + prs) * g);
       },
       [&](int, float3 r, float r2) {
         float g = cubic_gradW_scale(d, r2);
         float visc = d.ss_dim * d.ss_visc_b * d.ss_VS0 * (rhoi / d.ss_rho0) * dot3(vi, r) * frcp(r2 + e);
         float prs = -d.ss_rho0 * d.ss_VS0 * (pi2 + prsi / (d.ss_rho0 * d.ss_rho0));
         a = a + r * ((visc + prs) * g);
       });
  d.d_vel[i] = f4(a, 0);
}
__global__ void k_ss_integrate(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float4 v = d.vel[i], a = d.d_vel[i], p = d.pos[i];
  v = make_float4(v.x + a.x * dt, v.y + a.y * dt, v.z + a.z * dt, 0);
  d.vel[i] = v;
  d.pos[i] = make_float4(p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt, p.w);
}
