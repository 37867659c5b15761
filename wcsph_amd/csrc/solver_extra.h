// solver_extra.h -- the @ti.kernel loops of pcisph.py and iisph.py.
// Both scripts keep private kernel constants (pi = 3.1415926) like sesph.py; the kernels below run on a
// Dev view whose h / ml_h3 / mk_h3 hold those constants (sesph_view()).
#pragma once
#include "device.h"

// ---------------------------------------------------------------------------------------------- pcisph.py
// ref: pcisph.py:193-197
__global__ void k_pc_reset(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nl) d.vel[i] = make_float4(0, 0, 0, 0);
}
// ref: pcisph.py:199-218, viscous part (rho comes from the density kernel launched just before: the reference
// reads rho[j] while the same loop is still writing it -- a data race; here rho_j is the finished value, Q19)
__global__ void __launch_bounds__(BLOCK) k_pc_force(Dev d, float visc, float visc_b) {
  LOAD_I
  if (!own) return;
  const float3 vi = xyz(d.vel[i]);
  const float rhoi = d.rho[i];
  const float e = 0.01f * d.h2;
  float3 a = f3(d.gx, d.gy, d.gz);
  walk<8>(d, i, pi, d.pos, [&](int j, float4) { float4 v = nb_ld(d.vel, j); v.w = nb_ld(d.rho, j); return v; },
          [&](int, float3 r, float r2, float4 vj) {
            a = a + r * (d.ss_dim * visc * d.ss_mass * dot3(vi - xyz(vj), r) * frcp(vj.w * (r2 + e)) * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) {
            a = a + r * (d.ss_dim * visc_b * d.ss_VS0 * (rhoi / d.ss_rho0) * dot3(vi, r) * frcp(r2 + e) * cubic_gradW_scale(d, r2));
          });
  d.d_vel[i] = f4(a, 0);
}
// ref: pcisph.py:220-226
__global__ void k_pc_init_iter(Dev d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  d.vel_star[i] = d.vel[i];
  d.pos_star[i] = d.pos[i];
  d.pressure[i] = 0.f;
  d.d_vel_pre[i] = make_float4(0, 0, 0, 0);
}
// ref: pcisph.py:228-235
__global__ void k_pc_update_iter(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl) return;
  float3 vs = xyz(d.vel[i]) + (xyz(d.d_vel[i]) + xyz(d.d_vel_pre[i])) * dt;
  float4 p = d.pos[i];
  d.vel_star[i] = f4(vs, 0);
  d.pos_star[i] = make_float4(p.x + vs.x * dt, p.y + vs.y * dt, p.z + vs.z * dt, p.w);
  d.pressure[i] = 0.f;
}
// ref: pcisph.py:237-256 (the prediction uses the CURRENT positions, Q13)
__global__ void __launch_bounds__(BLOCK) k_pc_predict(Dev d, float dt, float pci_coff) {
  LOAD_I
  float e = 0.f;
  if (own) {
    float sl = 0.f, ss = 0.f;
    walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); }, [&](int, float3, float r2, NoPayload) { sl += cubic_W(d, r2); },
            [&](int, float3, float r2) { ss += cubic_W(d, r2); });
    float a = fmaxf(d.ss_VL0 * (d.mk_h3 + sl) + d.ss_VS0 * ss, 1.0f);
    d.adv_rho[i] = a;
    d.pressure[i] += pci_coff * (a - 1.0f) / (dt * dt);
    e = a - 1.0f;
  }
  block_accumulate(ACC_SLOT(d, ACC_ERR), e);
}
// ref: pcisph.py:258-278: gradW(pos_i - pos_star_j) for liquid j, gradW(pos_i - pos_j) for solid j
__global__ void __launch_bounds__(BLOCK) k_pc_pressure_force(Dev d) {
  LOAD_I
  if (!own) return;
  const float prsi = d.pressure[i];
  float3 a = f3(0, 0, 0);
  walk<8>(d, i, pi, d.pos_star, [&](int j, float4) { return nb_ld(d.pressure, j); },
          [&](int, float3 r, float r2, float pj) { a = a + r * (-d.ss_VL0 * (prsi + pj) * cubic_gradW_scale(d, r2)); },
          [&](int, float3 r, float r2) { a = a + r * (-d.ss_VS0 * prsi * cubic_gradW_scale(d, r2)); });
  d.d_vel_pre[i] = f4(a, 0);
}
// ref: pcisph.py:281-285
__global__ void k_pc_update_pos(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float3 v = xyz(d.vel[i]) + (xyz(d.d_vel[i]) + xyz(d.d_vel_pre[i])) * dt;
  float4 p = d.pos[i];
  d.vel[i] = f4(v, 0);
  d.pos[i] = make_float4(p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt, p.w);
}

// ---------------------------------------------------------------------------------------------- iisph.py
// ref: iisph.py:270-274
__global__ void k_ii_combine(Dev d, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nl || !owned(d, i)) return;
  float3 g = xyz(d.vel_guess[i]), v = xyz(d.vel[i]);
  d.d_vel[i] = f4(f3(d.gx, d.gy, d.gz) + (g - v) * (1.0f / dt), 0);
  d.vel_guess[i] = f4(g - v, 0);
}
// ref: iisph.py:276-291: vel += dt d_vel ; d_ii = sum -VL0 (rho0/rho_i)^2 gradW over EVERY entry
__global__ void __launch_bounds__(BLOCK) k_ii_dii(Dev d, float dt) {
  LOAD_I
  if (!own) return;
  float4 v = d.vel[i], a = d.d_vel[i];
  d.vel[i] = make_float4(v.x + dt * a.x, v.y + dt * a.y, v.z + dt * a.z, 0);
  const float inv_den = d.ss_rho0 / d.rho[i];
  const float c = -d.ss_VL0 * inv_den * inv_den;
  float3 s = f3(0, 0, 0);
  walk<8>(d, i, pi, d.pos, [](int, float4) { return NoPayload(); },
          [&](int, float3 r, float r2, NoPayload) { s = s + r * (c * cubic_gradW_scale(d, r2)); },
          [&](int, float3 r, float r2) { s = s + r * (c * cubic_gradW_scale(d, r2)); });
  d.d_ii[i] = f4(s, 0);
}
// ref: iisph.py:293-316
__global__ void __launch_bounds__(BLOCK) k_ii_aii(Dev d, float dt) {
  LOAD_I
  if (!own) return;
  const float density = d.rho[i] / d.ss_rho0;
  const float3 vi = xyz(d.vel[i]), dii = xyz(d.d_ii[i]);
  const float cji = d.ss_VL0 / (density * density);
  float adv = density, aii = 0.f;
  d.pressure_pre[i] = 0.5f * d.pressure[i];
  walk<8>(d, i, pi, d.pos, [&](int j, float4) { return nb_ld(d.vel, j); },
          [&](int, float3 r, float r2, float4 vj) {
            const float g = cubic_gradW_scale(d, r2);
            adv += dt * d.ss_VL0 * g * dot3(vi - xyz(vj), r);
            aii += d.ss_VL0 * g * dot3(dii - r * (cji * g), r);
          },
          [&](int, float3 r, float r2) {
            const float g = cubic_gradW_scale(d, r2);
            adv += dt * d.ss_VS0 * g * dot3(vi, r);
            aii += d.ss_VL0 * g * dot3(dii - r * (cji * g), r);
          });
  d.a_ii[i] = aii;
  d.adv_rho[i] = adv;
}
// ref: iisph.py:318-334
__global__ void __launch_bounds__(BLOCK) k_ii_dijpj(Dev d) {
  LOAD_I
  if (!own) return;
  float3 s = f3(0, 0, 0);
  struct RP { float rho, pp; };
  walk<8>(d, i, pi, d.pos, [&](int j, float4) { RP q; q.rho = nb_ld(d.rho, j); q.pp = nb_ld(d.pressure_pre, j); return q; },
          [&](int, float3 r, float r2, RP q) {
            const float dj = q.rho / d.ss_rho0;
            s = s + r * (-d.ss_VL0 / (dj * dj) * q.pp * cubic_gradW_scale(d, r2));
          },
          [&](int, float3, float) {});
  d.dij_pj[i] = f4(s, 0);
}
// ref: iisph.py:336-370
__global__ void __launch_bounds__(BLOCK) k_ii_pressure(Dev d, float dt, float omega) {
  LOAD_I
  float e = 0.f;
  if (own) {
    const float density = d.rho[i] / d.ss_rho0, ppi = d.pressure_pre[i];
    const float cji = d.ss_VL0 / (density * density);
    const float3 dpi = xyz(d.dij_pj[i]);
    float sum = 0.f;
    struct DP { float4 dij, dii; };   // dij.w = pressure_pre_j
    walk<4>(d, i, pi, d.pos, [&](int j, float4) { DP q; q.dij = nb_ld(d.dij_pj, j); q.dij.w = nb_ld(d.pressure_pre, j); q.dii = nb_ld(d.d_ii, j); return q; },
            [&](int, float3 r, float r2, DP q) {
              const float g = cubic_gradW_scale(d, r2);
              const float3 d_ji_pi = r * (cji * g * ppi);
              const float3 t = dpi - xyz(q.dii) * q.dij.w - (xyz(q.dij) - d_ji_pi);
              sum += d.ss_VL0 * g * dot3(t, r);
            },
            [&](int, float3 r, float r2) { sum += d.ss_VS0 * cubic_gradW_scale(d, r2) * dot3(dpi, r); });
    const float b = 1.0f - d.adv_rho[i], h2 = dt * dt, aii = d.a_ii[i], denom = aii * h2;
    float p = 0.f;
    if (fabsf(denom) > d.eps) p = fmaxf((1.0f - omega) * ppi + omega / denom * (b - h2 * sum), 0.0f);
    d.pressure[i] = p;
    if (p != 0.0f) e = (aii * p + sum) * h2 - b;
  }
  block_accumulate(ACC_SLOT(d, ACC_ERR), e);
}
// ref: iisph.py:372-391
__global__ void __launch_bounds__(BLOCK) k_ii_pressure_accel(Dev d) {
  LOAD_I
  if (!own) return;
  const float di = d.rho[i] / d.ss_rho0, dpi = d.pressure[i] / (di * di);
  float3 a = f3(0, 0, 0);
  struct RP { float rho, p; };
  walk<8>(d, i, pi, d.pos, [&](int j, float4) { RP q; q.rho = nb_ld(d.rho, j); q.p = nb_ld(d.pressure, j); return q; },
          [&](int, float3 r, float r2, RP q) {
            const float dj = q.rho / d.ss_rho0;
            a = a + r * (-d.ss_VL0 * (dpi + q.p / (dj * dj)) * cubic_gradW_scale(d, r2));
          },
          [&](int, float3 r, float r2) { a = a + r * (-d.ss_VS0 * dpi * cubic_gradW_scale(d, r2)); });
  d.d_vel[i] = f4(a, 0);
}
// ref: iisph.py:393-396 == sesph integrator (k_ss_integrate)
