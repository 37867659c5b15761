#!/usr/bin/env python3
"""Golden vectors from the reference's OWN text (build container only; nothing here travels to the GPU box).

    python tests/golden/make_reference_fixtures.py dfsph            ->  tests/golden/ref_dfsph_default_scene.npz
    python tests/golden/make_reference_fixtures.py dfsph_perturbed  ->  tests/golden/ref_dfsph_perturbed_default_scene.npz
    python tests/golden/make_reference_fixtures.py sesph            ->  tests/golden/ref_sesph_default_scene.npz
    python tests/golden/make_reference_fixtures.py iisph            ->  tests/golden/ref_iisph_default_scene.npz    (expanded + jittered state, see run_iisph)
    python tests/golden/make_reference_fixtures.py dfsph_squeezed   ->  tests/golden/ref_dfsph_squeezed_default_scene.npz   (round 4: 3 steps, Q5 / Q6 observable)
    python tests/golden/make_reference_fixtures.py dfsph_adhesion   ->  tests/golden/ref_dfsph_adhesion_default_scene.npz   (round 4: the solid branch of compute_tension)

`dfsph` starts from the script's own rest state, where most neighbour sums are identically zero for the first steps
(v = 0, omega = 0).  `dfsph_perturbed` is the same imported script with a non-trivial STATE written into its fields
before the loop body runs (positions jittered by +-0.2 R, a smooth velocity and spin field, tension_coff =
tension_coff_b = 0.05 -- the attributes the reference says to edit, ParticleData.py:80-81): every term of every pass
contributes, the solvers iterate.  The state is part of the fixture.

Puts tests/golden/taichi_standin (a minimal serial `taichi`: fields = NumPy arrays, f32 / i32 scalars, kernels run as
plain Python) first on sys.path, imports /root/reference/dfsph.py (or sesph.py) UNMODIFIED -- the script builds its
shipped default scene at import (8000 liquid particles + model/box_boundry.obj, resp. the 40^3 lattice shell), calls
reset_param() and, the stand-in's GUI not `running`, skips its main loop -- and then runs the main-loop body
(dfsph.py:606-617 / sesph.py:220-225) STEPS times through the script's own functions, recording every field the
parity tests compare, stage by stage.

The stand-in pins the reference's formulas, loop structure, constants and host control flow; it does not pin Taichi's
execution semantics (see taichi_standin/taichi.py).  tests/test_reference_fixtures.py compares the oracle (CPU) and the
HIP library (GPU) with these files.  Runtime: pure Python over ~2 M list entries per pass -- about an hour for dfsph.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "taichi_standin"))


def _name_fields(ti, *objs):
    for o in objs:
        for k, v in list(vars(o).items()):
            if isinstance(v, ti.Field):
                v.name = "%s.%s" % (type(o).__name__ if not isinstance(o, type(sys)) else o.__name__, k)


def perturb(ref, pd, out):
    """a deterministic non-trivial state, written through the fields' own from_numpy()"""
    rng = np.random.default_rng(20240901)
    nl = pd.liquid_count
    pos = pd.pos.to_numpy()
    pos[:nl] += rng.uniform(-0.2, 0.2, (nl, 3)).astype(np.float32) * np.float32(ref.particleRadius)
    pd.pos.from_numpy(pos)
    x, y, z = (pos[:nl, k].astype(np.float64) for k in range(3))
    vel = 0.5 * np.stack([np.sin(3.0 * y), np.cos(2.0 * x + z) - 0.3, np.sin(x - 2.0 * z)], 1)
    omega = 0.2 * np.stack([np.cos(4.0 * z), np.sin(3.0 * x), np.cos(2.0 * y + x)], 1)
    pd.vel.from_numpy(vel.astype(np.float32))
    pd.omega.from_numpy(omega.astype(np.float32))
    pd.tension_coff, pd.tension_coff_b = 0.05, 0.05
    out["vel0"], out["omega0"] = pd.vel.to_numpy(), pd.omega.to_numpy()
    out["tension_coff"], out["tension_coff_b"] = np.float32(0.05), np.float32(0.05)



def row_hashes(rows, counts):
    """all rows of HashGrid.neighbor as three numbers per row: sum, xor and an ORDER-SENSITIVE polynomial hash of the
    entries 0 .. count-1 (uint64 wrap-around: h = h * 1000003 + (j + 1)); a fixture of 8000 x 3 words instead of 8000 x 2048"""
    n = rows.shape[0]
    s = np.zeros(n, np.int64); x = np.zeros(n, np.int64); h = np.zeros(n, np.uint64)
    kmax = int(counts.max()) if n else 0
    with np.errstate(over="ignore"):
        for k in range(kmax):
            live = counts > k
            j = rows[:, k].astype(np.int64)
            s += np.where(live, j, 0)
            x ^= np.where(live, j, 0)
            h = np.where(live, h * np.uint64(1000003) + (j + 1).astype(np.uint64), h)
    return s, x, h


SQUEEZE_FIRST, SQUEEZE_FACTOR = 4400, 0.9


def squeeze_half(ref, pd, out):
    """perturb(), then the liquid particles with original index >= 4400 (x-planes 11..19 of the 20^3 block) compressed to 0.9
    of their spacing about their centroid.  Chosen with the oracle so that the reference's own quirks become OBSERVABLE in
    deltaT: the compressed half makes solve_pressure() iterate (pr_iter = 5 at step 0) while the particles the CFL tree
    actually covers -- original indices [0, 4096), Q5 -- stay calm, so the time-step cap does not bite and the dt rule
    shows which iteration count it read: step 1 keeps dt (it = max(vs_iter, pr_iter OF STEP 0) = 5, Q6), step 2 grows it
    by 1.1 (pr_iter of step 1 = 2).  A true velocity maximum would cap dt at ~0.0004 from step 1 on."""
    perturb(ref, pd, out)
    nl = pd.liquid_count
    pos = pd.pos.to_numpy()
    sel = np.arange(nl) >= SQUEEZE_FIRST
    c = pos[:nl][sel].mean(0, dtype=np.float64)
    pos[:nl][sel] = (c + SQUEEZE_FACTOR * (pos[:nl][sel].astype(np.float64) - c)).astype(np.float32)
    pd.pos.from_numpy(pos)
    out["squeeze_first"], out["squeeze_factor"] = np.int32(SQUEEZE_FIRST), np.float64(SQUEEZE_FACTOR)


ADH_N, ADH_R = 600, 0.2


def adhesion_ball(ref, pd, out):
    """perturb(), then the first 600 solid rows of the imported script moved onto a sphere of radius 0.2 around (0, 0.5, 0)
    -- inside the 0.26 gate of dfsph.py:300-302 and inside the liquid block, so that liquid particles sit at h/2 < r <= h
    from them and the adhesion term (kernels/AdhesionKernel.py:20-29) is non-zero.  No solid of model/box_boundry.obj lies
    inside that gate, which is why no other fixture exercises the branch."""
    perturb(ref, pd, out)
    nl = pd.liquid_count
    pos = pd.pos.to_numpy()
    k = np.arange(ADH_N) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / ADH_N); th = np.pi * (1.0 + 5.0 ** 0.5) * k        # Fibonacci sphere
    ball = np.stack([ADH_R * np.cos(th) * np.sin(phi), 0.5 + ADH_R * np.cos(phi), ADH_R * np.sin(th) * np.sin(phi)], 1)
    pos[nl:nl + ADH_N] = ball.astype(np.float32)
    pd.pos.from_numpy(pos)
    out["adh_n"], out["adh_r"] = np.int32(ADH_N), np.float64(ADH_R)


def run_dfsph(steps, perturbed=False, state=None):
    os.chdir(REF)                       # the script opens model/box_boundry.obj and MCData.txt by relative path
    import taichi as ti
    t0 = time.time()
    import dfsph as ref                 # builds the scene, reset_param()
    pd = ref.particle_data
    _name_fields(ti, ref, pd, pd.hash_grid)
    out = {}
    if state is not None:
        state(ref, pd, out)
    elif perturbed:
        perturb(ref, pd, out)
    out.update({"pos0": pd.pos.to_numpy(), "n_liquid": np.int32(pd.liquid_count), "n_solid": np.int32(pd.solid_count),
           "blockSize": pd.hash_grid.blockSize.to_numpy()[0], "maxb": pd.maxboundarynp.copy(), "minb": pd.minboundarynp.copy()})
    print("scene: %d liquid + %d solid, blockSize %s (%.0f s)" % (pd.liquid_count, pd.solid_count, out["blockSize"], time.time() - t0), flush=True)

    def snap(tag, **f):
        for k, v in f.items():
            out["s%d_%s_%s" % (step, tag, k)] = v.to_numpy() if hasattr(v, "to_numpy") else np.asarray(v)
        print("  step %d %-14s %.0f s" % (step, tag, time.time() - t0), flush=True)

    for step in range(steps):
        # ---- dfsph.py:606-617, in the script's own order ----
        pd.hash_grid.update_grid()
        snap("grid", neighborCount=pd.hash_grid.neighborCount, gridCount=pd.hash_grid.gridCount)
        if step == 0:
            rows0 = pd.hash_grid.neighbor.to_numpy()
            out["neighbor_rows_0_64"] = rows0[:64]
            if state is not None:        # round 4: every row of step 0, as three hashes per row
                out["rows_sum"], out["rows_xor"], out["rows_poly"] = row_hashes(rows0, pd.hash_grid.neighborCount.to_numpy())
        ref.compute_density()
        snap("density", rho=pd.rho)
        ref.compute_dfsph_coff()
        snap("coff", alpha=ref.alpha_coff)
        ref.solve_vel_divergence()
        snap("divergence", vel=pd.vel, adv_rho=pd.adv_rho, kappa_v=ref.kappa_v, alpha=ref.alpha_coff, dv_iter=np.int32(ref.dv_iter),
             avg_density_err=pd.avg_density_err)
        ref.compute_nonpressure_force()
        snap("nonpressure", d_vel=pd.d_vel, vel_guess=pd.vel_guess, omega=pd.omega, d_omega=pd.d_omega, normal=pd.normal,
             cg_Minv=pd.cg_Minv, vs_iter=np.int32(ref.vs_iter), cg_delta=pd.cg_delta, cg_delta_zero=pd.cg_delta_zero)
        ref.optimize_time_step()
        snap("dt", deltaT=ref.deltaT, vel_max0=pd.vel_max.to_numpy()[:1])
        ref.update_vel()
        ref.solve_pressure()
        snap("pressure", vel=pd.vel, adv_rho=pd.adv_rho, kappa=ref.kappa, pr_iter=np.int32(ref.pr_iter), avg_density_err=pd.avg_density_err)
        ref.update_pos()
        snap("end", pos=pd.pos)
    out["steps"] = np.int32(steps)
    out["oob_reads"] = np.array(["%s=%d" % kv for kv in sorted(ti.oob_reads.items())] or ["none"])
    print("out-of-bounds reads served as 0:", dict(ti.oob_reads), flush=True)
    return out


def run_adhesion():
    """grid, density, then clear_nonpressure + compute_tension through the script's own kernels (dfsph.py:264-305, 333-336):
    `normal` and `d_vel` with a live solid branch"""
    os.chdir(REF)
    import taichi as ti
    t0 = time.time()
    import dfsph as ref
    pd = ref.particle_data
    _name_fields(ti, ref, pd, pd.hash_grid)
    out = {}
    adhesion_ball(ref, pd, out)
    out.update({"pos0": pd.pos.to_numpy(), "n_liquid": np.int32(pd.liquid_count), "n_solid": np.int32(pd.solid_count),
                "blockSize": pd.hash_grid.blockSize.to_numpy()[0], "maxb": pd.maxboundarynp.copy(), "minb": pd.minboundarynp.copy()})
    pd.hash_grid.update_grid()
    out["neighborCount"], out["gridCount"] = pd.hash_grid.neighborCount.to_numpy(), pd.hash_grid.gridCount.to_numpy()
    out["rows_sum"], out["rows_xor"], out["rows_poly"] = row_hashes(pd.hash_grid.neighbor.to_numpy(), out["neighborCount"])
    print("  grid %.0f s" % (time.time() - t0), flush=True)
    ref.compute_density()
    out["rho"] = pd.rho.to_numpy()
    print("  density %.0f s" % (time.time() - t0), flush=True)
    ref.clear_nonpressure()
    ref.compute_tension()
    out["normal"], out["d_vel"] = pd.normal.to_numpy(), pd.d_vel.to_numpy()
    # the same two loops with the liquid coefficient off: what is left of d_vel - g is the solid branch alone
    pd.tension_coff = 0.0
    ref.clear_nonpressure()
    ref.compute_tension()
    out["d_vel_solid_only"] = pd.d_vel.to_numpy()
    print("  tension %.0f s" % (time.time() - t0), flush=True)
    out["oob_reads"] = np.array(["%s=%d" % kv for kv in sorted(ti.oob_reads.items())] or ["none"])
    return out


def run_sesph(steps):
    os.chdir(REF)
    import taichi as ti
    t0 = time.time()
    import sesph as ref
    pd = ref.particle_data
    _name_fields(ti, ref, pd, pd.hash_grid)
    out = {"pos0": pd.pos.to_numpy(), "n_liquid": np.int32(pd.liquid_count), "n_solid": np.int32(pd.solid_count),
           "blockSize": pd.hash_grid.blockSize.to_numpy()[0], "maxb": pd.maxboundarynp.copy(), "minb": pd.minboundarynp.copy()}
    print("scene: %d liquid + %d solid, blockSize %s (%.0f s)" % (pd.liquid_count, pd.solid_count, out["blockSize"], time.time() - t0), flush=True)
    for step in range(steps):
        # ---- sesph.py:220-225 ----
        pd.hash_grid.update_grid()
        out["s%d_neighborCount" % step] = pd.hash_grid.neighborCount.to_numpy()
        print("  step %d grid %.0f s" % (step, time.time() - t0), flush=True)
        ref.update_advection_density()
        out["s%d_rho_raw" % step] = ref.rho.to_numpy()
        ref.update_pressure()
        out["s%d_rho" % step], out["s%d_pressure" % step] = ref.rho.to_numpy(), ref.pressure.to_numpy()
        print("  step %d density %.0f s" % (step, time.time() - t0), flush=True)
        ref.compute_force()
        out["s%d_d_vel" % step] = ref.d_vel.to_numpy()
        ref.integrator_sesph()
        out["s%d_vel" % step], out["s%d_pos" % step] = ref.vel.to_numpy(), pd.pos.to_numpy()
        out["s%d_deltaT" % step] = ref.deltaT.to_numpy()
        print("  step %d end %.0f s" % (step, time.time() - t0), flush=True)
    out["steps"] = np.int32(steps)
    out["oob_reads"] = np.array(["%s=%d" % kv for kv in sorted(ti.oob_reads.items())] or ["none"])
    return out


def squeeze(ref, pd, out, factor=0.92, vel_field=None):
    """a compressed, jittered liquid block with a smooth velocity field, written through the fields' own from_numpy():
    rho > rho0, so the pressure loops of iisph.py / pcisph.py do real work (from their rest lattice rho = 800 < rho0 and
    every pressure stays 0)"""
    rng = np.random.default_rng(20240902)
    nl = pd.liquid_count
    pos = pd.pos.to_numpy()
    c = pos[:nl].mean(0, dtype=np.float64)
    pos[:nl] = (c + factor * (pos[:nl].astype(np.float64) - c) + rng.uniform(-0.1, 0.1, (nl, 3)) * ref.particleRadius).astype(np.float32)
    pd.pos.from_numpy(pos)
    x, y, z = (pos[:nl, k].astype(np.float64) for k in range(3))
    vel = 0.3 * np.stack([np.sin(3.0 * y), np.cos(2.0 * x + z) - 0.3, np.sin(x - 2.0 * z)], 1)
    (vel_field if vel_field is not None else ref.vel).from_numpy(vel.astype(np.float32))
    out["vel0"] = vel.astype(np.float32)
    out["squeeze"] = np.float64(factor)


IISPH_FACTOR = 1.06


def run_iisph(steps):
    os.chdir(REF)
    import taichi as ti
    t0 = time.time()
    import iisph as ref                 # builds the scene (8000 + model/box_boundry.obj), reset_param()
    pd = ref.particle_data
    _name_fields(ti, ref, pd, pd.hash_grid)
    out = {}
    # factor 1.06: the block slightly EXPANDED and jittered.  Any uniformly compressed state sends iisph.py's pressure loop to
    # its cap of 100 iterations (the script never refreshes pressure_pre) -- 100 x 4 minutes in the serial
    # stand-in; here 272 particles still carry positive pressure (the hash duplicates of faithful mode push their density
    # over rho0, Q1) and the loop ends after its forced two iterations (checked with the oracle before the hour was spent).
    squeeze(ref, pd, out, factor=IISPH_FACTOR)
    out.update({"pos0": pd.pos.to_numpy(), "n_liquid": np.int32(pd.liquid_count), "n_solid": np.int32(pd.solid_count),
                "blockSize": pd.hash_grid.blockSize.to_numpy()[0], "maxb": pd.maxboundarynp.copy(), "minb": pd.minboundarynp.copy()})
    print("scene: %d liquid + %d solid (%.0f s)" % (pd.liquid_count, pd.solid_count, time.time() - t0), flush=True)
    for step in range(steps):
        def snap(tag, **f):
            for k, v in f.items():
                out["s%d_%s_%s" % (step, tag, k)] = v.to_numpy() if hasattr(v, "to_numpy") else np.asarray(v)
            print("  step %d %-12s %.0f s" % (step, tag, time.time() - t0), flush=True)
        # ---- iisph.py:419-427 ----
        pd.hash_grid.update_grid()
        snap("grid", neighborCount=pd.hash_grid.neighborCount)
        ref.compute_density()
        snap("density", rho=ref.rho)
        ref.compute_nonpressure_force()
        snap("nonpressure", d_vel=ref.d_vel, vel_guess=ref.vel_guess, vs_iter=np.int32(ref.vs_iter), cg_delta=ref.cg_delta, cg_delta_zero=ref.cg_delta_zero)
        ref.compute_advection()
        snap("advection", d_ii=ref.d_ii, a_ii=ref.a_ii, adv_rho=ref.adv_rho, vel=ref.vel)
        ref.solve_pressure()
        snap("pressure", pressure=ref.pressure, dij_pj=ref.dij_pj, pr_iter=np.int32(ref.pr_iter), avg_density_err=ref.avg_density_err)
        ref.update_pos()
        snap("end", pos=pd.pos, vel=ref.vel, d_vel=ref.d_vel, deltaT=ref.deltaT)
    out["steps"] = np.int32(steps)
    out["oob_reads"] = np.array(["%s=%d" % kv for kv in sorted(ti.oob_reads.items())] or ["none"])
    return out


# pcisph.py cannot be pinned this way: its compute_nonpressure_force divides by rho[j] inside the loop that is still
# writing rho (pcisph.py:212-213, Q19) -- serially rho[j] is 0 for every j > i and the sums become inf / NaN, under Taichi
# it is a data race -- and update_iter_info stores to rho_err[i] for every i of a shape-(1) field (pcisph.py:234).  Tried
# in round 3; the library's pcisph stays checked against the oracle only (tests/test_gpu_pcisph_iisph.py).

if __name__ == "__main__":
    which = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else (3 if which == "dfsph_squeezed" else 2 if which.startswith("dfsph") else 1)
    if which == "dfsph_adhesion":
        res = run_adhesion()
    elif which == "dfsph_squeezed":
        res = run_dfsph(steps, state=squeeze_half)
    elif which.startswith("dfsph"):
        res = run_dfsph(steps, which == "dfsph_perturbed")
    elif which == "iisph":
        res = run_iisph(steps)
    else:
        res = run_sesph(steps)
    path = os.path.join(HERE, "ref_%s_default_scene.npz" % which)
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes")
