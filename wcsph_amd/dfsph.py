"""Headless drop-in for the reference's dfsph.py driver module (ref: dfsph.py:1-646).

Same module-level names as the reference script -- init_particle, reset_param,
compute_density, compute_dfsph_coff, solve_vel_divergence, compute_nonpressure_force,
optimize_time_step, update_vel, solve_pressure, update_pos and the finer kernels they
call; globals particle_data, deltaT, alpha_coff, kappa, kappa_v, vs_iter, dv_iter,
pr_iter -- so the reference main-loop body (dfsph.py:606-617) runs verbatim:

    particle_data.hash_grid.update_grid()
    compute_density(); compute_dfsph_coff(); solve_vel_divergence()
    compute_nonpressure_force(); optimize_time_step(); update_vel()
    solve_pressure(); update_pos()

Every kernel is a hand-written gfx950 kernel in libwcsph_hip.so; the host-side iteration
loops below are the reference's Python loops (same stopping rules, same scalar reads).
`step()` runs the same sequence through the library's fused host loop (no Python
round-trips).  No GUI / canvas / marching cubes (out of the hot path).
"""
import math

import numpy as np

from . import _lib, scene
from .ParticleData import ParticleData
from .field import Field, ScalarField
from .kernels.AdhesionKernel import AdhesionKernel
from .kernels.CohesionKernel import CohesionKernel
from .kernels.CubicKernel import CubicKernel

# ---- module constants (ref: dfsph.py:19-41) --------------------------------------------------
imgSizeX = 512                 # dfsph.py:19-20
imgSizeY = 512
current_time = 0.0
total_time = 5.0
eps = 1e-5
test_id = 0

particleRadius = 0.025
particleDimX = 20
particleDimY = 20
particleDimZ = 20
particleLiquidNum = particleDimX * particleDimY * particleDimZ

vs_iter = 0
dv_iter = 0
pr_iter = 0

user_max_t = 0.005
user_min_t = 0.0001

particle_data = None
kernel_c = kernel_adh = kernel_coh = None
deltaT = alpha_coff = kappa = kappa_v = None


def _call(name):
    ctx = particle_data._need_ctx()
    _lib.check(ctx, getattr(_lib.load(), name)(ctx))


def init_particle(filename=None, particleDim=None, neighbor_mode="faithful", boundary=None, maxNeighbour=2048, liquid=None, **attrs):
    """ref: dfsph.py:59-82.  filename: OBJ point cloud of boundary particles (`v x y z` lines) as in the
    reference; None generates the equivalent box shell (scene.box_shell).  particleDim scales the scene.
    liquid: explicit liquid positions instead of the particleDim^3 lattice (scene.dam_break_with_ball).
    attrs: ParticleData attributes set before the device setup (e.g. tension_coff=0.05; the reference edits them in
    ParticleData.py:61-87)."""
    global particle_data, kernel_c, kernel_adh, kernel_coh, deltaT, alpha_coff, kappa, kappa_v
    global particleDimX, particleDimY, particleDimZ, particleLiquidNum, vs_iter, dv_iter, pr_iter, current_time
    if particleDim is not None:
        if isinstance(particleDim, int):
            particleDim = (particleDim,) * 3
        particleDimX, particleDimY, particleDimZ = particleDim
    particleLiquidNum = particleDimX * particleDimY * particleDimZ
    vs_iter = dv_iter = pr_iter = 0
    current_time = 0.0

    particle_data = ParticleData(particleRadius, neighbor_mode=neighbor_mode, maxNeighbour=maxNeighbour)
    particle_data.user_max_t, particle_data.user_min_t = user_max_t, user_min_t
    for k, v in attrs.items():
        if not hasattr(particle_data, k):
            raise AttributeError("ParticleData has no attribute %r" % k)
        setattr(particle_data, k, v)
    if liquid is None:
        liquid = scene.dam_break_liquid(particleDimX, particleDimY, particleDimZ, particleRadius)
    particleLiquidNum = len(liquid)
    particle_data.add_liquid_points(liquid)

    kernel_c = CubicKernel(particle_data.hash_grid.searchR)
    kernel_adh = AdhesionKernel(particle_data.hash_grid.searchR)
    kernel_coh = CohesionKernel(particle_data.hash_grid.searchR)

    if filename is not None:
        particle_data.add_obj(filename)
    elif boundary is not None:
        particle_data.add_solid_points(boundary)
    else:
        L = max(particleDimX, particleDimY, particleDimZ) * particleRadius * 2.0
        particle_data.add_solid_points(scene.box_shell(L))
    particle_data.setup_data_gpu()
    particle_data.setup_data_cpu()

    deltaT = ScalarField(particle_data, "deltaT")
    alpha_coff = Field(particle_data, "alpha_coff", 1)
    kappa = Field(particle_data, "kappa", 1)
    kappa_v = Field(particle_data, "kappa_v", 1)
    return particle_data


# ---- kernels (one C-ABI call each) -----------------------------------------------------------------
def reset_param(): _call("wcsph_dfsph_reset_param")                    # noqa: E704  ref: dfsph.py:167-178
def compute_density(): _call("wcsph_dfsph_compute_density")            # noqa: E704  ref: dfsph.py:248-262
def compute_dfsph_coff(): _call("wcsph_dfsph_compute_coff")            # noqa: E704  ref: dfsph.py:345-372
def warmstart_divergence_vel(): _call("wcsph_dfsph_warmstart_divergence")  # noqa: E704  ref: dfsph.py:415-439
def begin_divergence_iter(): _call("wcsph_dfsph_begin_divergence")     # noqa: E704  ref: dfsph.py:441-446
def divergence_iter(): _call("wcsph_dfsph_divergence_iter")            # noqa: E704  ref: dfsph.py:449-477
def end_divergence_iter(): _call("wcsph_dfsph_end_divergence")         # noqa: E704  ref: dfsph.py:480-484
def warmstart_pressure(): _call("wcsph_dfsph_warmstart_pressure")      # noqa: E704  ref: dfsph.py:487-508
def begin_pressure_iter(): _call("wcsph_dfsph_begin_pressure")         # noqa: E704  ref: dfsph.py:511-516
def pressure_iter(): _call("wcsph_dfsph_pressure_iter")                # noqa: E704  ref: dfsph.py:518-547
def end_pressure_iter(): _call("wcsph_dfsph_end_pressure")             # noqa: E704  ref: dfsph.py:549-552
def clear_nonpressure(): _call("wcsph_dfsph_clear_nonpressure")        # noqa: E704  ref: dfsph.py:333-336
def compute_tension(): _call("wcsph_dfsph_tension")                    # noqa: E704  ref: dfsph.py:264-305
def init_viscosity_para(): _call("wcsph_dfsph_visc_init")              # noqa: E704  ref: dfsph.py:197-223
def compute_viscosity_force(): _call("wcsph_dfsph_visc_iter")          # noqa: E704  ref: dfsph.py:225-246
def end_viscosity(): _call("wcsph_dfsph_visc_end")                     # noqa: E704  ref: dfsph.py:339-343
def compute_vorticity(): _call("wcsph_dfsph_vorticity")                # noqa: E704  ref: dfsph.py:307-330
def update_vel(): _call("wcsph_dfsph_update_vel")                      # noqa: E704  ref: dfsph.py:572-575
def update_pos(): _call("wcsph_dfsph_update_pos")                      # noqa: E704  ref: dfsph.py:577-580


def cfl_time_step(index=None):
    """ref: dfsph.py:555-568.  The reference launches this ceil(log2 N) times with a growing stride; the
    library runs the whole tree as one max-reduction on the first call (index in (None, 1)) and the
    remaining strides are no-ops."""
    if index in (None, 1):
        _call("wcsph_dfsph_cfl_reduce")


# ---- host-side control: the reference's own Python loops -----------------------------------------------
def _push_iters():
    ctx = particle_data._need_ctx()
    _lib.load().wcsph_set_iters(ctx, vs_iter, dv_iter, pr_iter)


def compute_nonpressure_force():
    """ref: dfsph.py:84-103"""
    global vs_iter
    clear_nonpressure()
    compute_tension()
    init_viscosity_para()
    vs_iter = 0
    while vs_iter < 100:
        compute_viscosity_force()
        vs_iter += 1
        if particle_data.cg_delta[0] <= particle_data.viscosity_err * particle_data.cg_delta_zero[0] or \
                particle_data.cg_delta_zero[0] < eps:
            break
    end_viscosity()
    compute_vorticity()
    _push_iters()


def optimize_time_step():
    """ref: dfsph.py:107-129"""
    size = 1
    while size < particleLiquidNum:
        cfl_time_step(size)
        size = size * 2

    deltaT_np = deltaT.to_numpy()
    vel_max_np = particle_data.vel_max.to_numpy()

    if vel_max_np[0] > eps:
        cfl_factor = 0.5
        time_step = cfl_factor * 0.4 * particleRadius * 2.0 / math.sqrt(vel_max_np[0])
        time_step = min(time_step, user_max_t)
        time_step = max(time_step, user_min_t)

        iter = max(vs_iter, max(pr_iter, vs_iter))
        if iter > 10:
            deltaT_np[0] *= 0.9
        elif iter < 5:
            deltaT_np[0] *= 1.1

        deltaT_np[0] = min(deltaT_np[0], time_step)
        deltaT.from_numpy(deltaT_np)


def solve_vel_divergence():
    """ref: dfsph.py:131-146"""
    global dv_iter
    dv_iter = 0

    warmstart_divergence_vel()
    err = -0.1
    begin_divergence_iter()

    deltaT_np = deltaT.to_numpy()

    while (particle_data.avg_density_err.to_numpy()[0] > err) and (dv_iter < 10):
        divergence_iter()
        err = 0.001 * float(particleLiquidNum) / deltaT_np[0]
        dv_iter += 1

    end_divergence_iter()
    _push_iters()


def solve_pressure():
    """ref: dfsph.py:150-164"""
    global pr_iter

    warmstart_pressure()
    pr_iter = 0
    err = 0.0
    begin_pressure_iter()

    while (err > 0.001 or pr_iter < 2) and (pr_iter < 100):
        pressure_iter()
        err = particle_data.avg_density_err.to_numpy()[0] / float(particleLiquidNum)
        pr_iter += 1
    end_pressure_iter()
    _push_iters()


def _pull_iters():
    global vs_iter, dv_iter, pr_iter
    import ctypes as C
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.load().wcsph_get_iters(particle_data._need_ctx(), C.byref(a), C.byref(b), C.byref(c))
    vs_iter, dv_iter, pr_iter = a.value, b.value, c.value


_step_out = None


def step():
    """One pass of the reference main-loop body (dfsph.py:606-617) with the host loops inside the library.
    The GPU waits for nothing but this function between two steps (the library queues a whole step behind its own decisions):
    what happens here after the library call is on the critical path, so it is three ctypes calls on preallocated
    arguments and nothing else."""
    global current_time, vs_iter, dv_iter, pr_iter, _step_out
    import ctypes as C
    if _step_out is None:
        a, b, c, f = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        _step_out = (a, b, c, f, C.byref(a), C.byref(b), C.byref(c), C.byref(f), _lib.SCALAR_ID["deltaT"])
    a, b, c, f, ra, rb, rc, rf, dt_id = _step_out
    ctx, lib = particle_data._need_ctx(), _lib.load()
    _lib.check(ctx, lib.wcsph_dfsph_step(ctx))
    lib.wcsph_get_iters(ctx, ra, rb, rc)
    vs_iter, dv_iter, pr_iter = a.value, b.value, c.value
    _lib.check(ctx, lib.wcsph_get_scalar(ctx, dt_id, rf))
    dt = float(f.value)
    current_time += dt
    return dt


def step_fixed(dv=2, pr=4, vs=4):
    """The step with fixed solver iteration counts (equal-work micro-benchmark)."""
    global current_time
    ctx = particle_data._need_ctx()
    _lib.check(ctx, _lib.load().wcsph_dfsph_step_fixed(ctx, dv, pr, vs))
    _pull_iters()
    dt = float(deltaT.to_numpy()[0])
    current_time += dt
    return dt


def step_reference_loop():
    """The reference main-loop body, verbatim (dfsph.py:606-617), through the per-kernel entry points."""
    global current_time
    particle_data.hash_grid.update_grid()

    compute_density()
    compute_dfsph_coff()
    solve_vel_divergence()

    compute_nonpressure_force()
    optimize_time_step()
    update_vel()

    solve_pressure()
    update_pos()

    dt = deltaT.to_numpy()[0]
    current_time += dt
    return float(dt)


def draw_particle():
    """ref: dfsph.py:585-593 (liquid: white circles, every particle: a grey point)"""
    sph_canvas.draw_particle(particle_data)


sph_canvas = None


def main(steps=None, filename=None, particleDim=None, neighbor_mode="faithful", png_dir=None, surface_dir=None):
    """Headless version of the reference script body (dfsph.py:594-647).  png_dir: write the reference's frames
    (sph_canvas.export_png, :643, 30 fps of simulated time); surface_dir: write the marching-cubes meshes
    (particle_data.mc_grid.export_surface, :635, 20 fps)."""
    global sph_canvas
    from .Canvas import Canvas
    sph_canvas = Canvas(imgSizeX, imgSizeY)            # dfsph.py:596
    init_particle(filename, particleDim, neighbor_mode)
    reset_param()
    if png_dir is not None:
        sph_canvas.out_dir = png_dir
    if surface_dir is not None:
        particle_data.mc_grid.out_dir = surface_dir
    n = 0
    while steps is None or n < steps:
        sph_canvas.static_cam(0.0, 1.0, 0.0)           # dfsph.py:604
        dt = step()
        n += 1
        if png_dir is not None:
            sph_canvas.clear_canvas()                  # dfsph.py:621-622
            draw_particle()
            sph_canvas.export_png(current_time)        # dfsph.py:643
        if surface_dir is not None:
            particle_data.mc_grid.export_surface(current_time)   # dfsph.py:635
        print("time:%.3f" % current_time, "step:%.4f" % dt, "viscorcity:", vs_iter, "divergence:", dv_iter,
              "particle_data.pressure:", pr_iter)
        if math.isnan(particle_data.pos.to_numpy()[test_id, 0]) or current_time >= total_time:
            print(particle_data.adv_rho.to_numpy()[test_id], particle_data.pos.to_numpy()[test_id],
                  particle_data.d_vel.to_numpy()[test_id])
            break


if __name__ == "__main__":
    import sys
    main(steps=int(sys.argv[1]) if len(sys.argv) > 1 else None)
