"""Headless drop-in for the reference's iisph.py driver module (ref: iisph.py:1-445): same function names
(init_particle, reset_param, compute_density, init_viscosity_para, compute_viscosity_force, combine_nonpressure,
compute_nonpressure_force, compute_advection, update_iter_info, update_pressure_force, solve_pressure, update_pos)
and globals (particle_data, vel, vel_guess, d_vel, a_ii, d_ii, dij_pj, pressure_pre, pressure, rho, adv_rho,
deltaT, avg_density_err, cg_delta, cg_delta_zero, vs_iter, pr_iter); the main-loop body (iisph.py:419-428)
runs verbatim.  Reference behaviour kept: pressure_pre is only set in compute_advection, so every
solve_pressure() iteration computes the same relaxed-Jacobi update; fixed dt = 0.001 (no CFL call)."""
from . import _lib, scene
from .ParticleData import ParticleData
from .field import Field, ScalarField

current_time = 0.0
total_time = 5.0
test_id = 0
eps = 1e-5
particleRadius = 0.025
gridR = particleRadius * 2.0
invGridR = 1.0 / gridR
particleDimX = particleDimY = particleDimZ = 20
particleLiquidNum = particleDimX * particleDimY * particleDimZ
rho_L0 = 1000.0
rho_S0 = rho_L0
VL0 = particleRadius * particleRadius * particleRadius * 0.8 * 8.0
VS0 = VL0
liqiudMass = VL0 * rho_L0
boundary = 2.0
searchR = gridR * 2.0
dim_coff = 10.0
omega = 0.5
viscosity = 2.0
viscosity_b = 3.0
viscosity_err = 0.05
vs_iter = dv_iter = pr_iter = 0
user_max_t = 0.005
user_min_t = 0.00005

particle_data = None
vel = vel_guess = d_vel = a_ii = d_ii = dij_pj = pressure_pre = pressure = rho = adv_rho = None
deltaT = avg_density_err = cg_delta = cg_delta_old = cg_delta_zero = None


def _call(name):
    ctx = particle_data._need_ctx()
    _lib.check(ctx, getattr(_lib.load(), name)(ctx))


def liquid_block(nx=None, ny=None, nz=None):
    """ref: iisph.py:105-108"""
    import numpy as np
    nx, ny, nz = nx or particleDimX, ny or particleDimY, nz or particleDimZ
    i = np.arange(nx * ny * nz, dtype=np.int64)
    ZxY = nz * ny
    dis = particleRadius * 2.0
    return np.stack([(i // ZxY).astype(np.float64) * dis - particleRadius,
                     ((i % ZxY) // nz).astype(np.float64) * dis + 0.1,
                     (i % nz).astype(np.float64) * dis - particleRadius], 1)


def init_particle(filename=None, neighbor_mode="faithful", liquid=None, solid=None):
    """ref: iisph.py:98-112.  filename: OBJ point cloud of boundary particles (the reference passes
    model/box_boundry.obj); None generates the equivalent shell of the box [-1,1] x [0,2] x [-1,1]."""
    global particle_data, vel, vel_guess, d_vel, a_ii, d_ii, dij_pj, pressure_pre, pressure, rho, adv_rho
    global deltaT, avg_density_err, cg_delta, cg_delta_old, cg_delta_zero, current_time, vs_iter, pr_iter
    current_time, vs_iter, pr_iter = 0.0, 0, 0
    particle_data = ParticleData(gridR, neighbor_mode=neighbor_mode)   # the DIAMETER (Q11)
    pd = particle_data
    pd.sesph_particle_radius = particleRadius
    pd.list_radius = searchR
    pd.ss_VS0_factor = 1.0                                            # iisph.py:37 VS0 = VL0
    pd.ii_visc, pd.ii_visc_b, pd.ii_omega, pd.viscosity_err = viscosity, viscosity_b, omega, viscosity_err
    pd.add_liquid_points(liquid_block() if liquid is None else liquid)
    if solid is not None:
        pd.add_solid_points(solid)
    elif filename is not None:
        pd.add_obj(filename)
    else:
        pd.add_solid_points(scene.box_shell(1.0))
    pd.setup_data_gpu()
    pd.setup_data_cpu()
    vel, vel_guess, d_vel, pressure, rho, adv_rho = pd.vel, pd.vel_guess, pd.d_vel, pd.pressure, pd.rho, pd.adv_rho
    a_ii, pressure_pre = Field(pd, "a_ii", 1), Field(pd, "pressure_pre", 1)
    d_ii, dij_pj = Field(pd, "d_ii", 3), Field(pd, "dij_pj", 3)
    deltaT = ScalarField(pd, "deltaT")
    avg_density_err, cg_delta, cg_delta_old, cg_delta_zero = pd.avg_density_err, pd.cg_delta, pd.cg_delta_old, pd.cg_delta_zero
    return pd


def reset_param(): _call("wcsph_iisph_reset_param")                        # noqa: E704  ref: iisph.py:178-182
def compute_density(): _call("wcsph_iisph_compute_density")                # noqa: E704  ref: iisph.py:254-268
def init_viscosity_para(): _call("wcsph_iisph_visc_init")                  # noqa: E704  ref: iisph.py:200-229
def compute_viscosity_force(): _call("wcsph_iisph_visc_iter")              # noqa: E704  ref: iisph.py:231-252
def combine_nonpressure(): _call("wcsph_iisph_combine_nonpressure")        # noqa: E704  ref: iisph.py:270-274
def compute_advection(): _call("wcsph_iisph_compute_advection")            # noqa: E704  ref: iisph.py:276-316
def update_iter_info(): _call("wcsph_iisph_update_iter_info")              # noqa: E704  ref: iisph.py:318-334
def update_pressure_force(): _call("wcsph_iisph_update_pressure_force")    # noqa: E704  ref: iisph.py:336-370
def update_pos(): _call("wcsph_iisph_update_pos")                          # noqa: E704  ref: iisph.py:372-396


def compute_nonpressure_force():
    """ref: iisph.py:114-127"""
    global vs_iter
    init_viscosity_para()
    vs_iter = 0
    while vs_iter < 100:
        compute_viscosity_force()
        vs_iter += 1
        if cg_delta[0] <= viscosity_err * cg_delta_zero[0] or cg_delta_zero[0] < eps:
            break
    combine_nonpressure()


def solve_pressure():
    """ref: iisph.py:130-139"""
    global pr_iter
    pr_iter = 0
    err = 0.0
    while (err > 0.001 or pr_iter < 2) and (pr_iter < 100):
        update_iter_info()
        update_pressure_force()
        err = avg_density_err.to_numpy()[0] / float(particle_data.liquid_count)
        pr_iter += 1


def step():
    """ref: iisph.py:415-436 main-loop body (headless)"""
    global current_time
    particle_data.hash_grid.update_grid()
    compute_density()
    compute_nonpressure_force()
    compute_advection()
    solve_pressure()
    update_pos()
    dt = deltaT.to_numpy()[0]
    current_time += dt
    return float(dt)
