"""Headless drop-in for the reference's pcisph.py driver module (ref: pcisph.py:1-330): same function names
(init_particle, reset_param, compute_nonpressure_force, init_iter_info, update_iter_info, predict_density,
sovel_pressure, update_pos, GetPciCoff) and globals (particle_data, vel, d_vel, d_vel_pre, pos_star, vel_star,
rho, adv_rho, pressure, rho_err, deltaT, pr_iter, pci_coff); the main-loop body (pcisph.py:306-310) runs verbatim.

Reference behaviour kept: the density prediction uses the current positions (Q13), so every solver iteration
produces the same pressure; the particle DIAMETER goes to ParticleData (Q11).  One deviation: the reference reads
rho[j] while the same loop is still writing rho (a data race); here rho_j is this step's finished density (Q19).
"""
import numpy as np

from . import _lib, scene
from .ParticleData import ParticleData
from .field import Field, ScalarField

current_time = 0.0
total_time = 5.0
eps = 1e-5
test_id = 1000
particleRadius = 0.025
gridR = particleRadius * 2.0
searchR = gridR * 2.0
invGridR = 1.0 / gridR
boundary = 2.0
particleDimX = particleDimY = particleDimZ = 20
particleLiquidNum = particleDimX * particleDimY * particleDimZ
rho_0 = 1000.0
VL0 = particleRadius * particleRadius * particleRadius * 0.8 * 8.0
VS0 = VL0 * 2.0
liqiudMass = VL0 * rho_0
pi = 3.1415926
h3 = searchR * searchR * searchR
m_k = 8.0 / (pi * h3)
m_l = 48.0 / (pi * h3)
dim_coff = 10.0
viscosity = 0.05
viscosity_b = 0.0
pr_iter = 0

particle_data = None
pci_coff = None
vel = d_vel = d_vel_pre = pos_star = vel_star = rho = adv_rho = pressure = rho_err = deltaT = None


def GetPciCoff():
    """The PCISPH stiffness constant delta of a filled neighbourhood (ref: pcisph.py:87-115; host side, float64):
    1 / (2 V0^2 (|sum_j gradW_j|^2 + sum_j |gradW_j|^2)) over the lattice points of pitch 2R inside the support.
    One array expression over the whole lattice; the lattice coordinates are built by repeated addition from -h, as
    the reference's running xj is, so every coordinate has the reference's float64 rounding."""
    h, diam = searchR, 2.0 * particleRadius
    axis = [-h]
    while axis[-1] + diam <= h:
        axis.append(axis[-1] + diam)
    r = -np.stack(np.meshgrid(axis, axis, axis, indexing="ij"), -1).reshape(-1, 3)     # x_i - x_j with x_i = 0
    rl = np.sqrt((r * r).sum(1))
    q = rl / h
    ok = (rl < h) & (rl > 1.0e-5)
    with np.errstate(invalid="ignore", divide="ignore"):
        scale = np.where(q <= 0.5, m_l * q * (3.0 * q - 2.0), -m_l * (1.0 - q) ** 2) / (rl * h)
    grad = np.where(ok[:, None], scale[:, None] * r, 0.0)
    sum_grad = grad.sum(0)
    sum_grad2 = (grad * grad).sum()
    return float(1.0 / (2.0 * VL0 * VL0 * (sum_grad @ sum_grad + sum_grad2)))


def _call(name):
    ctx = particle_data._need_ctx()
    _lib.check(ctx, getattr(_lib.load(), name)(ctx))


def init_particle(filename=None, neighbor_mode="faithful", liquid=None, solid=None):
    """ref: pcisph.py:117-143 (the sesph.py scene; `filename` is unused by the reference as well)"""
    global particle_data, pci_coff, vel, d_vel, d_vel_pre, pos_star, vel_star, rho, adv_rho, pressure, rho_err, deltaT
    global current_time, pr_iter
    current_time, pr_iter = 0.0, 0
    pci_coff = GetPciCoff()
    particle_data = ParticleData(gridR, neighbor_mode=neighbor_mode)   # the DIAMETER, as the reference does (Q11)
    particle_data.sesph_particle_radius = particleRadius
    # the pressure force evaluates gradW at PREDICTED neighbour positions (pcisph.py:266-270): any particle of the
    # walked cells may enter the support, so keep the reference's raw list (unbounded radius, 2048 entries)
    particle_data.list_radius = -1.0
    particle_data.list_cap = particle_data.hash_grid.maxNeighbour
    particle_data.pci_coff = pci_coff
    particle_data.pc_visc, particle_data.pc_visc_b = viscosity, viscosity_b
    if liquid is None:
        liquid, solid = scene.sesph_scene(particleRadius, particleDimX, boundary)
    particle_data.add_liquid_points(liquid)
    particle_data.add_solid_points(solid)
    particle_data.setup_data_gpu()
    particle_data.setup_data_cpu()
    pd = particle_data
    vel, d_vel, rho, adv_rho, pressure = pd.vel, pd.d_vel, pd.rho, pd.adv_rho, pd.pressure
    d_vel_pre, pos_star, vel_star = Field(pd, "d_vel_pre", 3), Field(pd, "pos_star", 3), Field(pd, "vel_star", 3)
    rho_err, deltaT = ScalarField(pd, "rho_err"), ScalarField(pd, "deltaT")
    return particle_data


def reset_param(): _call("wcsph_pcisph_reset_param")                                # noqa: E704  ref: pcisph.py:193-197
def compute_nonpressure_force(): _call("wcsph_pcisph_compute_nonpressure_force")    # noqa: E704  ref: pcisph.py:199-218
def init_iter_info(): _call("wcsph_pcisph_init_iter_info")                          # noqa: E704  ref: pcisph.py:220-226
def update_iter_info(): _call("wcsph_pcisph_update_iter_info")                      # noqa: E704  ref: pcisph.py:228-235
def predict_density(): _call("wcsph_pcisph_predict_density")                        # noqa: E704  ref: pcisph.py:237-278
def update_pos(): _call("wcsph_pcisph_update_pos")                                  # noqa: E704  ref: pcisph.py:281-285


def sovel_pressure():
    """ref: pcisph.py:147-157 (the reference's spelling)"""
    global pr_iter
    pr_iter = 0
    err = 0.0
    init_iter_info()
    while (err > 0.01 or pr_iter < 3) and (pr_iter < 50):
        update_iter_info()
        predict_density()
        err = rho_err.to_numpy()[0] / float(particle_data.liquid_count)
        pr_iter += 1


def step():
    """ref: pcisph.py:303-321 main-loop body (headless)"""
    global current_time
    particle_data.hash_grid.update_grid()
    compute_nonpressure_force()
    sovel_pressure()
    update_pos()
    dt = deltaT.to_numpy()[0]
    current_time += dt
    return float(dt)
