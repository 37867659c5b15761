"""Headless drop-in for the reference's sesph.py driver module (ref: sesph.py:1-239):
same function names (init_particle, reset_param, update_advection_density, update_pressure,
compute_force, integrator_sesph) and globals (particle_data, vel, d_vel, rho, pressure,
deltaT); the reference main-loop body (sesph.py:220-225) runs verbatim.

The reference script keeps private vel/d_vel/rho/pressure fields and private kernel
constants (pi = 3.1415926) and passes the particle DIAMETER to ParticleData (Q11); both are
reproduced: the fields below alias the ParticleData ones, the constants travel in
wcsph_params.ss_*.
"""
import numpy as np

from . import _lib, scene
from .ParticleData import ParticleData
from .field import ScalarField

current_time = 0.0
total_time = 5.0
eps = 1e-5
test_id = 1000

particleRadius = 0.025
gridR = particleRadius * 2.0
searchR = gridR * 2.0
invGridR = 1.0 / gridR
particleDimX = particleDimY = particleDimZ = 20
particleLiquidNum = particleDimX * particleDimY * particleDimZ
boundary = 2.0
rho_0 = 1000.0
VL0 = particleRadius * particleRadius * particleRadius * 0.8 * 8.0
VS0 = VL0 * 2.0
liqiudMass = VL0 * rho_0
stiffness = 50000.0
dim_coff = 10.0
viscosity = 0.1
viscosity_b = 0.0

particle_data = None
vel = d_vel = rho = pressure = deltaT = None


def _call(name):
    ctx = particle_data._need_ctx()
    _lib.check(ctx, getattr(_lib.load(), name)(ctx))


def init_particle(filename=None, neighbor_mode="faithful"):
    """ref: sesph.py:66-92 (the filename argument is unused by the reference as well)."""
    global particle_data, vel, d_vel, rho, pressure, deltaT, current_time
    current_time = 0.0
    particle_data = ParticleData(gridR, neighbor_mode=neighbor_mode)  # the DIAMETER, as the reference does (Q11)
    particle_data.sesph_particle_radius = particleRadius
    particle_data.list_radius = searchR  # the script's kernel support (0.1), not the hash grid's 0.2 walk radius
    liquid, solid = scene.sesph_scene(particleRadius, particleDimX, boundary)
    particle_data.add_liquid_points(liquid)
    particle_data.add_solid_points(solid)
    particle_data.setup_data_gpu()
    particle_data.setup_data_cpu()
    vel, d_vel, rho, pressure = particle_data.vel, particle_data.d_vel, particle_data.rho, particle_data.pressure
    deltaT = ScalarField(particle_data, "deltaT")
    return particle_data


def reset_param(): _call("wcsph_sesph_reset_param")                  # noqa: E704  ref: sesph.py:130-135
def update_advection_density(): _call("wcsph_sesph_density")         # noqa: E704  ref: sesph.py:138-154
def update_pressure(): _call("wcsph_sesph_pressure")                 # noqa: E704  ref: sesph.py:158-166
def compute_force(): _call("wcsph_sesph_force")                      # noqa: E704  ref: sesph.py:168-188
def integrator_sesph(): _call("wcsph_sesph_integrate")               # noqa: E704  ref: sesph.py:191-196


def step():
    """ref: sesph.py:217-236 main-loop body (headless)."""
    global current_time
    particle_data.hash_grid.update_grid()

    update_advection_density()
    update_pressure()
    compute_force()
    integrator_sesph()

    dt = deltaT.to_numpy()[0]
    current_time += dt
    return float(dt)


def main(steps=100):
    init_particle("boundry.obj")
    reset_param()
    for _ in range(steps):
        dt = step()
        print("time:%.3f" % current_time, "step:%.4f" % dt)


if __name__ == "__main__":
    main()
