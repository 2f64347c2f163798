"""oceananigans.jl_amd -- MI355X-native NonhydrostaticModel time-stepping hot path of Oceananigans.jl.

Host-side mirror of the reference's operator surface (Architectures / Grids / Fields / Advection / Solvers /
TimeSteppers / Models.NonhydrostaticModels / DistributedComputations) over the C ABI of libocn_hip.so.
The directory name contains a dot, so import it through the root-level shim:

    import oceananigans_jl_amd as ocn
"""
from . import _lib
from ._lib import MATH_FAST, MATH_STRICT, OcnError
from .advection import WENO, UpwindBiased
from .architectures import CPU, GPU, on_architecture, sync_device, zeros
from .distributed import (Distributed, DistributedFFTBasedPoissonSolver, DistributedFourierTridiagonalPoissonSolver, Partition,
                          TorchDistributedFabric)
from .fields import CenterField, Field, XFaceField, YFaceField, ZFaceField, fill_halo_regions
from .grids import Bounded, Center, Face, Flat, FullyConnected, LeftConnected, Periodic, RectilinearGrid, RightConnected
from .models import (NonhydrostaticModel, QuasiAdamsBashforth2TimeStepper, RungeKutta3TimeStepper, ab2_step,
                     cache_previous_tendencies, calculate_pressure_correction, compute_auxiliaries, compute_diffusivities,
                     compute_tendencies, flush_tendencies, RK3Driver, ModelRK3Driver,
                     pressure_correct_velocities, rk3_substep, set, solve_for_pressure, time_step, update_hydrostatic_pressure,
                     update_state)
from .output import (AdvectiveCFL, DiffusiveCFL, NaNChecker, TimeStepWizard, cell_advection_timescale, cell_diffusion_timescale, hasnan, set_from_checkpoint,
                     write_checkpoint)
from .physics import (AnisotropicMinimumDissipation, BetaPlane, BoundaryCondition, BuoyancyTracer, Centered, FieldBoundaryConditions, FluxBoundaryCondition, FPlane,
                      GradientBoundaryCondition, LinearEquationOfState, OpenBoundaryCondition, ScalarDiffusivity, SeawaterBuoyancy,
                      ValueBoundaryCondition)
from .solvers import (BatchedTridiagonalSolver, FFTBasedPoissonSolver, FourierTridiagonalPoissonSolver,
                      nonhydrostatic_pressure_solver, solve)


from .hydrostatic import (AdamsBashforth3Scheme, ExplicitFreeSurface, ForwardBackwardScheme, HydrostaticFreeSurfaceModel, ImplicitFreeSurface, SplitExplicitFreeSurface,  # noqa: E402
                          VectorInvariant)


def set_math_mode(mode):
    """MATH_STRICT: reference evaluation order (bit-reproducible vs the CPU oracle); MATH_FAST: FMA + fused division."""
    _lib.call("ocn_set_math_mode", int(mode))
