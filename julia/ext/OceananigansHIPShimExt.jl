# OceananigansHIPShimExt.jl -- the package extension a maintainer adds to Oceananigans.jl (v0.96.19) to run the
# NonhydrostaticModel hot path on MI355X through libocn_hip.so (C ABI: include/ocn_hip.h).
#
# Pattern: ext/OceananigansMetalExt.jl:1-38 (a device tag for GPU{D}, `architecture`, `on_architecture`), extended with
# overrides of the CALLERS of the hot-path kernels: where the reference does `launch!(arch, grid, workspec, kernel!, args...)`
# (src/Utils/kernel_launching.jl:258-302) this extension does one `ccall` into the library.  Host code stays in Julia.
#
# Julia is not available in the image this library is built in, so this file has never been loaded by a Julia process.
# What IS checked (tests/test_julia_extension.py, no Julia needed): every `ccall((:ocn_..., lib), ...)` below names a symbol
# that include/ocn_hip.h declares, with the same number of arguments, the same return kind and the same kind (pointer,
# 32-bit integer, 64-bit integer, size, double) in every position.  The identical call sequences are exercised on the GPU
# through Python ctypes (oceananigans.jl_amd/) and through a plain C host (examples/c_abi_rk3.c).
#
# Conventions (SURVEY.md 8b): Julia owns every field array (GC-managed HIPShimArray, finalizer -> ocn_free); the library
# keeps pointers only inside explicit handles; arrays are OffsetArray parents (column-major, halos included); errors are a
# Cint status + ocn_last_error(), rethrown here; every call is enqueued on one HIP stream (C_NULL = default) and is
# asynchronous with respect to the host, like a KernelAbstractions launch; ccalls are wrapped in GC.@preserve.
module OceananigansHIPShimExt

using Oceananigans
using Oceananigans.Architectures: CPU, GPU
using Oceananigans.Grids: RectilinearGrid, Periodic, Bounded, Flat, FullyConnected, RightConnected, LeftConnected, topology
using Oceananigans.Fields: Field, XFaceField, YFaceField, ZFaceField
using Oceananigans.Models.NonhydrostaticModels: NonhydrostaticModel
using Oceananigans.TimeSteppers: RungeKutta3TimeStepper, tick!

import Oceananigans.Architectures: architecture, array_type, on_architecture, unsafe_free!
import Oceananigans.Utils: sync_device!
import Oceananigans.BoundaryConditions: fill_halo_regions!
import Oceananigans.TimeSteppers: rk3_substep!, ab2_step!, cache_previous_tendencies!, compute_tendencies!,
                                  calculate_pressure_correction!, pressure_correct_velocities!, time_step!
import Oceananigans.Models.NonhydrostaticModels: solve_for_pressure!, nonhydrostatic_pressure_solver,
                                                  compute_interior_tendency_contributions!
import Oceananigans.Models.NonhydrostaticModels: update_hydrostatic_pressure!
import Oceananigans.TurbulenceClosures: compute_diffusivities!

const lib = "libocn_hip"            # oceananigans.jl_amd/lib/libocn_hip.so on the loader path

"Status -> exception: OCN_ERR_INVALID_ARGUMENT becomes an ArgumentError like the reference's own argument checks."
function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:ocn_last_error, lib), Cstring, ()))
    status == -1 ? throw(ArgumentError(msg)) : error(msg)
end

library_version() = unsafe_string(ccall((:ocn_version, lib), Cstring, ()))

# ---- device array: owns a device pointer, finalizer -> ocn_free (src/Architectures.jl:127-135 pattern) ---------------
mutable struct HIPShimArray{T, N} <: AbstractArray{T, N}
    ptr  :: Ptr{T}
    dims :: NTuple{N, Int}
    function HIPShimArray{T, N}(dims::NTuple{N, Int}) where {T, N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ocn_malloc, lib), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, prod(dims) * sizeof(T)))  # zero-filled
        a = new{T, N}(Ptr{T}(p[]), dims)
        finalizer(x -> ccall((:ocn_free, lib), Cint, (Ptr{Cvoid},), x.ptr), a)
        return a
    end
end
Base.size(a::HIPShimArray) = a.dims
Base.unsafe_convert(::Type{Ptr{T}}, a::HIPShimArray{T}) where T = a.ptr
Base.fill!(a::HIPShimArray{T}, x) where T = (x == 0 || error("HIPShimArray: only zero fill is bound");
    check(ccall((:ocn_memset, lib), Cint, (Ptr{Cvoid}, Cint, Csize_t, Ptr{Cvoid}), a.ptr, 0, prod(a.dims) * sizeof(T), C_NULL)); a)
Base.copyto!(dst::HIPShimArray{T}, src::HIPShimArray{T}) where T = (                        # device_copy_to!
    check(ccall((:ocn_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                dst.ptr, src.ptr, prod(src.dims) * sizeof(T), C_NULL)); dst)

# ---- Architectures (src/Architectures.jl:70-146) ---------------------------------------------------------------------
struct HIPShim end                                    # device tag:  GPU(HIPShim())
const HIPGPU = GPU{HIPShim}
HIPGPU() = GPU(HIPShim())

function device_count()
    n = Ref{Cint}(0)
    check(ccall((:ocn_device_count, lib), Cint, (Ptr{Cint},), n))
    return Int(n[])
end
set_device!(id::Integer) = check(ccall((:ocn_set_device, lib), Cint, (Cint,), id))

architecture(::HIPShimArray) = HIPGPU()
array_type(::HIPGPU) = HIPShimArray
on_architecture(::HIPGPU, a::Number) = a
on_architecture(::HIPGPU, a::HIPShimArray) = a
function on_architecture(::HIPGPU, a::Array{T, N}) where {T, N}
    d = HIPShimArray{T, N}(size(a))
    GC.@preserve a check(ccall((:ocn_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                               d.ptr, pointer(a), sizeof(a), C_NULL))
    return d
end
function on_architecture(::CPU, d::HIPShimArray{T, N}) where {T, N}
    a = Array{T, N}(undef, d.dims)
    GC.@preserve a check(ccall((:ocn_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                               pointer(a), d.ptr, sizeof(a), C_NULL))
    sync_device!(HIPGPU())
    return a
end
unsafe_free!(d::HIPShimArray) = (ccall((:ocn_free, lib), Cint, (Ptr{Cvoid},), d.ptr); d.ptr = C_NULL; nothing)
Base.zeros(::HIPGPU, T, dims...) = HIPShimArray{T, length(dims)}(dims)          # src/Grids/zeros_and_ones.jl:9
sync_device!(::HIPGPU) = check(ccall((:ocn_sync, lib), Cint, (Ptr{Cvoid},), C_NULL))

"OCN_MATH_STRICT (0): the reference's operand order, bit-identical to its CPU arithmetic; OCN_MATH_FAST (1): FMA contraction."
set_math_mode!(mode::Integer) = check(ccall((:ocn_set_math_mode, lib), Cint, (Cint,), mode))
math_mode() = Int(ccall((:ocn_get_math_mode, lib), Cint, ()))
"ocn_grid.math of the grids this extension builds: 0 process default, 1 strict, 2 fast"
const HIP_MATH_MODE = Ref{Int32}(0)

# ---- struct ocn_grid (include/ocn_hip.h) ------------------------------------------------------------------------------
struct OcnGrid
    Nx::Int32; Ny::Int32; Nz::Int32; Hx::Int32; Hy::Int32; Hz::Int32
    tx::Int32; ty::Int32; tz::Int32; math::Int32
    dx::Float64; dy::Float64; dz::Float64; Lx::Float64; Ly::Float64; Lz::Float64
    dzc::Ptr{Float64}; dzf::Ptr{Float64}
end
topo_code(::Type{Periodic}) = Int32(0)
topo_code(::Type{Bounded}) = Int32(1)
topo_code(::Type{Flat}) = Int32(2)
topo_code(::Type{FullyConnected}) = Int32(3)
topo_code(::Type{RightConnected}) = Int32(4)   # the first slab of a Bounded partitioned x: wall on its west side
topo_code(::Type{LeftConnected}) = Int32(5)    # the last one: wall on its east side
const HIPGrid = RectilinearGrid{<:Any, <:Any, <:Any, <:Any, <:Any, <:Any, <:Any, <:HIPGPU}
function OcnGrid(g::RectilinearGrid)
    TX, TY, TZ = topology(g)
    stretched = !(g.z.Δᵃᵃᶜ isa Number)
    # `math`: 0 = the process default (set_math_mode!), 1 = strict, 2 = fast -- a per-grid / per-model choice (HIP_MATH_MODE[] here)
    OcnGrid(g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz, topo_code(TX), topo_code(TY), topo_code(TZ), HIP_MATH_MODE[],
            g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, stretched ? 0.0 : g.z.Δᵃᵃᶜ, g.Lx, g.Ly, g.Lz,
            stretched ? parent(g.z.Δᵃᵃᶜ).ptr : Ptr{Float64}(C_NULL),          # element 0 <-> k = 1-Hz
            stretched ? parent(g.z.Δᵃᵃᶠ).ptr + 8 : Ptr{Float64}(C_NULL))      # parent starts at k = -Hz -> skip one
end
loc_mask(f) = Int32((f isa XFaceField) | (f isa YFaceField) << 1 | (f isa ZFaceField) << 2)
dptr(f) = parent(f).ptr                                                          # Field -> OffsetArray -> HIPShimArray
dptrs(fields) = Ptr{Float64}[dptr(f) for f in fields]
locs(fields) = Int32[loc_mask(f) for f in fields]
range6(::Nothing) = Ptr{Int32}(C_NULL)                                           # :xyz, periphery excluded
range6(r::NTuple{6, Integer}) = Int32[r...]                                      # KernelParameters -> {i0,i1,j0,j1,k0,k1}

# ---- halo fills: fill_halo_regions!(fields...) (BoundaryConditions/fill_halo_regions.jl:50-67, Fields/field_tuples.jl:56-101)
function fill_halo_regions!(fields::NTuple{N, Field{<:Any, <:Any, <:Any, <:Any, <:HIPGrid}}, args...;
                            fill_boundary_normal_velocities = true, kw...) where N
    g = Ref(OcnGrid(first(fields).grid)); ptrs = dptrs(fields); ls = locs(fields)
    GC.@preserve fields ptrs ls check(ccall((:ocn_fill_halo_regions, lib), Cint,
        (Ref{OcnGrid}, Ptr{Ptr{Float64}}, Ptr{Int32}, Int32, Int32, Ptr{Cvoid}),
        g, ptrs, ls, N, fill_boundary_normal_velocities, C_NULL))
end
"fill_periodic_{west_and_east,south_and_north,bottom_and_top}_halo! (fill_halo_regions_periodic.jl:18-111), dir = 0, 1, 2"
function fill_periodic_halo!(fields, dir::Integer)
    g = Ref(OcnGrid(first(fields).grid)); ptrs = dptrs(fields); ls = locs(fields)
    GC.@preserve fields ptrs ls check(ccall((:ocn_fill_halo_periodic, lib), Cint,
        (Ref{OcnGrid}, Ptr{Ptr{Float64}}, Ptr{Int32}, Int32, Int32, Ptr{Cvoid}), g, ptrs, ls, length(fields), dir, C_NULL))
end

# ---- tendencies: compute_interior_tendency_contributions! (compute_nonhydrostatic_tendencies.jl:57-139) --------------
const HIPModel = NonhydrostaticModel{<:Any, <:Any, <:HIPGPU}
function compute_interior_tendency_contributions!(model::HIPModel, kernel_parameters; active_cells_map = nothing)
    g = Ref(OcnGrid(model.grid)); U = model.velocities; G = model.timestepper.Gⁿ; r = range6(kernel_parameters)
    GC.@preserve model r begin
        check(ccall((:ocn_compute_momentum_tendencies, lib), Cint,
              (Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}),
              g, dptr(U.u), dptr(U.v), dptr(U.w), dptr(G.u), dptr(G.v), dptr(G.w), r, C_NULL))
        for (n, c) in enumerate(model.tracers)
            check(ccall((:ocn_compute_tracer_tendency, lib), Cint,
                  (Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}),
                  g, dptr(U.u), dptr(U.v), dptr(U.w), dptr(c), dptr(G[n + 3]), r, C_NULL))
        end
    end
    return nothing
end

# ---- time steppers (runge_kutta_3.jl:160-208, quasi_adams_bashforth_2.jl:128-175, store_tendencies.jl:12-22) ----------
function stepper_arrays(model)
    fields = prognostic_fields(model)
    return fields, dptrs(fields), dptrs(model.timestepper.Gⁿ), dptrs(model.timestepper.G⁻), locs(fields)
end
function rk3_substep!(model::HIPModel, Δt, γ, ζ)
    fields, U, Gn, Gm, ls = stepper_arrays(model); g = Ref(OcnGrid(model.grid))
    GC.@preserve model U Gn Gm ls check(ccall((:ocn_rk3_substep, lib), Cint,
        (Ref{OcnGrid}, Int32, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Float64, Float64, Float64, Int32, Ptr{Cvoid}),
        g, length(fields), U, Gn, Gm, ls, Δt, γ, isnothing(ζ) ? 0.0 : ζ, !isnothing(ζ), C_NULL))
end
function ab2_step!(model::HIPModel, Δt)
    fields, U, Gn, Gm, ls = stepper_arrays(model); g = Ref(OcnGrid(model.grid))
    GC.@preserve model U Gn Gm ls check(ccall((:ocn_ab2_step, lib), Cint,
        (Ref{OcnGrid}, Int32, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Float64, Float64, Ptr{Cvoid}),
        g, length(fields), U, Gn, Gm, ls, Δt, model.timestepper.χ, C_NULL))
end
function cache_previous_tendencies!(model::HIPModel)
    fields, U, Gn, Gm, ls = stepper_arrays(model); g = Ref(OcnGrid(model.grid))
    GC.@preserve model Gn Gm ls check(ccall((:ocn_cache_previous_tendencies, lib), Cint,
        (Ref{OcnGrid}, Int32, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Ptr{Cvoid}), g, length(fields), Gm, Gn, ls, C_NULL))
end

# ---- pressure (NonhydrostaticModels.jl:25-62, solve_for_pressure.jl:57-82, pressure_correction.jl:8-50) ----------------
mutable struct HIPPoissonSolver
    handle :: Ptr{Cvoid}
    grid
end
function nonhydrostatic_pressure_solver(::HIPGPU, grid::RectilinearGrid)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ocn_poisson_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}), h, Ref(OcnGrid(grid))))
    s = HIPPoissonSolver(h[], grid)
    finalizer(x -> ccall((:ocn_poisson_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end
"kind 0 FFT-based, 1 Fourier-tridiagonal, 2 FFT-based with cosine transforms; r2c; inverse transform writes p directly"
function solver_info(s::HIPPoissonSolver)
    kind = Ref{Int32}(0); r2c = Ref{Int32}(0); direct = Ref{Int32}(0)
    check(ccall((:ocn_poisson_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}), s.handle, kind, r2c, direct))
    return (kind = kind[], r2c = r2c[] != 0, direct_out = direct[] != 0)
end
solve_for_pressure!(p, s::HIPPoissonSolver, Δt, U) = GC.@preserve p U check(ccall((:ocn_solve_for_pressure, lib), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Cvoid}),
    s.handle, dptr(p), dptr(U.u), dptr(U.v), dptr(U.w), Δt, C_NULL))
"compute_source_term! + solve!(ϕ, solver) as two calls (fft_based_poisson_solver.jl:95-125)"
function solve_for_pressure_in_two_calls!(p, s::HIPPoissonSolver, Δt, U)
    GC.@preserve p U begin
        check(ccall((:ocn_poisson_compute_source_term, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Cvoid}), s.handle, dptr(U.u), dptr(U.v), dptr(U.w), Δt, C_NULL))
        check(ccall((:ocn_poisson_solve, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), s.handle, dptr(p), C_NULL))
    end
end
pressure_correct_velocities!(model::HIPModel, Δt) = GC.@preserve model check(ccall(
    (:ocn_pressure_correct_velocities, lib), Cint,
    (Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Cvoid}),
    Ref(OcnGrid(model.grid)), dptr(model.velocities.u), dptr(model.velocities.v), dptr(model.velocities.w),
    dptr(model.pressures.pNHS), Δt, C_NULL))
"divᶜᶜᶜ of (u, v, w) into a halo-free Nx x Ny x Nz device array (tests: max|∇·u|)"
divergence!(div::HIPShimArray, grid, U) = GC.@preserve div U check(ccall((:ocn_divergence, lib), Cint,
    (Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    Ref(OcnGrid(grid)), dptr(U.u), dptr(U.v), dptr(U.w), div.ptr, C_NULL))

# ---- the whole fused RK3 time_step! behind one call (csrc/driver.hip; runge_kutta_3.jl:77-151) --------------------------
mutable struct HIPRK3Driver
    handle :: Ptr{Cvoid}
end
function HIPRK3Driver(model::HIPModel)               # after set!(model, ...): the interiors hold the initial velocities
    h = Ref{Ptr{Cvoid}}(C_NULL); U = model.velocities
    GC.@preserve model check(ccall((:ocn_rk3_driver_create, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
        h, Ref(OcnGrid(model.grid)), dptr(U.u), dptr(U.v), dptr(U.w), dptr(model.pressures.pNHS), model.pressure_solver.handle, C_NULL))
    d = HIPRK3Driver(h[])
    finalizer(x -> ccall((:ocn_rk3_driver_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), d)
    return d
end
const drivers = IdDict{Any, HIPRK3Driver}()
driver(model) = get!(() -> HIPRK3Driver(model), drivers, model)
function time_step!(model::NonhydrostaticModel{<:RungeKutta3TimeStepper, <:Any, <:HIPGPU}, Δt; callbacks = [])
    check(ccall((:ocn_rk3_driver_time_step, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), driver(model).handle, Δt, C_NULL))
    tick!(model.clock, Δt)                          # clock bookkeeping stays in Julia (runge_kutta_3.jl:95-142)
    return nothing
end
"before output writers / diagnostics / checkpoints read the fields: velocities home, deferred tendencies completed"
flush!(model::HIPModel) = check(ccall((:ocn_rk3_driver_flush, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), driver(model).handle, C_NULL))

# ---- config 4 physics (SURVEY 8f): struct ocn_model_terms and the *_terms entry points --------------------------------
struct OcnModelTerms
    advection::Int32; coriolis::Int32; closure::Int32; buoyancy::Int32
    f::Float64; nu::Float64; g::Float64; alpha::Float64; beta::Float64
    T::Ptr{Float64}; S::Ptr{Float64}; pHY::Ptr{Float64}; nu_e::Ptr{Float64}
    coriolis_beta::Float64; yc::Ptr{Float64}; yf::Ptr{Float64}          # BetaPlane: f = f + coriolis_beta * ynode (parent(grid.yᵃᶜᵃ), parent(grid.yᵃᶠᵃ))
end
function compute_tendencies_with_terms!(model::HIPModel, t::OcnModelTerms, κ::Vector{Float64}, κₑ::Vector{Ptr{Float64}}, kernel_parameters)
    g = Ref(OcnGrid(model.grid)); U = model.velocities; G = model.timestepper.Gⁿ; r = range6(kernel_parameters); tr = Ref(t)
    GC.@preserve model r begin
        check(ccall((:ocn_compute_momentum_tendencies_terms, lib), Cint,
              (Ref{OcnGrid}, Ref{OcnModelTerms}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}),
              g, tr, dptr(U.u), dptr(U.v), dptr(U.w), dptr(G.u), dptr(G.v), dptr(G.w), r, C_NULL))
        for (n, c) in enumerate(model.tracers)
            check(ccall((:ocn_compute_tracer_tendency_terms, lib), Cint,
                  (Ref{OcnGrid}, Ref{OcnModelTerms}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Cvoid}),
                  g, tr, κ[n], κₑ[n], dptr(U.u), dptr(U.v), dptr(U.w), dptr(c), dptr(G[n + 3]), r, C_NULL))
        end
    end
end
update_hydrostatic_pressure!(pHY′, grid::HIPGrid, t::OcnModelTerms) = GC.@preserve pHY′ check(ccall(
    (:ocn_update_hydrostatic_pressure, lib), Cint, (Ref{OcnGrid}, Ref{OcnModelTerms}, Ptr{Float64}, Ptr{Cvoid}),
    Ref(OcnGrid(grid)), Ref(t), dptr(pHY′), C_NULL))
function compute_amd_diffusivities!(νₑ, κₑ, grid::HIPGrid, Cν, Cκ::Vector{Float64}, U, tracers)
    cs = dptrs(tracers); ks = dptrs(κₑ)
    GC.@preserve νₑ κₑ U tracers cs ks Cκ check(ccall((:ocn_compute_amd_diffusivities, lib), Cint,
        (Ref{OcnGrid}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Cvoid}),
        Ref(OcnGrid(grid)), Cν, dptr(U.u), dptr(U.v), dptr(U.w), dptr(νₑ), length(tracers), Cκ, cs, ks, C_NULL))
end

# ---- diagnostics that stay on the device until the host reads them ----------------------------------------------------
"hasnan(field) (Models/nan_checker.jl:33): flag is a 1-element Int32 device array"
hasnan!(flag::HIPShimArray{Int32}, f) = GC.@preserve flag f check(ccall((:ocn_hasnan, lib), Cint,
    (Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Cvoid}), dptr(f), length(parent(f)), flag.ptr, C_NULL))
"cell_advection_timescale(grid, velocities) (Advection/cell_advection_timescale.jl:13-35)"
cell_advection_timescale!(out::HIPShimArray{Float64}, grid, U) = GC.@preserve out U check(ccall((:ocn_cell_advection_timescale, lib), Cint,
    (Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    Ref(OcnGrid(grid)), dptr(U.u), dptr(U.v), dptr(U.w), out.ptr, C_NULL))

# ---- Distributed (src/DistributedComputations/): the collectives are RCCL INSIDE the library (csrc/comm.hip); the Julia
#      side needs no MPI.jl in the time-stepping loop.  Only the 128-byte unique id has to reach every rank once. ---------
mutable struct HIPShimComm
    handle :: Ptr{Cvoid}
    rank   :: Int
    nranks :: Int
end
function HIPShimComm(rank, nranks, local_device, bcast)   # bcast(bytes) -> rank 0's bytes on every rank (MPI.bcast, a file, a socket ...)
    id = zeros(UInt8, 128)
    rank == 0 && check(ccall((:ocn_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    id = bcast(id)
    set_device!(local_device)                                                                   # distributed_architectures.jl:280-282
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ocn_comm_init, lib), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int32, Ptr{Cvoid}), h, rank, nranks, id))
    c = HIPShimComm(h[], rank, nranks)
    finalizer(x -> ccall((:ocn_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), c)
    return c
end
function comm_info(c::HIPShimComm)
    r = Ref{Int32}(0); n = Ref{Int32}(0); v = Ref{Int32}(0)
    check(ccall((:ocn_comm_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}), c.handle, r, n, v))
    return (rank = r[], nranks = n[], rccl_version = v[])
end
# fill_halo_event!(c, kernels!, bcs, loc, grid::DistributedGrid, buffers; async) (halo_communication.jl:210-229) for west / east:
#   local (y, z) fills first (fill_halo_regions.jl:148-196: ocn_fill_halo_regions on the FullyConnected local grid), then
function fill_x_halos_begin!(c::HIPShimComm, grid, fields)
    ptrs = dptrs(fields); ls = locs(fields)
    GC.@preserve fields ptrs ls check(ccall((:ocn_halo_exchange_begin, lib), Cint,
        (Ptr{Cvoid}, Ref{OcnGrid}, Ptr{Ptr{Float64}}, Ptr{Int32}, Int32, Ptr{Cvoid}), c.handle, Ref(OcnGrid(grid)), ptrs, ls, length(fields), C_NULL))
end
# ... interior tendencies run here (interleave_communication_and_computation.jl:29-67) ...
function synchronize_communication!(c::HIPShimComm, grid, fields)                              # distributed_fields.jl:58-75
    ptrs = dptrs(fields); ls = locs(fields)
    GC.@preserve fields ptrs ls check(ccall((:ocn_halo_exchange_end, lib), Cint,
        (Ptr{Cvoid}, Ref{OcnGrid}, Ptr{Ptr{Float64}}, Ptr{Int32}, Int32, Ptr{Cvoid}), c.handle, Ref(OcnGrid(grid)), ptrs, ls, length(fields), C_NULL))
end
#   no sync_device!, no Waitall: the exchange is ordered against the compute stream by events inside the library.
"one x plane instead of 2 Hx strips, for the two synchronous fills inside the projection: side 0 east (u[nx+1]), 1 west (p[0])"
exchange_plane!(c::HIPShimComm, grid, f, side::Integer) = GC.@preserve f check(ccall((:ocn_halo_exchange_plane, lib), Cint,
    (Ptr{Cvoid}, Ref{OcnGrid}, Ptr{Float64}, Int32, Int32, Ptr{Cvoid}), c.handle, Ref(OcnGrid(grid)), dptr(f), loc_mask(f), side, C_NULL))
"MPI.Allreduce of Δt / max|u| (simulation.jl:128-134): op 0 sum, 1 max, 2 min on a device buffer"
allreduce!(c::HIPShimComm, buf::HIPShimArray{Float64}, op::Integer) = GC.@preserve buf check(ccall((:ocn_comm_allreduce, lib), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Csize_t, Int32, Ptr{Cvoid}), c.handle, buf.ptr, length(buf), op, C_NULL))
barrier(c::HIPShimComm) = check(ccall((:ocn_comm_barrier, lib), Cint, (Ptr{Cvoid},), c.handle))
# Host waits with a deadline (a peer that never arrives ends in an error of THIS rank, not in a hang) and device-side exchange timing
sync_device_with_deadline(seconds::Real) = check(ccall((:ocn_sync_timeout, lib), Cint, (Ptr{Cvoid}, Float64), C_NULL, seconds))
wait_for_communication(c::HIPShimComm, seconds::Real) = check(ccall((:ocn_comm_wait, lib), Cint, (Ptr{Cvoid}, Float64), c.handle, seconds))
enable_exchange_timing!(c::HIPShimComm, on::Bool) = check(ccall((:ocn_comm_enable_stats, lib), Cint, (Ptr{Cvoid}, Int32), c.handle, on))
function exchange_timing(c::HIPShimComm)   # ms since the last call: strips, halo wait, solve exchange, pressure planes, single planes, counts
    ms = zeros(Float64, 8)
    GC.@preserve ms check(ccall((:ocn_comm_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.handle, ms))
    ms
end

# DistributedFFTBasedPoissonSolver / DistributedFourierTridiagonalPoissonSolver (distributed_fft_based_poisson_solver.jl:141-178)
mutable struct HIPDistributedPoissonSolver
    handle :: Ptr{Cvoid}
    comm   :: HIPShimComm
    fast   :: Int32      # ocn_dist_poisson_pipeline: 0 transposing rocFFT path, 1 / 2 slab pipelines, 3 transpose-free (csrc/xtri.hip)
end
function HIPDistributedPoissonSolver(local_grid, c::HIPShimComm, global_Lx; global_x_topology = Periodic)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    # (topology(global_grid, 1): the slabs of a Bounded x are RightConnected / FullyConnected / LeftConnected, so the solver is told)
    check(ccall((:ocn_dist_poisson_create_global, lib), Cint, (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}, Int32, Int32, Float64, Int32),
                h, Ref(OcnGrid(local_grid)), c.rank, c.nranks, global_Lx, topo_code(global_x_topology)))
    fast = Ref{Int32}(0)
    check(ccall((:ocn_dist_poisson_pipeline, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), h[], fast))
    s = HIPDistributedPoissonSolver(h[], c, fast[])
    finalizer(x -> ccall((:ocn_dist_poisson_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end
function solve_for_pressure!(p, s::HIPDistributedPoissonSolver, Δt, U)
    s.fast == 0 && error("the transposing path needs the four ocn_transpose_* calls as well: see INTEGRATION.md")
    GC.@preserve p U begin
        check(ccall((:ocn_dist_poisson_source_term, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Cvoid}),
                    s.handle, dptr(U.u), dptr(U.v), dptr(U.w), Δt, C_NULL))
        check(ccall((:ocn_dist_poisson_forward_yz, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.handle, C_NULL))
        check(ccall((:ocn_dist_poisson_exchange, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}), s.handle, s.comm.handle, 0, C_NULL))
        check(ccall((:ocn_dist_poisson_solve_x, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.handle, C_NULL))
        check(ccall((:ocn_dist_poisson_exchange, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}), s.handle, s.comm.handle, 1, C_NULL))
        check(ccall((:ocn_dist_poisson_backward_yz, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), s.handle, dptr(p), C_NULL))
    end
end

# ---- the distributed step behind one call per rank (csrc/driver.hip: exchanges and collectives are issued by the library too) ----------
"ONE RANK of a slab-x run: u*, v*, w* strips fly under the Poisson solve, only pressure planes cross after it, one full-slab launch"
function HIPRK3Driver(model::HIPModel, s::HIPDistributedPoissonSolver)
    h = Ref{Ptr{Cvoid}}(C_NULL); U = model.velocities
    GC.@preserve model check(ccall((:ocn_rk3_driver_create_distributed, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        h, Ref(OcnGrid(model.grid)), dptr(U.u), dptr(U.v), dptr(U.w), dptr(model.pressures.pNHS), s.handle, s.comm.handle, C_NULL))
    d = HIPRK3Driver(h[])
    finalizer(x -> ccall((:ocn_rk3_driver_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), d)
    return d
end
"defer = false: pressure_correct_velocities! of the third stage runs inside time_step! (one GPU only)"
defer_correction!(d::HIPRK3Driver, defer::Bool) = check(ccall((:ocn_rk3_driver_configure, lib), Cint, (Ptr{Cvoid}, Int32), d.handle, defer))
"where u, v, w and Gⁿ are right now (device pointers; valid until the next call)"
function driver_fields(d::HIPRK3Driver)
    r = [Ref{Ptr{Float64}}(C_NULL) for _ in 1:6]
    check(ccall((:ocn_rk3_driver_fields, lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}},
                Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), d.handle, r[1], r[2], r[3], r[4], r[5], r[6]))
    return map(x -> x[], r)
end

# ---- config 4's term set behind one call (csrc/model_driver.hip): struct ocn_bc, ocn_field_bcs, ocn_model_driver_desc -------------------
struct OcnBc
    kind::Int32; _pad::Int32
    value::Float64; coeff::Float64
    values::Ptr{Float64}
end
OcnBc() = OcnBc(0, 0, 0.0, 0.0, C_NULL)
struct OcnFieldBcs
    west::OcnBc; east::OcnBc; south::OcnBc; north::OcnBc; bottom::OcnBc; top::OcnBc
end
struct OcnModelDriverDesc
    terms::OcnModelTerms
    n_tracers::Int32; tracer_T::Int32; tracer_S::Int32; _pad::Int32
    kappa::NTuple{4, Float64}
    C_nu::Float64; C_kappa::NTuple{4, Float64}
    tracers::NTuple{4, Ptr{Float64}}
    nu_e::Ptr{Float64}; kappa_e::NTuple{4, Ptr{Float64}}
    pHY::Ptr{Float64}
    bcs::NTuple{7, Ptr{OcnFieldBcs}}
end
mutable struct HIPModelDriver
    handle :: Ptr{Cvoid}
    keep   :: Any                                     # the boundary-condition structs, alive while create runs
end
function HIPModelDriver(model::HIPModel, desc::OcnModelDriverDesc; keep = nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL); U = model.velocities
    GC.@preserve model keep check(ccall((:ocn_model_driver_create, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}, Ref{OcnModelDriverDesc}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
        h, Ref(OcnGrid(model.grid)), Ref(desc), dptr(U.u), dptr(U.v), dptr(U.w), dptr(model.pressures.pNHS), model.pressure_solver.handle, C_NULL))
    d = HIPModelDriver(h[], nothing)
    finalizer(x -> ccall((:ocn_model_driver_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), d)
    return d
end
"ONE RANK of a slab-x run of the same term set: every exchange (strips, planes, the solver's transposes) is issued by the library"
function HIPModelDriver(model::HIPModel, desc::OcnModelDriverDesc, s::HIPDistributedPoissonSolver; keep = nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL); U = model.velocities
    GC.@preserve model keep check(ccall((:ocn_model_driver_create_distributed, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ref{OcnGrid}, Ref{OcnModelDriverDesc}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        h, Ref(OcnGrid(model.grid)), Ref(desc), dptr(U.u), dptr(U.v), dptr(U.w), dptr(model.pressures.pNHS), s.handle, s.comm.handle, C_NULL))
    d = HIPModelDriver(h[], nothing)
    finalizer(x -> ccall((:ocn_model_driver_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), d)
    return d
end
"time_step!(model, Δt) with tracers, closures, buoyancy and boundary fluxes: halo fills, compute_auxiliaries!, tendencies, projection"
time_step!(d::HIPModelDriver, Δt) = check(ccall((:ocn_model_driver_time_step, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), d.handle, Δt, C_NULL))
flush!(d::HIPModelDriver) = check(ccall((:ocn_model_driver_flush, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), d.handle, C_NULL))
function driver_field(d::HIPModelDriver, n::Integer)              # 0, 1, 2 = u, v, w; 3 + n = tracer n
    f = Ref{Ptr{Float64}}(C_NULL); G = Ref{Ptr{Float64}}(C_NULL)
    check(ccall((:ocn_model_driver_field, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), d.handle, n, f, G))
    return (field = f[], G = G[])
end

end # module
