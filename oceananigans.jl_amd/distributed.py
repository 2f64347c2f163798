"""DistributedComputations, slab-x: mirrors src/DistributedComputations/ for `Partition(R)` along x.

  Distributed(child_arch; partition=Partition(R))           distributed_architectures.jl:167-297
  RectilinearGrid(arch::Distributed, ...) -> local grid       distributed_grids.jl:75-118 (x becomes FullyConnected, :339-346)
  fill_halo_regions! with x communication                     halo_communication.jl:100-366, Fields/field_boundary_buffers.jl:276-308
  interior / buffer tendency split, async exchange            Models/interleave_communication_and_computation.jl:9-67,
                                                              NonhydrostaticModels/compute_nonhydrostatic_buffer_tendencies.jl:10-83
  DistributedFFTBasedPoissonSolver                            distributed_fft_based_poisson_solver.jl:10-188
  transposes + all-to-all                                     distributed_transpose.jl:25-191, transposable_field.jl:4-104

MI355X design: one process per GPU; the data path is RCCL over xGMI BEHIND THE C ABI (csrc/comm.hip: ocn_comm_init,
ocn_halo_exchange_begin / _end, ocn_halo_exchange_plane, ocn_dist_poisson_exchange; `RcclFabric` below binds them through
ctypes like every other entry point).  Halo exchange = one grouped send/recv pair per neighbour for the whole field tuple
(6.4 MB per field-side at 512^3/8) on the communicator's own stream, ordered by events and overlapped with the interior
tendency kernel; transposes = one grouped all-to-all of equal 33.5 MB chunks (one xGMI link per peer at R = 8).
torch.distributed is used only (a) with the gloo backend to hand the 128-byte RCCL unique id to every rank and for host
barriers, and (b) as `TorchDistributedFabric` so that the same choreography can be exercised on CPU ranks (gloo) in tests with
injected CPU `ops`; the product ops are the HIP kernels of libocn_hip.
"""
import ctypes as C

import numpy as np
import os

import torch

from . import _lib
from .architectures import GPU, device, stream_ptr
from .grids import Bounded, FullyConnected, LeftConnected, Periodic, RectilinearGrid, RightConnected


class Partition:
    """Partition(Rx): equal slabs along x.  y/z partitions are outside the north-star scope."""

    def __init__(self, x=1, y=1, z=1):
        if y != 1 or z != 1:
            raise NotImplementedError("only slab-x partitions (Partition(R)) are implemented")
        if x < 1:
            raise ValueError("partition size must be >= 1")
        self.x, self.y, self.z = int(x), 1, 1


# --------------------------------------------------------------------------------------------------
# Fabric: point-to-point neighbour exchange and all-to-all between the R ranks.
# --------------------------------------------------------------------------------------------------
class TorchDistributedFabric:
    """torch.distributed process group (nccl = RCCL on GPUs, gloo on CPU ranks in tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("Distributed() needs torch.distributed.init_process_group() (one process per GPU)")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def start_exchange(self, sends, recvs):
        """sends: [(tensor, dest_rank)], recvs: [(tensor, src_rank)] in matching order per peer. Returns a waitable."""
        dist = self.dist
        ops = [dist.P2POp(dist.irecv, t, src, self.group) for t, src in recvs]
        ops += [dist.P2POp(dist.isend, t, dst, self.group) for t, dst in sends]
        return dist.batch_isend_irecv(ops)

    @staticmethod
    def wait(reqs):
        for r in reqs:
            r.wait()

    def all_to_all(self, recv, send):
        self.dist.all_to_all_single(recv, send, group=self.group)

    def all_gather(self, recv, send):
        self.dist.all_gather(list(recv.view(self.size, -1).unbind(0)), send, group=self.group)

    def allreduce_max(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t


class RcclFabric:
    """The product transport: an RCCL communicator owned by libocn_hip (include/ocn_hip.h, ocn_comm_*).  `bootstrap(obj)` must
    broadcast a bytes object from rank 0 to every rank (make_distributed uses a gloo process group; a Julia host would use any
    channel it likes)."""

    def __init__(self, rank, size, bootstrap):
        self.rank, self.size = int(rank), int(size)
        uid = (C.c_ubyte * 128)()
        if self.rank == 0:
            _lib.call("ocn_comm_unique_id", uid)
        raw = bootstrap(bytes(uid))
        uid = (C.c_ubyte * 128)(*raw)
        self._h = C.c_void_p()
        _lib.call("ocn_comm_init", C.byref(self._h), self.rank, self.size, uid)
        self._scratch = None

    def info(self):
        r, n, v = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.call("ocn_comm_info", self._h, C.byref(r), C.byref(n), C.byref(v))
        return {"rank": r.value, "ranks_seen_by_rccl": n.value, "rccl_version": v.value}

    # the x halo exchange of a field tuple: pack, grouped send / recv on the communication stream, wait (event), unpack
    def halo_exchange_begin(self, grid, fields):
        _lib.call("ocn_halo_exchange_begin", self._h, grid.cref, _lib.ptr_array([f.ptr for f in fields]),
                  _lib.i32_array([f.loc for f in fields]), len(fields), stream_ptr())

    def halo_exchange_end(self, grid, fields):
        _lib.call("ocn_halo_exchange_end", self._h, grid.cref, _lib.ptr_array([f.ptr for f in fields]),
                  _lib.i32_array([f.loc for f in fields]), len(fields), stream_ptr())

    def halo_exchange_plane(self, grid, f, side):
        _lib.call("ocn_halo_exchange_plane", self._h, grid.cref, f.ptr, f.loc, 0 if side == "east" else 1, stream_ptr())

    def exchange_strips(self, send_west, send_east, recv_west, recv_east):
        _lib.call("ocn_comm_exchange_strips", self._h, send_west.data_ptr(), send_east.data_ptr(), recv_west.data_ptr(), recv_east.data_ptr(),
                  send_west.numel(), stream_ptr())

    def halo_exchange_pressure(self, grid, p, u, dt_correct):
        _lib.call("ocn_halo_exchange_pressure", self._h, grid.cref, p.ptr, u.ptr, float(dt_correct), stream_ptr())

    def dist_poisson_exchange(self, handle, direction):
        _lib.call("ocn_dist_poisson_exchange", handle, self._h, int(direction), stream_ptr())

    def all_to_all(self, recv, send):
        _lib.call("ocn_comm_all_to_all", self._h, send.data_ptr(), recv.data_ptr(), send.numel() // self.size, stream_ptr())

    def all_gather(self, recv, send):
        _lib.call("ocn_comm_all_gather", self._h, send.data_ptr(), recv.data_ptr(), send.numel(), stream_ptr())

    def allreduce_max(self, t):
        _lib.call("ocn_comm_allreduce", self._h, t.data_ptr(), t.numel(), 1, stream_ptr())
        return t

    def allreduce_sum(self, t):
        _lib.call("ocn_comm_allreduce", self._h, t.data_ptr(), t.numel(), 0, stream_ptr())
        return t

    def barrier(self):
        torch.cuda.current_stream().synchronize()
        _lib.call("ocn_comm_barrier", self._h)

    def close(self):
        if self._h is not None and self._h.value:
            _lib.lib().ocn_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalFabric(RcclFabric):
    """The library's in-process transport (ocn_comm_init_local): the ranks are THREADS of this process sharing one GPU.  Same entry
    points, schedules, pack / unpack launches and stream ordering as the RCCL transport -- what a one-GPU box can exercise of an R-rank
    run besides RCCL itself (RCCL refuses two ranks on one device).  Test transport: host-blocking staging copies."""

    def __init__(self, rank, size, group_key):
        self.rank, self.size = int(rank), int(size)
        self._h = C.c_void_p()
        _lib.call("ocn_comm_init_local", C.byref(self._h), self.rank, self.size, int(group_key))
        self._scratch = None


class ReplicaFabric(RcclFabric):
    """Rank 0 of `size` IDENTICAL ranks (ocn_comm_init_replica): every receive is a device copy from this rank's own send buffer to
    the mirror-image peer.  Times what one rank of an R-rank run costs before any link time, through the library's R-rank schedules,
    kernels and C drivers (tools/bench_dist_rank.py); not a way to run a model."""

    def __init__(self, size):
        self.rank, self.size = 0, int(size)
        self._h = C.c_void_p()
        _lib.call("ocn_comm_init_replica", C.byref(self._h), self.size)
        self._scratch = None


def make_distributed(rank=None, world_size=None, local_rank=None, force_communication=None):
    """Distributed(GPU(); partition = Partition(world_size)) with the RCCL transport of libocn_hip.  One process per GPU, started by
    torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment).  A gloo process group
    carries the unique id (and nothing else in the time-stepping loop)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world_size)

    def bootstrap(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    fabric = RcclFabric(rank, world_size, bootstrap)
    return Distributed(GPU(), partition=Partition(world_size), fabric=fabric, force_communication=force_communication)


# --------------------------------------------------------------------------------------------------
# Ops: the device kernels the choreography calls (HIP; tests may inject CPU restatements).
# --------------------------------------------------------------------------------------------------
class HipOps:
    name = "hip"

    def new_buffer(self, arch, n):
        return torch.zeros(n, dtype=torch.float64, device=device(arch.child_architecture))

    def local_fill(self, grid, fields, fbnv):
        from .fields import local_fill_halo_regions
        local_fill_halo_regions(grid, fields, fbnv)

    def pack_x(self, grid, f, west, east):
        _lib.call("ocn_halo_pack_x", grid.cref, f.ptr, f.loc, west.data_ptr(), east.data_ptr(), stream_ptr())

    def unpack_x(self, grid, f, west, east):
        _lib.call("ocn_halo_unpack_x", grid.cref, f.ptr, f.loc, west.data_ptr(), east.data_ptr(), stream_ptr())

    def plane_x(self, grid, f, which, buf, unpack):
        _lib.call("ocn_halo_plane_x", grid.cref, f.ptr, f.loc, int(which), buf.data_ptr(), int(bool(unpack)), stream_ptr())

    def pack_pressure(self, grid, p, u, dt_correct, west, east):
        _lib.call("ocn_halo_pack_pressure", grid.cref, p.ptr, u.ptr, float(dt_correct), west.data_ptr(), east.data_ptr(), stream_ptr())

    def unpack_pressure(self, grid, p, u, west, east):
        _lib.call("ocn_halo_unpack_pressure", grid.cref, p.ptr, u.ptr, west.data_ptr(), east.data_ptr(), stream_ptr())

    def pack_x_fields(self, grid, fields, west, east):
        _lib.call("ocn_halo_pack_x_fields", grid.cref, _lib.ptr_array([f.ptr for f in fields]), _lib.i32_array([f.loc for f in fields]),
                  len(fields), west.data_ptr(), east.data_ptr(), stream_ptr())

    def unpack_x_fields(self, grid, fields, west, east):
        _lib.call("ocn_halo_unpack_x_fields", grid.cref, _lib.ptr_array([f.ptr for f in fields]), _lib.i32_array([f.loc for f in fields]),
                  len(fields), west.data_ptr(), east.data_ptr(), stream_ptr())

    def sync(self):
        torch.cuda.current_stream().synchronize()

    def make_dist_poisson(self, grid, arch):
        return _HipDistPoisson(grid, arch)


class Distributed:
    """Distributed(child_architecture; partition): one rank of an x-slab decomposition."""

    def __init__(self, child_architecture=None, partition=None, fabric=None, ops=None, force_communication=None):
        """force_communication (default: the environment variable OCN_FORCE_DISTRIBUTED == "1"): with ONE rank, still treat x as a
        partitioned (FullyConnected) direction and run every exchange through the transport -- a rank then sends its strips and
        transposes to itself.  This is how the RCCL path is exercised on a one-GPU box."""
        self.force_communication = (os.environ.get("OCN_FORCE_DISTRIBUTED") == "1") if force_communication is None else bool(force_communication)
        self.child_architecture = child_architecture if child_architecture is not None else GPU()
        self.fabric = fabric if fabric is not None else TorchDistributedFabric()
        self.ops = ops if ops is not None else HipOps()
        self.partition = partition if partition is not None else Partition(self.fabric.size)
        if self.partition.x != self.fabric.size:
            raise ValueError(f"Partition({self.partition.x}) does not match the number of ranks {self.fabric.size}")
        self.local_rank = self.fabric.rank
        self.ranks = (self.partition.x, 1, 1)
        R = self.partition.x
        # neighbours wrap around (distributed_architectures.jl:386-429)
        self.west_rank = (self.local_rank - 1) % R
        self.east_rank = (self.local_rank + 1) % R
        self._buffers = {}
        self._pending = None

    @property
    def device(self):
        return device(self.child_architecture)

    @property
    def communicates(self):
        """x halos come from the transport (R > 1, or one rank talking to itself when force_communication is set)"""
        return self.partition.x > 1 or self.force_communication

    def __repr__(self):
        return f"Distributed({self.child_architecture}, rank {self.local_rank} of {self.partition.x})"

    # ---- halo communication ------------------------------------------------------------------
    def _halo_buffers(self, fields):
        """(send_w, send_e, recv_w, recv_e) for the whole tuple -- the strips of the fields follow one another, so that the exchange
        is one message per neighbour -- and each field's (offset, length) inside them."""
        key = tuple((f.ptr, f.loc) for f in fields)
        b = self._buffers.get(key)
        if b is None:
            g = fields[0].grid
            spans, off = [], 0
            for f in fields:
                sx, sy, sz = g.parent_shape(f.loc)
                n = g.Hx * sy * sz  # OneDBuffers: full cross-section, corners travel with the sides (:70-75)
                spans.append((off, n))
                off += n
            b = (tuple(self.ops.new_buffer(self, off) for _ in range(4)), tuple(spans))
            self._buffers[key] = b
        return b

    def start_halo_exchange(self, fields):
        """Pack + post the x exchange for a tuple of fields (fill_halo_event! with async=true)."""
        g = fields[0].grid
        if not self.communicates:
            return None
        self.finish_halo_exchange()  # an exchange left in flight by a deferred update_state! completes first
        fields = tuple(fields)
        if hasattr(self.fabric, "halo_exchange_begin"):  # RCCL behind the C ABI: pack + grouped send / recv in the library
            self.fabric.halo_exchange_begin(g, fields)
            self._pending = (None, fields)
            return self._pending
        (sw, se, rw, re), spans = self._halo_buffers(fields)
        if hasattr(self.ops, "pack_x_fields"):
            self.ops.pack_x_fields(g, fields, sw, se)
        else:
            for f, (o, n) in zip(fields, spans):
                self.ops.pack_x(g, f, sw[o:o + n], se[o:o + n])
        # (the reference calls sync_device! before posting MPI messages, halo_communication.jl:272, 303; torch.distributed
        #  collectives are ordered after the current stream's work, so no host synchronisation is needed here)
        # my west strip becomes the west neighbour's east halo, and vice versa.  With R = 2 both neighbours are the same
        # peer: receives are then posted in the order (from east, from west) so that they pair up with the peer's
        # (west, east) sends.
        sends = [(sw, self.west_rank), (se, self.east_rank)]
        recvs = [(rw, self.west_rank), (re, self.east_rank)] if self.partition.x > 2 else [(re, self.east_rank), (rw, self.west_rank)]
        self._pending = (self.fabric.start_exchange(sends, recvs), fields)
        return self._pending

    def finish_halo_exchange(self):
        """synchronize_communication! (distributed_fields.jl:58-75): wait + unpack."""
        if self._pending is None:
            return
        reqs, fields = self._pending
        g = fields[0].grid
        if reqs is None:
            self.fabric.halo_exchange_end(g, fields)
            self._pending = None
            return
        self.fabric.wait(reqs)
        (sw, se, rw, re), spans = self._halo_buffers(fields)
        if hasattr(self.ops, "unpack_x_fields"):
            self.ops.unpack_x_fields(g, fields, rw, re)
        else:
            for f, (o, n) in zip(fields, spans):
                self.ops.unpack_x(g, f, rw[o:o + n], re[o:o + n])
        self._pending = None

    def fill_neighbour_plane(self, fields, f, side):
        """Local (y, z) fills of `fields`, then ONE x-plane of `f` from a neighbour instead of the 2 Hx strips of every field:
        side = "east": f[nx+1, :, :] <- east neighbour's f[1, :, :] (what divᶜᶜᶜ reads of u); side = "west": f[0, :, :] <- west
        neighbour's f[nx, :, :] (what ∂xᶠᶜᶜ reads of the pressure).  For the two synchronous fills inside the pressure projection,
        whose fields get their complete exchange in the following update_state!."""
        g = f.grid
        self.finish_halo_exchange()
        self.ops.local_fill(g, fields, True)
        if not self.communicates or not hasattr(self.ops, "plane_x"):
            return self.fill_halo_regions(fields) if self.communicates else None
        if hasattr(self.fabric, "halo_exchange_plane"):
            return self.fabric.halo_exchange_plane(g, f, side)
        sx, sy, sz = g.parent_shape(f.loc)
        key = ("plane", f.loc)
        b = self._buffers.get(key)
        if b is None:
            b = self._buffers[key] = (self.ops.new_buffer(self, sy * sz), self.ops.new_buffer(self, sy * sz))
        sbuf, rbuf = b
        east = side == "east"
        # the plane I need from the east is my east neighbour's WEST interior plane, and vice versa
        self.ops.plane_x(g, f, 0 if east else 1, sbuf, False)
        reqs = self.fabric.start_exchange([(sbuf, self.west_rank if east else self.east_rank)],
                                          [(rbuf, self.east_rank if east else self.west_rank)])
        self.fabric.wait(reqs)
        self.ops.plane_x(g, f, 1 if east else 0, rbuf, True)

    def exchange_strips(self, send_west, send_east, recv_west, recv_east):
        """One contiguous strip per x neighbour, synchronous in stream order: my west strip arrives as the west neighbour's recv_east.
        (The wide halos of the split-explicit substepping, hydrostatic.py.)"""
        self.finish_halo_exchange()  # (both transports: a deferred end-of-step exchange may still be in flight)
        if hasattr(self.fabric, "exchange_strips"):
            return self.fabric.exchange_strips(send_west, send_east, recv_west, recv_east)
        sends = [(send_west, self.west_rank), (send_east, self.east_rank)]
        recvs = ([(recv_west, self.west_rank), (recv_east, self.east_rank)] if self.partition.x > 2
                 else [(recv_east, self.east_rank), (recv_west, self.west_rank)])
        self.fabric.wait(self.fabric.start_exchange(sends, recvs))

    def barrier(self):
        """MPI.Barrier for users of a Distributed architecture: completes a halo exchange that a deferred update_state! left in
        flight, then the transport's barrier (RcclFabric.barrier alone refuses while an exchange is pending)."""
        self.finish_halo_exchange()
        b = getattr(self.fabric, "barrier", None)
        if b is not None:
            b()

    # ---- the correction-on-load stage of a slab ---------------------------------------------------------------
    def exchange_plane(self, f, side):
        """ONE x plane of f from a neighbour, nothing else (fill_neighbour_plane without its local fills)"""
        g = f.grid
        if hasattr(self.fabric, "halo_exchange_plane"):
            return self.fabric.halo_exchange_plane(g, f, side)
        sx, sy, sz = g.parent_shape(f.loc)
        key = ("plane", f.loc)
        b = self._buffers.get(key)
        if b is None:
            b = self._buffers[key] = (self.ops.new_buffer(self, sy * sz), self.ops.new_buffer(self, sy * sz))
        sbuf, rbuf = b
        east = side == "east"
        self.ops.plane_x(g, f, 0 if east else 1, sbuf, False)
        reqs = self.fabric.start_exchange([(sbuf, self.west_rank if east else self.east_rank)],
                                          [(rbuf, self.east_rank if east else self.west_rank)])
        self.fabric.wait(reqs)
        self.ops.plane_x(g, f, 1 if east else 0, rbuf, True)

    def exchange_pressure(self, p, u, dt_correct):
        """The x-halo payload of the correction-on-load stage (include/ocn_hip.h, ocn_halo_exchange_pressure): Hx pressure planes per
        side + the owner-corrected u plane for the westmost halo column; in stream order."""
        g = p.grid
        if hasattr(self.fabric, "halo_exchange_pressure"):
            return self.fabric.halo_exchange_pressure(g, p, u, dt_correct)
        sx, sy, sz = g.parent_shape(p.loc)
        b = self._buffers.get("pressure")
        if b is None:
            b = self._buffers["pressure"] = tuple(self.ops.new_buffer(self, (g.Hx + 1) * sy * sz) for _ in range(4))
        sw, se, rw, re = b
        self.ops.pack_pressure(g, p, u, dt_correct, sw, se)
        sends = [(sw, self.west_rank), (se, self.east_rank)]
        recvs = [(rw, self.west_rank), (re, self.east_rank)] if self.partition.x > 2 else [(re, self.east_rank), (rw, self.west_rank)]
        self.fabric.wait(self.fabric.start_exchange(sends, recvs))
        self.ops.unpack_pressure(g, p, u, rw, re)

    def correct_on_load_supported(self, model):
        """One full-slab tendency launch per stage that applies the previous stage's pressure correction on load -- no 3-wide buffer
        strips, no separate pressure_correct launch, no post-solve exchange of three velocity fields (OCN_DIST_CORRECT_ON_LOAD=0
        keeps the interior / strip split of interleave_communication_and_computation.jl:29-67)."""
        g = model.grid
        return (self.communicates and os.environ.get("OCN_DIST_CORRECT_ON_LOAD", "1") != "0" and getattr(self.ops, "name", "") == "hip"
                and model.fuse_stage_boundaries and not model._general_fused
                and g.topology[1] == Periodic and g.topology[2] == Periodic
                and g.Nx >= max(16, g.Hx + 1) and g.Ny >= 8 and g.Nz >= 4)

    def project_and_advance(self, model, dt, stage_dt, gamma_next, zeta_next):
        """Everything between two RK3 substeps on a slab (runge_kutta_3.jl:103-118), re-cut so that the exchange of the UNCORRECTED u*, v*,
        w* flies under the pressure solve and only pressure planes move after it:
          local y / z fills of u*, v*, w* -> u*[nx+1] plane (the divergence needs it) -> strips of u*, v*, w* in flight ->
          solve_for_pressure! -> strips unpacked -> p planes (+ the owner-corrected u plane) -> ONE launch over the whole slab:
          correction on load + tendencies + the next substep.
        Results identical to pressure_correct_velocities! + update_state! + rk3_substep! (bit for bit in strict math)."""
        from . import models
        g = model.grid
        vel = tuple(model.velocities)
        self.finish_halo_exchange()
        self.ops.local_fill(g, vel, True)
        self.exchange_plane(model.u, "east")
        self.start_halo_exchange(vel)
        models.solve_for_pressure(model.pNHS, model.pressure_solver, stage_dt, vel)
        self.finish_halo_exchange()
        self.exchange_pressure(model.pNHS, model.u, stage_dt)
        models.cache_previous_tendencies(model)
        models.update_state_and_rk3_substep(model, dt, gamma_next, zeta_next, fill_halos=False, p_correct=model.pNHS, dt_correct=stage_dt)

    def fill_halo_regions(self, fields, fbnv=True):
        """Local (y, z) fills first, communication last (fill_halo_regions.jl:148-196); synchronous."""
        g = fields[0].grid
        self.ops.local_fill(g, fields, fbnv)
        if self.start_halo_exchange(fields) is not None:
            self.finish_halo_exchange()

    # ---- update_state! with interior / buffer overlap -------------------------------------------
    def update_state(self, model, compute_tendencies=True, defer_exchange=False):
        from . import models
        g = model.grid
        fields = model.prognostic_fields()
        if (defer_exchange and not compute_tendencies and self.communicates and not getattr(model, "general_terms", False)
                and os.environ.get("OCN_DIST_DEFER_EXCHANGE", "1") != "0"):
            # end of a step whose last tendency launch is deferred: start the exchange and leave it in flight; the next step's
            # fused launch (update_state_fused with fill_halos=False) overlaps it with its interior range, flush_tendencies and
            # every new exchange complete it first
            self.ops.local_fill(g, fields, False)
            self.start_halo_exchange(fields)
            return
        if getattr(model, "general_terms", False):
            # Coriolis / closure / buoyancy / boundary conditions
            if compute_tendencies:
                self.update_state_general(model, lambda rng=None: models.compute_tendencies_(model, rng, boundary_contributions=False))
                models.compute_boundary_tendency_contributions(model)
            else:
                self.fill_halo_regions(fields, False)
                models.compute_auxiliaries(model)
            return
        self.ops.local_fill(g, fields, False)
        pending = self.start_halo_exchange(fields)
        nx, Hx = g.Nx, g.Hx
        if compute_tendencies:
            if pending is None:
                models.compute_tendencies_(model)
            else:  # interior_tendency_kernel_parameters (interleave_communication_and_computation.jl:29-67)
                if nx - 2 * Hx >= 1:
                    models.compute_tendencies_(model, (Hx + 1, nx - Hx, 1, g.Ny, 1, g.Nz))
        self.finish_halo_exchange()
        if compute_tendencies and pending is not None:
            # buffer_tendency_kernel_parameters (compute_nonhydrostatic_buffer_tendencies.jl:28-39)
            w1 = min(Hx, nx)
            e0 = max(nx - Hx + 1, w1 + 1)
            self._two_strips(lambda rng: models.compute_tendencies_(model, rng), (1, w1, 1, g.Ny, 1, g.Nz),
                             (e0, nx, 1, g.Ny, 1, g.Nz) if e0 <= nx else None)

    def update_state_general(self, model, launch):
        """update_state! of a model with the extra terms (hydrostatic pressure anomaly, eddy diffusivities): the exchange of the
        prognostic fields overlaps the auxiliaries and the tendencies of the columns that do not read x-halos; the edge and
        halo columns of the auxiliaries are computed from the exchanged halos afterwards -- the buffer recomputation of
        compute_nonhydrostatic_buffer_tendencies.jl:55-68 instead of a second exchange (the halo values of νₑ, κₑ equal what
        the neighbour computes for its own edge column: same inputs, same arithmetic) -- followed by the two buffer strips.
        `launch(rng)` runs the tendency kernels of the i, j, k range `rng` (None = the whole slab)."""
        from . import models
        g = model.grid
        fields = model.prognostic_fields()
        nx, Hx = g.Nx, g.Hx
        d = model.diffusivity_fields
        aux = (() if d is None else (d["nu_e"],) + tuple(d["kappa_e"]))
        split = (self.communicates and nx - 2 * Hx >= 1 and Hx >= 2 and len(model.tracers) <= 4
                 and os.environ.get("OCN_DIST_GENERAL_OVERLAP", "1") != "0")
        if not split:
            self.fill_halo_regions(fields, False)
            models.compute_auxiliaries(model)
            launch(None)
            return
        self.ops.local_fill(g, fields, False)
        self.start_halo_exchange(fields)
        # interior: νₑ, κₑ of columns 2..nx-1 read u, v, w, c at i-1..i+1 (local); pHY′ of a column reads that column only
        models.compute_diffusivities(model, (2, nx - 1))
        models.update_hydrostatic_pressure(model, (1, nx))
        if aux:
            self.ops.local_fill(g, aux, True)  # their y / z halos (and bottom / top conditions)
        launch((Hx + 1, nx - Hx, 1, g.Ny, 1, g.Nz))
        self.finish_halo_exchange()
        for rng in ((0, 1), (nx, nx + 1)):
            models.compute_diffusivities(model, rng)
        for rng in ((0, 0), (nx + 1, nx + 1)):
            models.update_hydrostatic_pressure(model, rng)
        if aux:
            self.ops.local_fill(g, aux, True)
        w1 = min(Hx, nx)
        e0 = max(nx - Hx + 1, w1 + 1)
        self._two_strips(launch, (1, w1, 1, g.Ny, 1, g.Nz), (e0, nx, 1, g.Ny, 1, g.Nz) if e0 <= nx else None)

    def update_state_fused(self, model, launch, fill_halos=True):
        """update_state! + the next rk3 substep with the fused launch: same interior / buffer split and overlap as update_state."""
        g = model.grid
        fields = model.prognostic_fields()
        pending = self._pending  # (an exchange started at the end of the previous step, when fill_halos is False)
        if fill_halos:
            self.ops.local_fill(g, fields, False)
            pending = self.start_halo_exchange(fields)
        nx, Hx = g.Nx, g.Hx
        if pending is None:
            launch()
            return
        if getattr(model, "dist_correct_on_load", False) and os.environ.get("OCN_DIST_STRIPS", "0") == "0":
            # no buffer strips: one launch over the whole slab once the halos are in (the 3-wide strips cost more than the overlap buys)
            self.finish_halo_exchange()
            launch()
            return
        if nx - 2 * Hx >= 1:
            launch((Hx + 1, nx - Hx, 1, g.Ny, 1, g.Nz))
        self.finish_halo_exchange()
        w1 = min(Hx, nx)
        e0 = max(nx - Hx + 1, w1 + 1)
        self._two_strips(launch, (1, w1, 1, g.Ny, 1, g.Nz), (e0, nx, 1, g.Ny, 1, g.Nz) if e0 <= nx else None)

    def _two_strips(self, launch, west, east):
        """The west and east buffer strips (compute_nonhydrostatic_buffer_tendencies.jl:28-39) are independent and each too small to
        fill the chip (3 x Ny x Nz cells): the east one runs on a side stream next to the west one.  Ordered by events against the
        compute stream on both sides; OCN_DIST_CONCURRENT_STRIPS=0 runs them one after the other."""
        if east is None:
            return launch(west)
        if not torch.cuda.is_available() or os.environ.get("OCN_DIST_CONCURRENT_STRIPS", "1") == "0" or getattr(self.ops, "name", "") != "hip":
            launch(west)
            return launch(east)
        cur = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream()
            self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
        self._fork.record(cur)
        self._side_stream.wait_event(self._fork)
        with torch.cuda.stream(self._side_stream):
            launch(east)
            self._join.record(self._side_stream)
        launch(west)
        cur.wait_event(self._join)

    def pressure_solver(self, grid):
        if grid.topology[2] == Bounded:
            return DistributedFourierTridiagonalPoissonSolver(grid)
        return DistributedFFTBasedPoissonSolver(grid)


def distributed_rectilinear_grid(arch, size, x=None, y=None, z=None, topology=(Periodic, Periodic, Periodic), halo=None):
    """RectilinearGrid(arch::Distributed; ...): the rank-local portion (distributed_grids.jl:75-118)."""
    R, r = arch.partition.x, arch.local_rank
    Nx = size[0]
    if topology[0] not in (Periodic, Bounded):
        raise NotImplementedError("the partitioned x direction must be Periodic or Bounded")
    if topology[0] == Bounded and (tuple(topology[1:]) != (Bounded, Bounded) or (R == 1 and arch.communicates)):
        # (distributed_fft_based_poisson_solver.jl:62-66: if y is Periodic, so must x be; if z is Periodic, so must y and x be)
        raise NotImplementedError("a Bounded partitioned x needs (Bounded, Bounded, Bounded) and more than one rank")
    if Nx % R:
        raise ValueError(f"Nx = {Nx} must be divisible by the number of ranks {R} (equal slabs)")
    nx = Nx // R
    if R > 1 or arch.communicates:
        # partition_coordinate(c::Tuple, ...) (partition_assemble.jl:63-76), same fp64 arithmetic
        dl = (float(x[1]) - float(x[0])) / Nx
        lo = float(x[0])
        for _ in range(r):
            lo = lo + dl * nx
        xl = (lo, lo + dl * nx)
        # insert_connected_topology (distributed_grids.jl:339-346): a Bounded x leaves a wall on the first and on the last slab
        tx = FullyConnected if (topology[0] == Periodic or 0 < r < R - 1) else (RightConnected if r == 0 else LeftConnected)
        topo = (tx,) + tuple(topology[1:])
    else:
        xl, topo = x, tuple(topology)
    g = RectilinearGrid(arch, (nx,) + tuple(size[1:]), x=xl, y=y, z=z, topology=topo, halo=halo, _local=True)
    g.global_size = tuple(size)
    g.global_topology = tuple(topology)
    from fractions import Fraction
    g.global_Lx = float(Fraction(float(x[1])) - Fraction(float(x[0])))
    return g


class _HipDistPoisson:
    """Device side of the distributed solver: ocn_dist_poisson_* handle of libocn_hip."""

    def __init__(self, grid, arch):
        self.grid, self.R = grid, arch.partition.x
        self._h = C.c_void_p()
        gtx = _lib.OCN_BOUNDED if getattr(grid, "global_topology", (Periodic,))[0] == Bounded else _lib.OCN_PERIODIC
        _lib.call("ocn_dist_poisson_create_global", C.byref(self._h), grid.cref, arch.local_rank, self.R,
                  C.c_double(getattr(grid, "global_Lx", grid.Lx * self.R)), gtx)
        ptrs = [C.c_void_p() for _ in range(4)]
        _lib.call("ocn_dist_poisson_buffers", self._h, *[C.byref(p) for p in ptrs])
        nyt, nel, r2c = C.c_int32(), C.c_int64(), C.c_int32()
        _lib.call("ocn_dist_poisson_layout", self._h, C.byref(nyt), C.byref(nel), C.byref(r2c))
        self.nyt, self.r2c = nyt.value, bool(r2c.value)  # y extent of the transposed data (padded half spectrum if r2c)
        # every rank must use the same transposed extent: if any rank fell back to complex plans (rocFFT plan self test,
        # csrc/poisson.hip), all ranks do
        reduce = getattr(arch.fabric, "allreduce_max", None)
        if reduce is not None and self.R > 1 and grid.topology[2] != Bounded:
            flag = torch.tensor([0.0 if self.r2c else 1.0], dtype=torch.float64, device=arch.device)
            if float(reduce(flag).item()) > 0 and self.r2c:
                import os
                _lib.lib().ocn_dist_poisson_destroy(self._h)
                old_env = os.environ.get("OCN_POISSON_C2C")
                os.environ["OCN_POISSON_C2C"] = "1"
                try:
                    self._h = C.c_void_p()
                    _lib.call("ocn_dist_poisson_create", C.byref(self._h), grid.cref, arch.local_rank, self.R,
                              C.c_double(getattr(grid, "global_Lx", grid.Lx * self.R)))
                finally:
                    if old_env is None:
                        os.environ.pop("OCN_POISSON_C2C", None)
                    else:
                        os.environ["OCN_POISSON_C2C"] = old_env
                _lib.call("ocn_dist_poisson_buffers", self._h, *[C.byref(p) for p in ptrs])
                _lib.call("ocn_dist_poisson_layout", self._h, C.byref(nyt), C.byref(nel), C.byref(r2c))
                self.nyt, self.r2c = nyt.value, bool(r2c.value)
        fast = C.c_int32()
        _lib.call("ocn_dist_poisson_pipeline", self._h, C.byref(fast))
        self.fast = fast.value  # slab pipeline: no pack / unpack passes (1: periodic z, 2: tridiagonal flavour; ocn_hip.h)
        n = nel.value * 2
        # wrap the library-owned exchange buffers as tensors (no copy) so torch.distributed can move them
        if self.fast == 3:  # transpose-free pipeline: the only exchange is an all-gather of the interface values
            gp, per_rank = [C.c_void_p(), C.c_void_p()], C.c_int64()
            _lib.call("ocn_dist_poisson_gather_buffers", self._h, C.byref(gp[0]), C.byref(gp[1]), C.byref(per_rank))
            self.gsend = _wrap_device_buffer(gp[0].value, per_rank.value, arch.device)
            self.grecv = _wrap_device_buffer(gp[1].value, per_rank.value * self.R, arch.device)
            self.send = None
        else:
            self.send = _wrap_device_buffer(ptrs[2].value, n, arch.device)
        self.recv = _wrap_device_buffer(ptrs[3].value, n, arch.device)
        self.yfield_ptr, self.xfield_ptr = ptrs[0].value, ptrs[1].value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ocn_dist_poisson_destroy(h)
            except Exception:
                pass
            self._h = None

    def _dims(self):
        g = self.grid
        return g.Nx, self.nyt, g.Nz, self.R

    def source_term(self, u, v, w, dt):
        _lib.call("ocn_dist_poisson_source_term", self._h, u.ptr, v.ptr, w.ptr, float(dt), stream_ptr())

    def forward_yz(self):
        _lib.call("ocn_dist_poisson_forward_yz", self._h, stream_ptr())

    def pack_y_to_x(self):
        _lib.call("ocn_transpose_pack_y_to_x", *self._dims(), self.yfield_ptr, self.send.data_ptr(), stream_ptr())

    def unpack_x_from_y(self, rbuf):
        _lib.call("ocn_transpose_unpack_x_from_y", *self._dims(), rbuf.data_ptr(), self.xfield_ptr, stream_ptr())

    def solve_x(self):
        _lib.call("ocn_dist_poisson_solve_x", self._h, stream_ptr())

    def pack_x_to_y(self):
        _lib.call("ocn_transpose_pack_x_to_y", *self._dims(), self.xfield_ptr, self.send.data_ptr(), stream_ptr())

    def unpack_y_from_x(self, rbuf):
        _lib.call("ocn_transpose_unpack_y_from_x", *self._dims(), rbuf.data_ptr(), self.yfield_ptr, stream_ptr())

    def backward_yz(self, p):
        _lib.call("ocn_dist_poisson_backward_yz", self._h, p.ptr, stream_ptr())


class DistributedFFTBasedPoissonSolver:
    """DistributedFFTBasedPoissonSolver for slab-x (Ry == 1), solve! (distributed_fft_based_poisson_solver.jl:141-178):
    FFT_z, FFT_y local -> y->x all-to-all -> FFT_x -> divide by (λx+λy+λz), rank 0 zeroes mode (1,1,1) -> IFFT_x ->
    x->y all-to-all -> IFFT_y, IFFT_z -> real part into the local pressure interior."""

    def __init__(self, grid):
        arch = grid.architecture
        self.grid, self.arch = grid, arch
        self.R = arch.partition.x
        self._check_topology(grid)
        if grid.Ny % self.R:
            raise ValueError(f"Ny = {grid.Ny} must be divisible by Rx = {self.R}")  # :211-229
        self.impl = arch.ops.make_dist_poisson(grid, arch)

    @staticmethod
    def _check_topology(grid):
        if grid.topology[1] != Periodic or grid.topology[2] != Periodic:
            raise NotImplementedError("DistributedFFTBasedPoissonSolver: (x-partitioned, Periodic, Periodic) only")

    def compute_source_term(self, u, v, w, dt):
        self.impl.source_term(u, v, w, dt)

    def _all_to_all(self):
        """transpose_*!: pack -> sync_device! -> Alltoallv! -> unpack (distributed_transpose.jl:185-191)"""
        if self.R == 1:
            return self.impl.send
        self.arch.fabric.all_to_all(self.impl.recv, self.impl.send)
        return self.impl.recv

    def _all_gather(self, recv, send):
        ag = getattr(self.arch.fabric, "all_gather", None)
        if ag is not None:
            return ag(recv, send)
        # a fabric without an all-gather: every peer's chunk of an all-to-all is my whole message
        self.arch.fabric.all_to_all(recv, send.repeat(self.R))

    def solve(self, p):
        impl = self.impl
        ex = getattr(self.arch.fabric, "dist_poisson_exchange", None)
        if getattr(impl, "fast", 0) == 3:
            # transpose-free pipeline (csrc/xtri.hip): y, z transforms and the local x blocks of the cyclic tridiagonal systems, ONE
            # all-gather of two numbers per mode, interface systems + correction, inverse transforms
            impl.forward_yz()
            if ex is not None and hasattr(impl, "_h") and self.arch.communicates:
                ex(impl._h, 0)
            elif self.R == 1:
                impl.grecv.copy_(impl.gsend)
            else:
                self._all_gather(impl.grecv, impl.gsend)
            impl.solve_x()
            impl.backward_yz(p)
            return p
        if ex is not None and hasattr(impl, "_h") and self.arch.communicates:
            # the product path: both transposes are ocn_dist_poisson_exchange (grouped ncclSend / ncclRecv in the library)
            impl.forward_yz()
            if not getattr(impl, "fast", False):
                impl.pack_y_to_x()
            ex(impl._h, 0)
            if not getattr(impl, "fast", False):
                impl.unpack_x_from_y(impl.recv)
            impl.solve_x()
            if not getattr(impl, "fast", False):
                impl.pack_x_to_y()
            ex(impl._h, 1)
            if not getattr(impl, "fast", False):
                impl.unpack_y_from_x(impl.recv)
            impl.backward_yz(p)
            return p
        if getattr(impl, "fast", False):
            # slab pipeline of libocn_hip: the transforms read / write the exchange layout themselves
            impl.forward_yz()
            self.arch.fabric.all_to_all(impl.recv, impl.send)
            impl.solve_x()
            if impl.fast == 2:  # the tridiagonal flavour leaves its solution in `send`
                self.arch.fabric.all_to_all(impl.recv, impl.send)
            else:
                self.arch.fabric.all_to_all(impl.send, impl.recv)
            impl.backward_yz(p)
            return p
        impl.forward_yz()
        impl.pack_y_to_x()
        impl.unpack_x_from_y(self._all_to_all())
        impl.solve_x()
        impl.pack_x_to_y()
        impl.unpack_y_from_x(self._all_to_all())
        impl.backward_yz(p)
        return p


class DistributedFourierTridiagonalPoissonSolver(DistributedFFTBasedPoissonSolver):
    """DistributedFourierTridiagonalPoissonSolver, ZStretched flavour, for slab-x and a Bounded (regular or stretched) z
    (distributed_fft_tridiagonal_solver.jl:149-292): FFT_y local -> y->x all-to-all -> FFT_x -> batched Thomas sweep in z
    with this rank's ky range -> IFFT_x -> x->y all-to-all -> IFFT_y -> real part.  z stays local in both layouts, so the
    reference's additional transposes to a z-local pencil are not needed; the same `solve` choreography as the FFT solver
    applies with the library's handle in its tridiagonal mode.  The (kx, ky) = (0, 0) column gets the zero-mean gauge of the
    single-process solver (the reference's distributed solver leaves that constant undetermined).  A Bounded y (the channel:
    (Periodic, Bounded, Bounded)) takes the cosine transforms of the single-process solver in place of FFT_y / IFFT_y."""

    @staticmethod
    def _check_topology(grid):
        if grid.topology[1] not in (Periodic, Bounded) or grid.topology[2] != Bounded:
            raise NotImplementedError("DistributedFourierTridiagonalPoissonSolver: (x-partitioned, Periodic or Bounded, Bounded) only")


class _DevBuf:
    """__cuda_array_interface__ view of a library-owned device allocation."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _wrap_device_buffer(ptr, n, dev):
    return torch.as_tensor(_DevBuf(ptr, n), device=dev)
