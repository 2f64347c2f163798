"""Architectures: mirrors src/Architectures.jl (CPU, GPU, device, on_architecture, zeros, sync_device!).

`GPU()` is the MI355X device of this process.  Device memory, streams and (for Distributed)
collectives are provided by PyTorch-ROCm; every kernel is in libocn_hip.so.
"""
import numpy as np
import torch


class AbstractArchitecture:
    pass


class CPU(AbstractArchitecture):
    """Present for API parity only: this package *is* the GPU backend (the CPU path is the reference's)."""

    def __init__(self):
        raise NotImplementedError("oceananigans.jl_amd implements the GPU() architecture only; "
                                  "run the CPU() path with the reference Oceananigans.jl")


class GPU(AbstractArchitecture):
    """GPU(device): src/Architectures.jl:33-46."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GPU() requires a visible MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)

    def __repr__(self):
        return f"GPU({self.device})"


def device(arch):
    return arch.device


def child_architecture(arch):
    return getattr(arch, "child_architecture", arch)


def stream_ptr():
    """hipStream_t of torch's current stream, as an integer usable for a void* argument."""
    return torch.cuda.current_stream().cuda_stream


def zeros(arch, shape):
    """zeros(arch, FT, N...) (src/Grids/zeros_and_ones.jl:9). `shape` is (sx, sy, sz) in the reference's
    column-major order; storage is a C-contiguous tensor of shape (sz, sy, sx) = the same bytes."""
    return torch.zeros(tuple(reversed(shape)), dtype=torch.float64, device=device(child_architecture(arch)))


def on_architecture(arch, a):
    """on_architecture(arch, array) (src/Architectures.jl:86-118): numpy -> device tensor, tensor -> tensor."""
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device(child_architecture(arch)))
    return a.to(device(child_architecture(arch)))


def sync_device(arch=None):
    """sync_device! (src/Utils/multi_region_transformation.jl:183-186)"""
    torch.cuda.current_stream().synchronize()
