"""Shared helpers for the parity tests: build the same grid / fields on the oracle and on the GPU."""
import numpy as np

TOPO_NAME = {"P": "Periodic", "B": "Bounded", "F": "Flat"}


def stretched_faces(Nz, Lz=1.0, power=1.6):
    """A smooth stretched-z face set (surface-refined), deterministic."""
    s = np.linspace(0.0, 1.0, Nz + 1)
    return -Lz * (1 - s) ** power


def make_pair(O, ocn, size, topo="PPP", x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), halo=(3, 3, 3)):
    """Returns (oracle grid, product grid) for the same RectilinearGrid."""
    og = O.Grid(size, x=x, y=y, z=z, topology=topo, halo=halo)
    nonflat = [d for d in range(3) if topo[d] != "F"]
    psize = tuple(size[d] for d in nonflat)
    phalo = tuple(halo[d] for d in nonflat)
    pg = ocn.RectilinearGrid(ocn.GPU(), size=psize, x=x, y=y, z=None if topo[2] == "F" else z,
                             topology=tuple(TOPO_NAME[t] for t in topo), halo=phalo)
    return og, pg


def random_parent(og, loc, rng, lo=-1.0, hi=1.0):
    """Random values everywhere in the parent array (halos included), F-ordered [i,j,k]."""
    a = og.zeros(loc)
    a[...] = rng.uniform(lo, hi, a.shape)
    return a


def to_dev(ocn, pg, loc, a):
    """oracle parent array [i,j,k] (F-order) -> product Field with identical bytes."""
    import torch
    f = ocn.Field(loc, pg)
    f.data.copy_(torch.from_numpy(np.ascontiguousarray(a.T)))
    return f


def from_dev(f):
    return f.data.cpu().numpy().T
