"""Host-side helpers of the 2-D tile decomposition (one process per GPU): rank <-> tile
mapping, neighbour table and bootstrap of the library's communicator through
torch.distributed (RCCL on GPUs, gloo in the CPU tests).  Mirrors what the library does
internally (msom_create_tiled: rank r owns tile (r % px, r // px))."""
import ctypes

import numpy as np

TILE_GRIDS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}
# direction order of the library (comm.h): W E S N SW SE NW NE
DIRS = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1), "SW": (-1, -1), "SE": (1, -1), "NW": (-1, 1), "NE": (1, 1)}


def tile_grid(world):
    if world not in TILE_GRIDS:
        raise ValueError(f"no tile grid defined for {world} ranks (use 1, 2, 4 or 8)")
    return TILE_GRIDS[world]


def tile_of_rank(rank, px, py):
    return rank % px, rank // px


OPPOSITE = dict(W="E", E="W", S="N", N="S", SW="NE", NE="SW", SE="NW", NW="SE")


def neighbours(rank, px, py, periodic=False):
    """periodic (sbc = -1 on tiles, msqg/qg.h:842-846): the neighbours wrap around; with 1 or 2 tiles per side both
    neighbours of an axis are the same rank (or the tile itself)"""
    ix, iy = tile_of_rank(rank, px, py)
    out = {}
    for name, (dx, dy) in DIRS.items():
        jx, jy = ix + dx, iy + dy
        if periodic:
            out[name] = (jy % py) * px + jx % px
        else:
            out[name] = jy * px + jx if (0 <= jx < px and 0 <= jy < py) else -1
    return out


def walls(rank, px, py, periodic=False):
    ix, iy = tile_of_rank(rank, px, py)
    if periodic:
        return dict(W=False, E=False, S=False, N=False)
    return dict(W=ix == 0, E=ix == px - 1, S=iy == 0, N=iy == py - 1)


def exchange_order(names):
    """posting order of one exchange (comm.hip): sends in the order of the direction they travel in, receives in the order
    of the direction the INCOMING message travels in (the opposite of the edge), so that two messages between the same
    pair of ranks pair up"""
    order = list(DIRS)
    sends = sorted(names, key=order.index)
    recvs = sorted(names, key=lambda n: order.index(OPPOSITE[n]))
    return sends, recvs


def tile_slice(rank, px, py, nx, ny):
    """(slice_y, slice_x) of this rank's tile in a global [.., gny, gnx] array."""
    ix, iy = tile_of_rank(rank, px, py)
    return slice(iy * ny, (iy + 1) * ny), slice(ix * nx, (ix + 1) * nx)


def broadcast_unique_id(dist, make_id, device="cpu"):
    """Rank 0 creates the 128-byte communicator id (ncclUniqueId), everybody receives it."""
    import torch

    uid = torch.zeros(128, dtype=torch.uint8)
    if dist.get_rank() == 0:
        uid = torch.frombuffer(bytearray(make_id()), dtype=torch.uint8).clone()
    uid = uid.to(device)
    dist.broadcast(uid, 0)
    return uid.cpu().numpy().tobytes()


def rccl_unique_id(lib):
    buf = (ctypes.c_char * 128)()
    if lib.msom_comm_unique_id(buf) != 0:
        raise RuntimeError(lib.msom_last_error().decode())
    return buf.raw


def synthetic_tile(psi_fn, rank, px, py, nl, nx, ny):
    """Sample a global analytic field on this rank's tile: psi_fn(l, y01, x01) with normalised
    cell-centre coordinates of the GLOBAL domain."""
    ix, iy = tile_of_rank(rank, px, py)
    x = (np.arange(ix * nx, (ix + 1) * nx) + 0.5) / (nx * px)
    y = (np.arange(iy * ny, (iy + 1) * ny) + 0.5) / (ny * py)
    return np.stack([psi_fn(l, y, x) for l in range(nl)])
