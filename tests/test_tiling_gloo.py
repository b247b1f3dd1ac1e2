"""world_size-2 / -4 `gloo` tests on CPU of the N > 1 path's host logic and exchange protocol:
rank <-> tile mapping, communicator-id bootstrap through torch.distributed, and the
red-black sweep protocol over tiles (colour by GLOBAL (i + j), wall ghosts lag, tile-edge
ghosts refreshed by a halo exchange after every colour half-sweep), checked bit for bit
against the single-domain oracle.  The HIP implementation of the same protocol is tested on
the GPU box (tests/test_gpu_tiled.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc  # noqa: E402
from msom_amd import tiling  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, px, py, tile, nl, nsweeps, outdir, periodic=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = tiling.broadcast_unique_id(dist, lambda: b"MSOMLOCL" + bytes(range(8)) + bytes(112))
        assert uid[:8] == b"MSOMLOCL" and uid[8:16] == bytes(range(8))
        nb, wl = tiling.neighbours(rank, px, py, periodic), tiling.walls(rank, px, py, periodic)
        gnx, gny = tile * px, tile * py
        rng = np.random.default_rng(0)
        da_g, res_g = rng.standard_normal((nl, gny, gnx)), rng.standard_normal((nl, gny, gnx))
        sy, sx = tiling.tile_slice(rank, px, py, tile, tile)
        # tile with one ghost ring
        a = np.zeros((nl, tile + 2, tile + 2))
        a[:, 1:-1, 1:-1] = da_g[:, sy, sx]
        b = res_g[:, sy, sx]
        o = orc.Oracle(orc.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else "") + ("sbc = -1\n" if periodic else "")), smoother=orc.GS_RB)
        o.set_const()
        D = 80.0 / gnx
        S = [(o.param(f"Fr_{l}") / o.param("Rom")) ** 2 for l in range(nl - 1)]
        idh0 = [o.param(f"idh0_{l}") for l in range(nl)]
        idh1 = [o.param(f"idh1_{l}") for l in range(nl)]

        def wall_ghosts():
            if wl["W"]: a[:, 1:-1, 0] = -a[:, 1:-1, 1]
            if wl["E"]: a[:, 1:-1, -1] = -a[:, 1:-1, -2]
            if wl["S"]: a[:, 0, 1:-1] = -a[:, 1, 1:-1]
            if wl["N"]: a[:, -1, 1:-1] = -a[:, -2, 1:-1]

        def exchange():
            edges = {"W": (a[:, 1:-1, 1], (slice(None), slice(1, -1), 0)), "E": (a[:, 1:-1, -2], (slice(None), slice(1, -1), -1)),
                     "S": (a[:, 1, 1:-1], (slice(None), 0, slice(1, -1))), "N": (a[:, -2, 1:-1], (slice(None), -1, slice(1, -1)))}
            names = [n for n in edges if nb[n] >= 0]
            sends, recvs = tiling.exchange_order(names)
            sbuf = {n: torch.from_numpy(np.ascontiguousarray(edges[n][0])) for n in names}
            rbuf = {n: torch.empty_like(sbuf[n]) for n in names}
            # a tile that is its own neighbour (one tile per side, periodic): the message of direction d lands at the opposite edge
            for n in names:
                if nb[n] == rank:
                    rbuf[tiling.OPPOSITE[n]] = sbuf[n].clone()
            ops = [dist.P2POp(dist.isend, sbuf[n], nb[n]) for n in sends if nb[n] != rank]
            ops += [dist.P2POp(dist.irecv, rbuf[n], nb[n]) for n in recvs if nb[n] != rank]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for n in names:
                a[edges[n][1]] = rbuf[n].numpy()

        def column(i, j):  # relax_layer column solve in the reference's operation order
            t0, t1, t2, rhs = np.zeros(nl), np.zeros(nl), np.zeros(nl), np.zeros(nl)
            for l in range(nl):
                rhs[l] = -(D * D) * b[l, j - 1, i - 1]
                t0[l] = -(D * D) * S[l - 1] * idh0[l] if l > 0 else 0.0
                t2[l] = -(D * D) * S[l] * idh1[l] if l < nl - 1 else 0.0
                t1[l] = -t2[l] if l == 0 else (-t0[l] - t2[l] if l < nl - 1 else -t0[l])
                rhs[l] += 1.0 * a[l, j, i + 1] + 1.0 * a[l, j, i - 1]
                t1[l] += 2.0
                rhs[l] += 1.0 * a[l, j + 1, i] + 1.0 * a[l, j - 1, i]
                t1[l] += 2.0
            for l in range(1, nl):
                t1[l] -= t0[l] * t2[l - 1] / t1[l - 1]
                rhs[l] -= t0[l] * rhs[l - 1] / t1[l - 1]
            x = np.zeros(nl)
            x[nl - 1] = rhs[nl - 1] / t1[nl - 1]
            for l in range(nl - 2, -1, -1):
                x[l] = (rhs[l] - t2[l] * x[l + 1]) / t1[l]
            return x

        ix, iy = tiling.tile_of_rank(rank, px, py)
        wall_ghosts()
        exchange()
        for _ in range(nsweeps):
            for c in (0, 1):
                new = {}
                for j in range(1, tile + 1):
                    for i in range(1, tile + 1):
                        gi, gj = ix * tile + i - 1, iy * tile + j - 1
                        if (gi + gj) % 2 == c:
                            new[(i, j)] = column(i, j)
                for (i, j), x in new.items():
                    a[:, j, i] = x
                exchange()          # tile-edge ghosts after every colour
            wall_ghosts()           # boundary_level(): wall ghosts lag within the sweep
        np.save(os.path.join(outdir, f"tile{rank}.npy"), a[:, 1:-1, 1:-1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("px,py", [(2, 1), (2, 2)])
def test_tiled_red_black_protocol_over_gloo(tmp_path, px, py):
    tile, nl, nsweeps = 8, 3, 2
    world = px * py
    mp.spawn(_worker, args=(world, _free_port(), px, py, tile, nl, nsweeps, str(tmp_path)), nprocs=world, join=True)
    gnx, gny = tile * px, tile * py
    rng = np.random.default_rng(0)
    da_g, res_g = rng.standard_normal((nl, gny, gnx)), rng.standard_normal((nl, gny, gnx))
    o = orc.Oracle(orc.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else "")), smoother=orc.GS_RB)
    o.set_const()
    ref = o.relax(0, da_g, res_g, nsweeps)
    got = np.zeros_like(ref)
    for r in range(world):
        sy, sx = tiling.tile_slice(r, px, py, tile, tile)
        got[:, sy, sx] = np.load(tmp_path / f"tile{r}.npy")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("px,py", [(2, 1), (2, 2)])
def test_periodic_tiling_protocol_over_gloo(tmp_path, px, py):
    """sbc = -1 on tiles: neighbours wrap around, no wall anywhere; with 2 tiles per side both neighbours of an axis are the
    SAME rank, with 1 the tile itself -- the posting order of tiling.exchange_order (the one comm.hip uses) keeps the two
    messages of a pair apart.  Bit for bit against the periodic single-domain oracle."""
    tile, nl, nsweeps = 8, 3, 2
    world = px * py
    mp.spawn(_worker, args=(world, _free_port(), px, py, tile, nl, nsweeps, str(tmp_path), True), nprocs=world, join=True)
    gnx, gny = tile * px, tile * py
    rng = np.random.default_rng(0)
    da_g, res_g = rng.standard_normal((nl, gny, gnx)), rng.standard_normal((nl, gny, gnx))
    o = orc.Oracle(orc.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else "") + "sbc = -1\n"), smoother=orc.GS_RB)
    o.set_const()
    ref = o.relax(0, da_g, res_g, nsweeps)
    got = np.zeros_like(ref)
    for r in range(world):
        sy, sx = tiling.tile_slice(r, px, py, tile, tile)
        got[:, sy, sx] = np.load(tmp_path / f"tile{r}.npy")
    assert np.array_equal(got, ref)


def test_tile_tables():
    assert tiling.tile_grid(8) == (2, 4) and tiling.tile_grid(1) == (1, 1)
    nb = tiling.neighbours(3, 2, 4)          # tile (1, 1)
    assert nb == dict(W=2, E=-1, S=1, N=5, SW=0, SE=-1, NW=4, NE=-1)
    assert tiling.walls(0, 2, 4) == dict(W=True, E=False, S=True, N=False)
    assert tiling.walls(7, 2, 4) == dict(W=False, E=True, S=False, N=True)
    assert tiling.neighbours(0, 2, 4, periodic=True) == dict(W=1, E=1, S=6, N=2, SW=7, SE=7, NW=3, NE=3)
    assert tiling.neighbours(0, 2, 1, periodic=True)["S"] == 0 and not any(tiling.walls(0, 2, 1, periodic=True).values())
    assert tiling.exchange_order(["W", "E", "S", "N"]) == (["W", "E", "S", "N"], ["E", "W", "N", "S"])
    with pytest.raises(ValueError):
        tiling.tile_grid(3)
    f = tiling.synthetic_tile(lambda l, y, x: np.outer(y, x) + l, 3, 2, 2, 2, 4, 4)
    assert f.shape == (2, 4, 4) and f[1, 0, 0] == pytest.approx((4.5 / 8) * (4.5 / 8) + 1)
