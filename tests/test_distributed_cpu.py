"""world_size-2 test of the sharding / exchange logic (pathtracer_amd.distributed)
on the CPU with the gloo backend.  The phases are computed by the CPU oracle
plugged in as the backend, so only the host-side distribution logic is under
test here: photon quotas and Halton striding, the all-gather of both maps, the
shoot-counter reduction, pixel-row ownership and the final plane merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_py as O
from pathtracer_amd import distributed as D
from pathtracer_amd.scene import Scene, asset_scene

W = H = 40
GS, CS, ITERS = 3000, 400, 2


class OracleBackend:
    """The backend interface of pathtracer_amd.distributed on top of the CPU oracle (tests only)."""

    def __init__(self, scene, param, rank, world):
        self.osc = O.OracleScene(scene, W, H)
        self.r = O.OracleSppm(self.osc, param, scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED,
                              rank=rank, world=world, threads=2)
        self.param, self.rank, self.world = param, rank, world
        self.cap = [D.quota(param.globalSize, 0, world), D.quota(param.causticSize, 0, world)]

    def photon_pass(self, it, stats=None):
        self.r.photon_pass(it)

    def export_local(self, which):
        cap = max(self.cap[which], 1)
        pos, power, th, ph, shot = self.r.photons(which, cap)
        n = len(pos)
        P = torch.zeros((cap, 3)); Wt = torch.zeros((cap, 3)); Dr = torch.zeros((cap,), dtype=torch.int16)
        P[:n] = torch.from_numpy(pos.copy()); Wt[:n] = torch.from_numpy(power.copy())
        Dr[:n] = torch.from_numpy((th.astype(np.uint16) | (ph.astype(np.uint16) << 8)).view(np.int16).copy())
        return P, Wt, Dr, n, shot

    def set_map(self, which, pos, power, direction, n, shot_total):
        d = direction[:n].numpy().view(np.uint16)
        self.r.set_map_photons(which, pos[:n].numpy(), power[:n].numpy(), (d & 255).astype(np.uint8),
                               (d >> 8).astype(np.uint8), shot_total)

    def build_maps(self, stats=None):
        self.r.build_maps()

    def camera_pass(self, it, stats=None):
        self.r.camera_pass(it)

    def export_planes(self):
        return torch.from_numpy(self.r.pixels().T.copy().reshape(13, H, W))


def _scene_and_param():
    sc = Scene.load(asset_scene("cornell_sphere.scene"))
    return sc, sc.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, prm = _scene_and_param()
    be = OracleBackend(sc, prm, rank, world)
    totals = []
    for it in range(ITERS):
        totals.append(D.iteration(be, it, world))
    merged = D.gather_planes(be, rank, world, H)
    sg, scc, _ = be.r.counters()
    np.save(os.path.join(out_dir, f"counters_{rank}.npy"), np.array([sg, scc] + [v for t in totals for pair in t if pair for v in pair]))
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_quota_and_row_ownership():
    assert [D.quota(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert sum(D.quota(200000, r, 8) for r in range(8)) == 200000
    owners = [D.row_owner(y, 3) for y in range(48)]
    assert owners[:8] == [0] * 8 and owners[8:16] == [1] * 8 and owners[16:24] == [2] * 8 and owners[24] == 0
    assert D.row_owner(5, 1) == 0


def test_two_ranks_over_gloo_match_a_sequential_emulation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    merged = np.load(tmp_path / "merged.npy")
    c0 = np.load(tmp_path / "counters_0.npy")
    c1 = np.load(tmp_path / "counters_1.npy")
    assert (c0 == c1).all()                                  # every rank holds the same totals
    # the same two ranks emulated in one process, exchange done by hand
    sc, prm = _scene_and_param()
    bes = [OracleBackend(sc, prm, r, world) for r in range(world)]
    for it in range(ITERS):
        for be in bes:
            be.photon_pass(it)
        for which in (0, 1):
            parts = [be.export_local(which) for be in bes]
            n_tot = sum(p[3] for p in parts)
            shot = sum(p[4] for p in parts)
            cat = [torch.cat([p[i][:p[3]] for p in parts]) for i in range(3)]
            for be in bes:
                be.set_map(which, cat[0], cat[1], cat[2], n_tot, shot)
            if which == 0:
                assert n_tot == GS
            else:
                assert n_tot == CS
        for be in bes:
            be.build_maps()
            be.camera_pass(it)
    planes = [be.export_planes().numpy() for be in bes]
    expect = planes[0].copy()
    for y in range(H):
        expect[:, y, :] = planes[D.row_owner(y, world)][:, y, :]
    assert np.array_equal(merged, expect)
    sg, scc, _ = bes[0].r.counters()
    assert c0[0] == sg and c0[1] == scc
    # rows a rank does not own keep their initial state
    for r in range(world):
        other = [y for y in range(H) if D.row_owner(y, world) != r]
        assert (planes[r][:11, other, :] == 0).all() and (planes[r][11:, other, :] == np.float32(0.08)).all()
    # the two ranks traced disjoint Halton subsequences: their photon sets differ
    a, b = bes[0].export_local(0), bes[1].export_local(0)
    assert a[3] == D.quota(GS, 0, world) and b[3] == D.quota(GS, 1, world)
    assert not torch.equal(a[0][:100], b[0][:100])


def test_sharded_render_agrees_statistically_with_single_rank():
    """Sharding changes which Halton indices fill the maps, not the estimate: mean image levels agree."""
    sc, prm = _scene_and_param()
    single = OracleBackend(sc, prm, 0, 1)
    for it in range(ITERS):
        D.iteration(single, it, 1)
    img1 = single.r.resolve().astype(np.float64)
    bes = [OracleBackend(sc, prm, r, 2) for r in range(2)]
    for it in range(ITERS):
        for be in bes:
            be.photon_pass(it)
        for which in (0, 1):
            parts = [be.export_local(which) for be in bes]
            cat = [torch.cat([p[i][:p[3]] for p in parts]) for i in range(3)]
            for be in bes:
                be.set_map(which, cat[0], cat[1], cat[2], sum(p[3] for p in parts), sum(p[4] for p in parts))
        for be in bes:
            be.build_maps()
            be.camera_pass(it)
    planes = [be.export_planes().numpy() for be in bes]
    merged = planes[0].copy()
    for y in range(H):
        merged[:, y, :] = planes[D.row_owner(y, 2)][:, y, :]
    bes[0].r.set_pixels(merged.reshape(13, -1).T.copy())
    img2 = bes[0].r.resolve().astype(np.float64)
    assert abs(img1.mean() - img2.mean()) / img1.mean() < 0.05
