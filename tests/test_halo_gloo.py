"""world_size-2 (and 3) gloo tests of the patch sharding + halo exchange plans on CPU: the sharded
element-local partial sums, reduced over the halo, must equal the single-rank global result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, form, q, pn=2, ne=2, npatch=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mimsem_amd.mesh import CubedSphere
        from mimsem_amd.partition import HaloExchanger, build_plans, patches_of_rank
        from mimsem_amd.topo import Topo
        cs = CubedSphere(pn, ne, npatch)
        nG = cs.nDofs0G if form == 0 else cs.nDofs1G
        pids = patches_of_rank(npatch, world, rank)
        topos = [Topo(cs, p) for p in pids]
        if form == 0:
            touched = np.unique(np.concatenate([t.loc0 for t in topos]))
        else:
            touched = np.unique(np.concatenate([t.loc1 for t in topos]))
        plan = build_plans(cs, world, rank, touched if form == 0 else np.zeros(0, np.int32), touched if form == 1 else np.zeros(0, np.int32))[form]
        ex = HaloExchanger(plan)
        # element-local "partial sums": each patch adds (pid+1)*(gid+1) at every slot it touches, 2 levels
        v = torch.zeros(2, touched.size, dtype=torch.float64)
        for t in topos:
            g = t.loc0 if form == 0 else t.loc1
            s = torch.as_tensor(np.searchsorted(touched, g)).long()
            v[0].index_add_(0, s, torch.as_tensor((t.pi + 1.0) * (g + 1.0)))
            v[1].index_add_(0, s, torch.as_tensor((t.pi + 2.0) * (g + 1.0)))
        v2 = v.clone()
        ex.reverse_add(v)
        ex.forward_insert(v)
        ex.sum_all(v2)                                  # the one-exchange form (edges) / its two-step fallback (nodes)
        assert torch.equal(v, v2) and ex.pairwise == (form == 1)
        # expected: the all-patch sum at every slot I hold
        want = np.zeros((2, nG))
        for p in range(npatch):
            t = Topo(cs, p); g = t.loc0 if form == 0 else t.loc1
            np.add.at(want[0], g, (p + 1.0) * (g + 1.0)); np.add.at(want[1], g, (p + 2.0) * (g + 1.0))
        ok = np.array_equal(v.numpy(), want[:, touched])
        q.put((rank, bool(ok), int((~plan.owned).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,form", [(2, 0), (2, 1), (3, 1), (3, 0)])
def test_sharded_halo_matches_global(world, form):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, form, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sum(ng for _, _, ng in res) > 0          # the test really exchanged ghosts


@pytest.mark.parametrize("form", [0, 1])
def test_config1_six_ranks_halo(form):
    """BASELINE config 1: p=3, 8x8 elements per face, 6 MPI ranks = one patch per rank -- the reference's own decomposition
    (scr/Setup.py pn=3 ne=8 6 procs); the 6-rank halo reduce / ghost fill equals the global sums on every rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 6, port, form, q, 3, 8, 6)) for r in range(6)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(ng > 0 for _, _, ng in res)          # every rank holds ghosts on this decomposition


def test_ownership_tables_partition_all_ids():
    from mimsem_amd.mesh import CubedSphere
    from mimsem_amd.partition import owner_tables
    cs = CubedSphere(3, 4, 24)
    own0, own1 = owner_tables(cs)
    # owned counts per patch equal the reference's n0l / n1l (scr/Proc2.py:54-70)
    for p in cs.patches:
        assert (own0 == p.pid).sum() == p.n0l
        assert (own1 == p.pid).sum() == p.n1xl + p.n1yl
