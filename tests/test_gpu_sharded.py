"""Sharded (multi-GPU style) operator application on ONE GPU: the sphere's patches are split over R virtual
ranks, each with its own device context and compacted numbering; partial sums are exchanged with the very
HaloPlan slot lists and device pack/unpack kernels the RCCL path uses (transport emulated by handing the packed
buffers over in-process).  Result must equal the single-context global apply."""
import numpy as np
import pytest
import torch

from mimsem_amd.workloads import SCALE, z_levels

gpu = pytest.mark.gpu


def _build(world, nk=3, pn=3, ne=4, npatch=24):
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.partition import build_plans, patches_of_rank
    from mimsem_amd.topo import Topo
    cs = CubedSphere(pn, ne, npatch)
    coords = sphere_coords(pn, ne)
    ranks = []
    for r in range(world):
        pids = patches_of_rank(npatch, world, r)
        topos = [Topo(cs, p, nk) for p in pids]
        geoms = [Geom(t, cs, coords, nk) for t in topos]
        for g in geoms:
            g.set_levels(z_levels(nk, g.n0))
        dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
        ranks.append((dm, Engine(dm), build_plans(cs, world, r, dm.gid0, dm.gid1)))
    return cs, ranks


@gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_apply_matches_global(world):
    nk = 3
    cs, single = _build(1, nk)
    dm1, eng1, _ = single[0]
    rng = np.random.default_rng(5)
    xg = rng.standard_normal((nk, cs.nDofs1G))           # global 1-form field, indexed by global edge id
    hg = rng.uniform(1, 2, (nk, cs.nDofs2G)) * 1e6
    assert np.array_equal(dm1.gid1, np.arange(cs.nDofs1G))
    want = eng1.apply("UHMAT", eng1.tensor(xg), f=eng1.tensor(hg), lev0=0, scale=SCALE, flags=1).cpu().numpy()

    cs, ranks = _build(world, nk)
    ys, dev = [], []
    for dm, eng, plans in ranks:
        x = eng.tensor(xg[:, dm.gid1]); h = eng.tensor(hg[:, dm.gid2])
        ys.append(eng.apply("UHMAT", x, f=h, lev0=0, scale=SCALE, flags=1))
    # REVERSE/ADD: every rank packs its ghost partial sums per owner; owners unpack-add in rank order
    packed = {}
    for r, (dm, eng, plans) in enumerate(ranks):
        p1 = plans[1]
        for owner, slots in p1.ghost_slots.items():
            idx = torch.as_tensor(slots, dtype=torch.int32, device=eng.device)
            packed[(r, owner)] = eng.halo_pack(idx, ys[r])
    for o, (dm, eng, plans) in enumerate(ranks):
        p1 = plans[1]
        for src in sorted(p1.mirror_slots):
            idx = torch.as_tensor(p1.mirror_slots[src], dtype=torch.int32, device=eng.device)
            buf = packed[(src, o)]
            assert buf.shape[1] == idx.numel()                 # both sides agree on the message length
            eng.halo_unpack(idx, buf, ys[o], add=True)
    # owners now hold the full sums: compare owned slots with the global result (1e-10: sums associate differently)
    seen = np.zeros(cs.nDofs1G, dtype=bool)
    for r, (dm, eng, plans) in enumerate(ranks):
        own = plans[1].owned
        got = ys[r].cpu().numpy()[:, own]
        ref = want[:, dm.gid1[own]]
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-12
        seen[dm.gid1[own]] = True
    assert seen.all()                                          # every global edge has exactly one owner rank
    # FORWARD/INSERT: ghosts receive the owners' totals -> every rank's full local vector matches
    packed = {}
    for o, (dm, eng, plans) in enumerate(ranks):
        for dst, slots in plans[1].mirror_slots.items():
            idx = torch.as_tensor(slots, dtype=torch.int32, device=eng.device)
            packed[(o, dst)] = eng.halo_pack(idx, ys[o])
    for r, (dm, eng, plans) in enumerate(ranks):
        for owner, slots in plans[1].ghost_slots.items():
            idx = torch.as_tensor(slots, dtype=torch.int32, device=eng.device)
            eng.halo_unpack(idx, packed[(owner, r)], ys[r], add=False)
        got = ys[r].cpu().numpy()
        assert np.linalg.norm(got - want[:, dm.gid1]) / np.linalg.norm(want) < 1e-12


def test_plan_consistency_for_benchmark_grid():
    """pure host check at the bench.py configuration (p=3, 24x24x6, 24 patches) for 2/4/8 ranks"""
    from mimsem_amd.mesh import CubedSphere
    from mimsem_amd.partition import build_plans, patches_of_rank
    cs = CubedSphere(3, 24, 24)
    for world in (2, 4, 8):
        plans = []
        for r in range(world):
            pids = patches_of_rank(24, world, r)
            g1 = np.unique(np.concatenate([cs.patches[p].loc1 for p in pids]))
            g0 = np.unique(np.concatenate([cs.patches[p].loc0 for p in pids]))
            plans.append((g0, g1, build_plans(cs, world, r, g0, g1)))
        for a in range(world):
            for b in range(world):
                if a == b:
                    continue
                for form in (0, 1):
                    ga, gb = plans[a][form], plans[b][form]
                    send = plans[a][2][form].ghost_slots.get(b)
                    recv = plans[b][2][form].mirror_slots.get(a)
                    assert (send is None) == (recv is None)
                    if send is not None:
                        assert np.array_equal(ga[send], gb[recv])      # same global ids, same order


def _emulated_exchange(exs, vs, send_key, recv_key, add):
    """what HaloExchanger._exchange does, with the all_to_all_single replaced by in-process copies between the virtual ranks"""
    nlev = vs[0].shape[0]
    sb = [ex._buffers(send_key, nlev, torch.float64) for ex in exs]
    rb = [ex._buffers(recv_key, nlev, torch.float64) for ex in exs]
    for ex, v, b in zip(exs, vs, sb):
        ex._move(ex.sides[send_key], 0, b, v)
    for src, ex in enumerate(exs):
        snd = ex.sides[send_key]
        for i, dst in enumerate(snd["ranks"]):
            rcv = exs[dst].sides[recv_key]
            j = rcv["ranks"].index(src)
            a, b = int(snd["off"][i]) * nlev, int(snd["off"][i + 1]) * nlev
            c, d = int(rcv["off"][j]) * nlev, int(rcv["off"][j + 1]) * nlev
            assert b - a == d - c
            rb[dst][c:d].copy_(sb[src][a:b])
    for ex, v, b in zip(exs, vs, rb):
        rcv = ex.sides[recv_key]
        if add:
            for (a, c) in rcv["ranges"]:
                ex._move(rcv, 2, b, v, a, c)
        else:
            ex._move(rcv, 1, b, v)


@gpu
@pytest.mark.parametrize("world,form", [(2, 1), (4, 0), (8, 1), (8, 0)])
def test_fused_halo_segments_match_global(world, form):
    """the one-launch segment pack/unpack (mimsem_halo_segments) + the alltoallv buffer layout of HaloExchanger, 0- and 1-forms
    (cube-corner nodes are ghosts of several ranks: the ADD is split into rank-ordered ranges)"""
    from mimsem_amd.partition import HaloExchanger
    nk = 3
    cs, single = _build(1, nk)
    dm1, eng1, _ = single[0]
    rng = np.random.default_rng(11)
    if form == 1:
        xg = rng.standard_normal((nk, cs.nDofs1G))
        want = eng1.apply("UMAT", eng1.tensor(xg), lev0=0, scale=SCALE, flags=1).cpu().numpy()
    else:
        xg = rng.standard_normal((nk, cs.nDofs0G))
        want = eng1.apply("PMAT", eng1.tensor(xg), lev0=0, scale=SCALE).cpu().numpy()
    cs, ranks = _build(world, nk)
    exs, ys, gids = [], [], []
    for dm, eng, plans in ranks:
        gid = dm.gid1 if form == 1 else dm.gid0
        x = eng.tensor(xg[:, gid])
        ys.append(eng.apply("UMAT" if form == 1 else "PMAT", x, lev0=0, scale=SCALE, flags=1 if form == 1 else 0))
        exs.append(HaloExchanger(plans[form], engine=eng)); gids.append(gid)
    assert form == 1 or any(len(ex.sides["mirror"]["ranges"]) > 1 for ex in exs) or world < 8
    _emulated_exchange(exs, ys, "ghost", "mirror", add=True)
    _emulated_exchange(exs, ys, "mirror", "ghost", add=False)
    for y, gid in zip(ys, gids):
        assert np.linalg.norm(y.cpu().numpy() - want[:, gid]) / np.linalg.norm(want) < 1e-12
