"""The sharded operator path of bench.py --gpus N as REAL processes (one per rank, all on the one GPU of the test box):
each rank owns its patches, applies the operator with its own device context and reduces the halo through HaloExchanger's
device pack/unpack kernels; the transport is gloo (staged through host memory) because RCCL refuses two ranks on one device.
The reduced result must equal the single-context global apply on every rank's slots."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mimsem_amd.device import DeviceMesh, Engine
        from mimsem_amd.geom import Geom
        from mimsem_amd.mesh import CubedSphere, sphere_coords
        from mimsem_amd.partition import CHalo, HaloExchanger, build_plans, patches_of_rank
        from mimsem_amd.topo import Topo
        from mimsem_amd.workloads import SCALE, z_levels
        pn, ne, npatch, nk = 3, 4, 24, 3
        cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)
        rng = np.random.default_rng(123)                                  # the same global field on every rank
        xg = rng.standard_normal((nk, cs.nDofs1G)); x0g = rng.standard_normal((nk, cs.nDofs0G))

        def build(pids):
            topos = [Topo(cs, p, nk) for p in pids]
            geoms = [Geom(t, cs, coords, nk) for t in topos]
            for g in geoms:
                g.set_levels(z_levels(nk, g.n0))
            dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
            return dm, Engine(dm)
        dm, eng = build(patches_of_rank(npatch, world, rank))
        plans = build_plans(cs, world, rank, dm.gid0, dm.gid1)
        ok = True
        for form, op, xglob, gid in ((1, "UMAT", xg, dm.gid1), (0, "PMAT", x0g, dm.gid0)):
            y = eng.apply(op, eng.tensor(xglob[:, gid]), lev0=0, scale=SCALE, flags=1 if form == 1 else 0)
            halo = HaloExchanger(plans[form], engine=eng)
            halo.reverse_add(y)                    # owners hold the sums
            halo.forward_insert(y)                 # ghosts too
            halo.reverse_add(y.clone())            # second use of the cached buffers
            # the same exchange through the C ABI (mimsem_halo_create/_begin/_end, host-callback transport): bit for bit
            y2 = eng.apply(op, eng.tensor(xglob[:, gid]), lev0=0, scale=SCALE, flags=1 if form == 1 else 0)
            ch = CHalo(plans[form], eng, max_nlev=nk, transport="dist")
            ch.reverse_add(y2); ch.forward_insert(y2)
            ok = ok and bool(torch.equal(y2, y))
            # ... and through the ONE-SIDED transport (round 6: receive buffers exported through hipIpc, the pack kernel writes straight into the
            # neighbour's buffer and publishes a sequence flag, the unpack waits for the flags: kernels only) -- bit for bit, over several
            # exchanges in a row (both parities of the double receive buffer), the symmetric "pair" exchange included, and RECORDED in a graph
            cp = CHalo(plans[form], eng, max_nlev=nk, transport="peer")
            y4 = eng.apply(op, eng.tensor(xglob[:, gid]), lev0=0, scale=SCALE, flags=1 if form == 1 else 0)
            cp.reverse_add(y4); cp.forward_insert(y4)
            same = bool(torch.equal(y4, y))
            for rep in range(3):
                a = eng.apply(op, eng.tensor(xglob[:, gid] * (rep + 2.0)), lev0=0, scale=SCALE, flags=1 if form == 1 else 0); b = a.clone()
                if form == 1:
                    ch.sum_all(a); cp.sum_all(b)
                else:
                    ch.reverse_add(a); ch.forward_insert(a); cp.reverse_add(b); cp.forward_insert(b)
                same = same and bool(torch.equal(a, b))
            if form == 1:
                src = eng.apply(op, eng.tensor(xglob[:, gid] * 7.0), lev0=0, scale=SCALE, flags=1); want_g = src.clone(); ch.sum_all(want_g)
                buf = torch.zeros_like(src)

                def recorded():
                    buf.copy_(src)
                    cp.sum_all(buf)
                graph, _ = eng.capture(recorded)
                for _ in range(3):                                     # every replay is a new exchange: sequence number and parity live on the device
                    graph.replay()
                    torch.cuda.synchronize()
                    same = same and bool(torch.equal(buf, want_g))
            same = same and cp.peer_timeouts() == {}
            if not same:
                print("rank", rank, "one-sided transport differs from the callback transport (form %d)" % form, cp.peer_timeouts(), flush=True)
            ok = ok and same
            cp.close()
            ch.close()
            if form == 1:
                # DistEngine with the interior / boundary split: boundary groups, exchange in flight, interior groups, unpack
                from mimsem_amd.distributed import DistEngine
                dm2, eng2 = build(patches_of_rank(npatch, world, rank))
                de = DistEngine(eng2, cs, world, rank, overlap=True)
                y3 = de.apply(op, eng2.tensor(xglob[:, gid]), lev0=0, scale=SCALE, flags=1)
                ok = ok and bool(torch.equal(y3, y))
                # ... and the 0-form completion of the same DistEngine: REVERSE/ADD then FORWARD/INSERT plans of the C ABI
                p3 = de.apply("PMAT", eng2.tensor(x0g[:, dm2.gid0]), lev0=0, scale=SCALE, flags=0)
                p0 = eng.apply("PMAT", eng.tensor(x0g[:, dm.gid0]), lev0=0, scale=SCALE, flags=0)
                h0 = HaloExchanger(plans[0], engine=eng); h0.reverse_add(p0); h0.forward_insert(p0)
                ok = ok and bool(torch.equal(p3, p0))
                de.close()
            if rank == 0:
                dm1, eng1 = build(list(range(npatch)))
                want = eng1.apply(op, eng1.tensor(xglob), lev0=0, scale=SCALE, flags=1 if form == 1 else 0).cpu().numpy()
                t = torch.as_tensor(want)
            else:
                t = torch.empty(nk, cs.nDofs1G if form == 1 else cs.nDofs0G, dtype=torch.float64)
            dist.broadcast(t, 0)
            want = t.numpy()
            err = np.linalg.norm(y.cpu().numpy() - want[:, gid]) / np.linalg.norm(want)
            ok = ok and bool(err < 1e-12)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_apply_and_halo_as_processes(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_bench_multi_rank_control_flow_rehearsal():
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one rank per process), except that both ranks
    sit on the one GPU of the test box and the transport is gloo (MIMSEM_BENCH_REHEARSAL=1): exercises sharding, halo plans,
    the barrier / max-over-ranks timing and the JSON line of the N > 1 path"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIMSEM_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, [len(l) for l in lines]        # ONE compact line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["patches_per_gpu"] == 12 and d["config"]["units_per_step"] == 103680
    assert d["config"]["halo_transport"] == "dist"                     # the C ABI's halo plans with the host-callback transport (rehearsal)
    assert "extras_with_errors" not in d, d
    full = json.load(open(os.path.join(root, d["extras_file"])))      # the full object sits beside it
    for key in ("weak_scaled", "column_sharded", "horiz_sharded"):     # the N > 1 extras with real work per rank ran too
        assert key in full and "error" not in full[key], full.get(key)
        assert key in d["summary"]
    assert full["weak_scaled"]["scaling"] == "weak" and full["weak_scaled"]["units_per_rank"] == 829440
    assert d["summary"]["weak_scaled"]["value"] == pytest.approx(full["weak_scaled"]["value"], rel=1e-3)


def test_bench_started_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a user, or a driver without torchrun, starts it): the parent
    starts the ranks as a CHILD torch.distributed.run before touching the GPU, relays rank 0's compact line and returns its status"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MIMSEM_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-column",
                          "--no-horiz-sharded"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["config"]["patches_per_gpu"] == 12
    # the mismatch the old code let through silently
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_extras_watchdog_keeps_the_headline_line():
    """N > 1: an extra that does not come back (here: a budget of 0 s) must not cost the headline -- rank 0 prints the
    line measured so far with `extras_watchdog` set (naming the extra that was in flight) -- and must not pass for a clean run
    either: every rank leaves with exit status 3"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIMSEM_BENCH_REHEARSAL="1", MIMSEM_BENCH_EXTRAS_BUDGET="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode != 0, "a run whose extras hung must not exit 0"
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "roofline" in d
    assert "in_flight" in d["extras_watchdog"] and "exit status 3" in d["extras_watchdog"]["note"]


def _sw_worker(rank, world, port, q, peer=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mimsem_amd.device import DeviceMesh, Engine
        from mimsem_amd.distributed import DistEngine
        from mimsem_amd.geom import Geom
        from mimsem_amd.mesh import CubedSphere, sphere_coords
        from mimsem_amd.partition import patches_of_rank
        from mimsem_amd.sweqn import SWEqn, williamson2
        from mimsem_amd.topo import Topo
        pn, ne, npatch = 3, 4, 6
        cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)

        def build(pids):
            topos = [Topo(cs, p, 1) for p in pids]
            geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
            for g in geoms:
                g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
            dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
            xq = np.zeros((int(max(g.loc0.max() for g in geoms)) + 1, 3))
            for g in geoms:
                xq[g.loc0] = coords[g.loc0]
            return dm, Engine(dm), xq[dm.gidq]
        dm, eng, xq = build(patches_of_rank(npatch, world, rank))
        # world 2: the C ABI's plans on the ONE-SIDED transport -- exchanges that are kernels only, so the whole Picard iteration is RECORDED as a
        # hipGraph on every rank (exchanges inside); world 3: the host-staged exchanger, the same launches eagerly
        peer = (world == 2) if peer is None else peer
        deng = DistEngine(eng, cs, world, rank, overlap=True, transport="peer") if peer else DistEngine(eng, cs, world, rank)
        S = SWEqn(deng, xq)
        uq, hq = williamson2(torch.as_tensor(xq, device=eng.device), alpha=0.0)
        lam = torch.atan2(torch.as_tensor(xq[:, 1]), torch.as_tensor(xq[:, 0])).to(eng.device)
        uq = uq + torch.stack([3.0 * torch.sin(2 * lam), 2.0 * torch.cos(lam)], dim=1)            # perturbed: every term active
        u0, h0 = S.init1(uq), S.init2(hq)
        import torch.distributed as tdist
        calls = {"n": 0}
        real_all_reduce = tdist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        tdist.all_reduce = counting_all_reduce
        u1, h1 = S.solve(u0, h0, 360.0, nits=2, q_exact=False)
        tdist.all_reduce = real_all_reduce
        # round 6: the FIXED-LENGTH mode runs over the halo -- both Picard iterations, no fallback, no re-estimate, and ONE all-reduce per
        # Picard iteration (the check norms) after the set-up (counted over a second step: the spectral estimates of the first need dots)
        fixed = S.fixed_iterations == 2 and S.adaptive_iterations == 0 and S.recalibrations == 0
        hist1 = list(S.history)
        calls["n"] = 0
        tdist.all_reduce = counting_all_reduce
        u2, h2 = S.solve(u1, h1, 360.0, nits=2, q_exact=False)
        tdist.all_reduce = real_all_reduce
        fixed = fixed and S.fixed_iterations == 4 and S.adaptive_iterations == 0 and calls["n"] == 2
        if peer:
            fixed = fixed and S._pg.record and all(g is not None for g, _, _ in S._pg.graphs.values()) and deng.chalo.peer_timeouts() == {}
        if not fixed:
            print("rank", rank, "fixed-length mode not engaged:", S.fixed_iterations, S.adaptive_iterations, S.recalibrations, calls, getattr(S, "last_miss", None), flush=True)
        ug = deng.gather_owned(1, u1, dm.gid1, cs.nDofs1G).cpu().numpy()
        hg = deng.gather_owned(2, h1, dm.gid2, cs.nDofs2G).cpu().numpy()
        ok = fixed
        if rank == 0:                                # the same step on ONE context holding the whole sphere: the graphed fixed-length path
            dm1, eng1, xq1 = build(list(range(npatch)))
            S1 = SWEqn(eng1, xq1)
            uq1, hq1 = williamson2(torch.as_tensor(xq1, device=eng1.device), alpha=0.0)
            lam1 = torch.atan2(torch.as_tensor(xq1[:, 1]), torch.as_tensor(xq1[:, 0])).to(eng1.device)
            uq1 = uq1 + torch.stack([3.0 * torch.sin(2 * lam1), 2.0 * torch.cos(lam1)], dim=1)
            a0, b0 = S1.init1(uq1), S1.init2(hq1)
            a1, b1 = S1.solve(a0, b0, 360.0, nits=2, q_exact=False)
            eu = np.linalg.norm(ug - a1.cpu().numpy()) / np.linalg.norm(a1.cpu().numpy())
            eh = np.linalg.norm(hg - b1.cpu().numpy()) / np.linalg.norm(b1.cpu().numpy())
            du = np.linalg.norm((a1 - a0).cpu().numpy())
            same_counts = S1.fixed_iterations == 2 and {k: S.its[k] for k in ("A", "F", "q")} == {k: S1.its[k] for k in ("A", "F", "q")}
            ok = bool(fixed and same_counts and eu < 1e-10 and eh < 1e-11 and du > 0 and np.allclose(hist1, S1.history, rtol=1e-6))
            # ... and the adaptive path (Krylov solves to the same tolerance) lands on the same state
            S2 = SWEqn(eng1, xq1, use_graphs=False)
            c1, d1 = S2.solve(a0, b0, 360.0, nits=2, q_exact=False)
            eu2 = np.linalg.norm(ug - c1.cpu().numpy()) / np.linalg.norm(c1.cpu().numpy())
            ok = ok and bool(eu2 < 1e-10)
            print("sharded fixed-length SW step, world %d: |u - u_1ctx| = %.2e  |h - h_1ctx| = %.2e  |u - u_adaptive| = %.2e  steps %s" % (world, eu, eh, eu2, dict(S.its)), flush=True)
            if not ok:
                print("sharded SW step mismatch", eu, eh, eu2, hist1, S1.history, S.its, S1.its, flush=True)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,peer", [(2, True), (3, False), (3, True)], ids=["2_one_sided_recorded", "3_host_staged_eager", "3_one_sided_recorded"])
def test_sharded_shallow_water_step_as_processes(world, peer):
    """N3 on several ranks: SWEqn over a DistEngine takes the same Picard step as the single-context run IN THE FIXED-LENGTH MODE (Chebyshev
    solves with the halo exchanges inside, no all-reduce in any solve, one all-reduce of the check norms per Picard iteration), with the same
    step counts; the adaptive path agrees too"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sw_worker, args=(r, world, port, q, peer)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def _hs_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mimsem_amd.device import DeviceMesh, Engine
        from mimsem_amd.distributed import DistEngine
        from mimsem_amd.geom import Geom
        from mimsem_amd.horizsolve import HorizSolve
        from mimsem_amd.mesh import CubedSphere, sphere_coords
        from mimsem_amd.partition import patches_of_rank
        from mimsem_amd.topo import Topo
        from tests.helpers import z_levels
        pn, ne, npatch, nk = 3, 2, 6, 3
        cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)

        def build(pids):
            topos = [Topo(cs, p, nk) for p in pids]
            geoms = [Geom(t, cs, coords, nk) for t in topos]
            for g in geoms:
                g.set_levels(z_levels(nk, g.n0))
            dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
            xq = np.zeros((int(max(g.loc0.max() for g in geoms)) + 1, 3))
            for g in geoms:
                xq[g.loc0] = coords[g.loc0]
            return dm, Engine(dm), xq[dm.gidq]
        r = np.random.default_rng(31)                          # identical global fields on every rank
        N1, N2 = cs.nDofs1G, cs.nDofs2G
        dm, eng, xq = build(patches_of_rank(npatch, world, rank))
        area = float(dm.det.mean()) * 4.0 / (pn * pn); dz = float(dm.thick.mean()); ln = area ** 0.5
        G = dict(u1=r.standard_normal((nk, N1)) * 20.0 * ln * dz, h1=r.uniform(0.8, 1.2, (nk, N2)) * area * dz,
                 th=r.uniform(290, 310, (nk, N2)) * area * dz, Pi=r.uniform(900, 1000, (nk, N2)) * area * dz,
                 vz=r.standard_normal((nk - 1, N2)) * area, dudz=r.standard_normal((nk - 1, N1)) * 1e-3 * ln)
        G["u2"] = G["u1"] * 1.03; G["h2"] = G["h1"] * 1.01

        def run(e, d, xq_):
            hs = HorizSolve(e, quad_coords=xq_)
            t = lambda key, gid: e.tensor(G[key][:, gid])
            u1, u2, dud = t("u1", d.gid1), t("u2", d.gid1), t("dudz", d.gid1)
            h1, h2, th, Pi, vz = (t(k, d.gid2) for k in ("h1", "h2", "th", "Pi", "vz"))
            dF, dG, Fk, Gk = hs.advection_rhs_ec(u1, u2, h1, h2, th)
            fu = hs.momentum_rhs_ec(th, dud, dud, vz, vz, Pi, u1, u2, h1, h2, Fx=Fk, Fk=Fk)
            return fu, dG, hs.k2i, hs
        import torch.distributed as tdist
        calls = {"n": 0}
        real_all_reduce = tdist.all_reduce

        def counting_all_reduce(*a, **k):
            calls["n"] += 1
            return real_all_reduce(*a, **k)
        deng = DistEngine(eng, cs, world, rank)
        fu, dG, k2i, hs_d = run(deng, dm, xq)
        hs_d.verify()
        # round 6: the 1-form mass solves of the sharded evaluation are FIXED-LENGTH Chebyshev iterations with the exchanges inside -- a second
        # evaluation (the spectral bounds are known) holds exactly TWO all-reduces: the kinetic-to-internal exchange sum and the log of the
        # seven solves' check norms, none inside a solve
        tdist.all_reduce = counting_all_reduce
        t = lambda key, gid: eng.tensor(G[key][:, gid])
        dF2, dG2, Fk2, Gk2 = hs_d.advection_rhs_ec(t("u1", dm.gid1), t("u2", dm.gid1), t("h1", dm.gid2), t("h2", dm.gid2), t("th", dm.gid2))
        in_solves = calls["n"]
        checks = hs_d.verify()
        tdist.all_reduce = real_all_reduce
        cheb = bool(hs_d.m1.chebyshev and in_solves == 0 and calls["n"] == 1 and checks and hs_d.m1.solves_checked >= 10 and hs_d.m1.solves_missed == 0)
        if not cheb:
            print("rank", rank, "sharded mass solves: chebyshev", hs_d.m1.chebyshev, "all-reduces inside", in_solves, "total", calls["n"], "checks", checks,
                  hs_d.m1.solves_checked, hs_d.m1.solves_missed, hs_d.m1.worst_check, flush=True)
        fug = deng.gather_owned(1, fu, dm.gid1, N1).cpu().numpy()
        dGg = deng.gather_owned(2, dG, dm.gid2, N2).cpu().numpy()
        ok = cheb
        if rank == 0:
            dm1, eng1, xq1 = build(list(range(npatch)))
            f1, g1, k1, _ = run(eng1, dm1, xq1)
            e1 = np.linalg.norm(fug - f1.cpu().numpy()) / np.linalg.norm(f1.cpu().numpy())
            e2 = np.linalg.norm(dGg - g1.cpu().numpy()) / np.linalg.norm(g1.cpu().numpy())
            ok = bool(cheb and e1 < 1e-9 and e2 < 1e-9 and abs(k2i - k1) < 1e-9 * abs(k1))
            if not ok:
                print("sharded HorizSolve mismatch", e1, e2, k2i, k1, flush=True)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_horizsolve_rhs_as_processes():
    """N2 on two ranks: HorizSolve (advection_rhs_ec + momentum_rhs_ec with viscosity) over a DistEngine equals the one-context result"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
