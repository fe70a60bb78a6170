"""The C++ host shim (mimsem_amd/host/mimsem_shim.hpp) compiled with g++ against the C ABI and run as a
reference-style call site; parity against the oracle inside the executable."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name="test_shim", extra=()):
    exe = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-O2", "-std=c++17", *extra, os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "mimsem_amd"), "-lmimsem_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "mimsem_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_links(tmp_path, oracle):
    """CPU: the header-only shim compiles with plain g++ (no HIP headers) and links against the C ABI"""
    from mimsem_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    assert os.path.exists(_build(str(tmp_path)))
    assert os.path.exists(_build(str(tmp_path), "test_ksp"))
    assert os.path.exists(_build(str(tmp_path), "test_sw"))
    assert os.path.exists(_build(str(tmp_path), "test_horiz"))
    assert os.path.exists(_build(str(tmp_path), "test_vert"))
    assert os.path.exists(_build(str(tmp_path), "test_sw_sharded", ["-pthread"]))
    assert os.path.exists(_build(str(tmp_path), "test_horiz_sharded", ["-pthread"]))


@pytest.mark.gpu
def test_shim_call_sites_match_oracle(tmp_path, oracle):
    out = subprocess.run([_build(str(tmp_path))], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "OK" in out.stdout


@pytest.mark.gpu
def test_ksp_call_sites_match_dense_solves(tmp_path, oracle):
    """HorizSolve::grad, HorizSolve::diagnose_fluxes and one kspA solve of SWEqn::solve written in C++ over the shim's KSP class
    (mimsem_ksp_*: the solve loops inside the library) against dense solves of the oracle's matrices"""
    out = subprocess.run([_build(str(tmp_path), "test_ksp")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("q_exact,nits,dt,topo,pn", [(False, 2, 360.0, False, 3), (True, 4, 600.0, False, 3), (False, 3, 300.0, True, 3),
                                                       (False, 2, 360.0, False, 2), (False, 2, 360.0, False, 4), (False, 2, 360.0, False, 5)],
                         ids=["galewsky_style", "williamson2_style", "with_topography", "order_2", "order_4", "order_5"])
def test_sw_step_driven_from_cpp(tmp_path, oracle, q_exact, nits, dt, topo, pn):
    """SWEqn::solve (src/SWEqn_Picard.cpp:727-791) orchestrated in C++ (mimsem_amd/host/mimsem_sweqn.hpp: KSP objects, fixed-length
    Chebyshev solves, the same as one hipGraph per Picard iteration) on the cubed sphere of tests/test_gpu_sweqn.py, against the numpy
    oracle's step (oracle/sw_oracle.py: dense matrices, LU for every KSPSolve).  Tolerance 1e-9 as for the Python host."""
    import numpy as np
    from mimsem_amd.device import DeviceMesh
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import write_sw_case
    from oracle import sw_oracle
    from tests.helpers import rel_l2
    ne, nsteps = 2, 2
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, 1) for p in range(6)]
    geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
    for g in geoms:
        g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
    dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
    O = sw_oracle.SWOracle(cs, topos, geoms, coords)
    th = np.arcsin(O.xq[:, 2] / 6371220.0); lam = np.arctan2(O.xq[:, 1], O.xq[:, 0])
    U0, H0 = 38.61068276698372, 2998.1154702758267
    uq = np.stack([U0 * np.cos(th) + 3.0 * np.sin(2 * lam) * np.cos(th), 2.0 * np.cos(lam) * np.cos(th) ** 2], axis=1)
    hq = H0 - (6371220.0 * 7.292e-5 * U0 + 0.5 * U0 * U0) * np.sin(th) ** 2 / 9.80616 + 40.0 * np.cos(th) * np.sin(lam)
    u0, h0 = O.init1(uq), O.init2(hq)
    # an isolated mountain (2-form dofs), the `bot` argument of SWEqn::solve (src/SWEqn_Picard.cpp:727; Phi += g M2 bot, :301-303)
    bot = O.init2(300.0 * np.exp(-((lam - 0.5) ** 2 + (th - 0.4) ** 2) / 0.1)) if topo else None
    fin, fout = str(tmp_path / "sw_in.bin"), str(tmp_path / "sw_out.bin")
    write_sw_case(fin, dm, O.fg, u0, h0, dt, nsteps, nits, q_exact, bot=bot)
    out = subprocess.run([_build(str(tmp_path), "test_sw"), fin, fout], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "ALL OK" in out.stdout
    ur, hr = u0, h0
    for _ in range(nsteps):
        ur, hr = O.solve(ur, hr, dt, nits=nits, q_exact=q_exact, bot=bot)
    ur, hr = O.solve(ur, hr, 0.5 * dt, nits=nits, q_exact=q_exact, bot=bot)       # (the C++ test ends with a step at half the time step)
    res = np.fromfile(fout, dtype=np.float64).reshape(4, dm.n1 + dm.n2)
    for mode in range(4):
        assert rel_l2(res[mode, :dm.n1], ur) < 1e-9 and rel_l2(res[mode, dm.n1:], hr) < 1e-9, mode


@pytest.mark.gpu
@pytest.mark.parametrize("m1,pn", [("chebyshev", 3), ("ksp", 3), ("chebyshev", 2), ("chebyshev", 4), ("any", 5)],
                         ids=["chebyshev", "ksp", "order_2", "order_4", "order_5"])
def test_horizsolve_driven_from_cpp(tmp_path, oracle, m1, pn):
    """N2 from C++: HorizSolve::advection_rhs_ec / diagnose_Phi / diagnose_q / momentum_rhs_ec (eul/HorizSolve.cpp:380-786) written over the C
    ABI (mimsem_amd/host/mimsem_horizsolve.hpp), all levels per call, against the dense restatement oracle/horiz_oracle.py -- the fields,
    mesh and tolerances of tests/test_gpu_next_rows.py::test_horizsolve_right_hand_sides"""
    import numpy as np
    from mimsem_amd.device import DeviceMesh
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import mesh_arrays, write_arrays, z_levels
    from oracle import horiz_oracle as ho
    ne, nk = 2, 3
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, nk) for p in range(6)]
    geoms = [Geom(t, cs, coords, nk) for t in topos]
    levs = z_levels(nk, geoms[0].n0)
    for g in geoms:
        g.set_levels(levs)
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    gd = ho.GlobalDense(cs, topos, geoms, coords, levs)
    H = ho.HorizOracle(gd)
    r = np.random.default_rng(31)
    area = np.mean([P.det.mean() for P in gd.P]) * 4.0 / (pn * pn); dz = np.mean([P.thick.mean() for P in gd.P]); ln = np.sqrt(area)
    N0, N1, N2 = dm.n0, gd.N1, gd.N2
    u1 = r.standard_normal((nk, N1)) * 20.0 * ln * dz; u2 = u1 * (1 + 0.05 * r.standard_normal((nk, N1)))
    h1 = r.uniform(0.8, 1.2, (nk, N2)) * area * dz; h2 = h1 * (1 + 0.01 * r.standard_normal((nk, N2)))
    th = r.uniform(290, 310, (nk, N2)) * area * dz; Pi = r.uniform(900, 1000, (nk, N2)) * area * dz
    velz = r.standard_normal((nk - 1, N2)) * area; velz2 = velz * (1 + 0.05 * r.standard_normal(velz.shape))
    dudz = r.standard_normal((nk - 1, N1)) * 1e-3 * ln; dudz2 = dudz * 1.1
    Fz = velz * 0.7
    dwdx = r.standard_normal((nk - 1, N1)) * 2e-4 * ln; dwdx2 = dwdx * 0.9
    fg = np.broadcast_to(H.fg, (nk, N0)) if np.ndim(H.fg) == 1 else H.fg
    arrays = mesh_arrays(dm)
    arrays.update(fg=fg, u1=u1, u2=u2, h1=h1, h2=h2, theta=th, Pi=Pi, velz1=velz, velz2=velz2, dudz1=dudz, dudz2=dudz2, Fz=Fz, dwdx1=dwdx, dwdx2=dwdx2)
    fin, fout = str(tmp_path / "horiz_in.arr"), str(tmp_path / "horiz_out.bin")
    write_arrays(fin, arrays)
    out = subprocess.run([_build(str(tmp_path), "test_horiz"), fin, fout] + (["ksp"] if m1 == "ksp" else []), capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "DONE" in out.stdout
    if m1 != "any":                                  # (order 5: whichever mass solve the library supports there; the results must agree either way)
        assert ("fixed-length Chebyshev" in out.stdout) == (m1 == "chebyshev")
    res = np.fromfile(fout, dtype=np.float64)
    pos = [0]

    def take(rows, n):
        a = res[pos[0]:pos[0] + rows * n].reshape(rows, n); pos[0] += rows * n
        return a
    gF, gG, gFk, gGk, gPhi, gq, fuA, fuB, fuC = (take(nk, N2), take(nk, N2), take(nk, N1), take(nk, N1), take(nk, N2), take(nk, N0), take(nk, N1),
                                                  take(nk, N1), take(nk, N1))
    k2iA, k2iB, del2 = res[pos[0]:pos[0] + 3]
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert abs(del2 - H.del2) < 1e-6 * abs(H.del2)
    dF, dG, Fk, Gk = H.advection_rhs_ec(u1, u2, h1, h2, th)
    assert rel(gFk, Fk) < 1e-10 and rel(gGk, Gk) < 1e-10
    assert rel(gF, dF) < 1e-9 and rel(gG, dG) < 1e-9
    for lev in range(nk):
        assert rel(gPhi[lev], H.diagnose_Phi(lev, u1[lev], u2[lev], velz, velz2)) < 1e-10
        assert rel(gq[lev], H.diagnose_q(lev, h1[lev], u1[lev])) < 1e-10
    for got, k2i_got, use_F, use_w in ((fuA, k2iA, False, False), (fuB, k2iB, True, False), (fuC, None, True, True)):
        k2i = 0.0
        for lev in range(nk):
            want, k = H.momentum_rhs_ec(lev, th[lev], dudz, dudz2, velz, velz2, Pi[lev], u1[lev], u2[lev], h1[lev], h2[lev],
                                        Fx=Fk[lev] if use_F else None, Fz=Fz if use_F else None, Fk=Fk[lev],
                                        dwdx1=dwdx if use_w else None, dwdx2=dwdx2 if use_w else None)
            k2i += k
            assert rel(got[lev], want) < 1e-8, (lev, use_F, use_w)
        if k2i_got is not None:
            assert abs(k2i_got - k2i) < 1e-8 * abs(k2i)
    assert rel(fuC, fuB) > 1e-6                      # the dwdx term is not lost in the noise of the comparison


@pytest.mark.gpu
@pytest.mark.parametrize("patch", [(3, 2, 6, 1, 6), (4, 1, 6, 0, 5), (2, 2, 6, 3, 4)], ids=lambda p: "p%d_ne%d_np%d_pi%d_nk%d" % p)
def test_vertical_newton_loop_driven_from_cpp(tmp_path, oracle, patch):
    """The caller of the column path from C++: VertSolve::solve_schur_eta (eul/VertSolve.cpp:1721-1973) as mimsem_host::VertSolveEta
    (mimsem_amd/host/mimsem_vertsolve.hpp) on the library's fused entry points, against the column-by-column numpy restatement
    oracle/vert_oracle.py -- the patch, state, forcing and tolerances of tests/test_gpu_column.py::test_vertical_newton_loop_matches_oracle"""
    import numpy as np
    from mimsem_amd.device import DeviceMesh
    from mimsem_amd.workloads import mesh_arrays, write_arrays
    from oracle import vert_oracle
    from tests.helpers import make_patch, rel_l2
    VSCALE = 1.0e8
    pn, ne, nprocs, pi, nk = patch
    cs, topo, geom, P, rng = make_patch(oracle, pn, ne, nprocs, pi, nk=nk, seed=7 * pn + nk)
    dm = DeviceMesh([topo], [geom], nk=nk, numbering="local")
    r = np.random.default_rng(29)
    nEl, n2 = P.nEl, P.n2e
    dt = 0.5
    levs = geom.levs
    W, Q = P.arr("W", (P.mp12, n2)), P.arr("Q", (P.mp12,))
    inds0 = geom.all_inds0_l()
    zv = np.zeros((nEl, nk * n2))
    for e in range(nEl):
        for k in range(nk):
            gz = 9.80616 * (levs[k, inds0[e]] + levs[k + 1, inds0[e]])
            zv[e, k * n2:(k + 1) * n2] = W.T @ (VSCALE * 0.5 * Q * gz)
    wd = np.diff(P.arr("qx", (P.mp1,)))
    wj = np.outer(wd, wd).ravel()
    detm = P.det.mean(axis=1)
    thm = np.stack([[P.thick[k, inds0[e]].mean() for k in range(nk)] for e in range(nEl)])
    rho_v, th_v = np.linspace(1.2, 0.5, nk), np.linspace(290.0, 330.0, nk)
    pi_v = 1004.5 * (287.0 * rho_v * th_v / 1.0e5) ** (287.0 / 717.5)
    col = lambda v: (v[None, :, None] * wj[None, None, :] * detm[:, None, None] * thm[:, :, None]).reshape(nEl, nk * n2)
    pert = lambda: 1.0 + 1e-4 * r.standard_normal((nEl, nk * n2))
    rho, rt, exner = col(rho_v) * pert(), col(rho_v * th_v) * pert(), col(pi_v) * pert()
    velz = np.zeros((nEl, (nk - 1) * n2))
    lat = np.ascontiguousarray(P.sq[:, 1][P.elinds("q")])
    udwdx = 1e-3 * r.standard_normal((nEl, (nk - 1) * n2)) * float(np.abs(zv).mean()) / 9.80616 / 1.5e4
    arrays = mesh_arrays(dm)
    arrays.update(dt=np.array([dt]), zv=zv, velz=velz, rho=rho, rt=rt, exner=exner, lat=lat, udwdx=udwdx)
    fin, fout = str(tmp_path / "vert_in.arr"), str(tmp_path / "vert_out.bin")
    write_arrays(fin, arrays)
    out = subprocess.run([_build(str(tmp_path), "test_vert"), fin, fout], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "DONE" in out.stdout
    res = np.fromfile(fout, dtype=np.float64)
    pos = [0]

    def take(shape):
        n = int(np.prod(shape)); a = res[pos[0]:pos[0] + n].reshape(shape); pos[0] += n
        return a
    for its, kw in ((3, dict()), (2, dict(hs_forcing=True, udwdx=udwdx))):
        got = [take(velz.shape), take(rho.shape), take(rt.shape), take(exner.shape)]
        hist = take((its, 4))
        want = vert_oracle.solve_schur_eta(P, dt, velz, rho, rt, exner, zv, its, **kw)
        for a, b, name in zip(got, want[:4], ("velz", "rho", "rt", "exner")):
            assert np.all(np.isfinite(b)) and rel_l2(a, b) < 1e-10, (name, kw.keys())
        for hd, ho in zip(hist, want[4]):
            for j, k in enumerate(("exner", "w", "rho", "eta")):
                assert abs(hd[j] - ho[k]) <= 1e-5 * ho[k] + 1e-15, (k, hd[j], ho[k])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sw_step_driven_from_cpp(tmp_path, oracle, world):
    """N3 on several ranks from C++ (round 6): src::SWEqn over a Shard (mimsem_sweqn.hpp) -- the fixed-length Chebyshev solves with the halo
    exchanges inside, ONE all-reduce of the check norms per Picard iteration, the spectral regions from the host's own all-reduced Arnoldi
    process -- with the ranks as threads of one process on the one GPU (tests/cpp/test_sw_sharded.cpp).  Three Galewsky-style steps against the
    one-context run of the Python host: state to 1e-10, the executable itself asserts the all-reduce count and that no check missed."""
    import numpy as np
    import torch
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.partition import build_plans, patches_of_rank
    from mimsem_amd.sweqn import SWEqn, williamson2
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import mesh_arrays, write_arrays
    pn, ne, npatch, nsteps = 3, 4, 6, 3
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
        return dm, xq[dm.gidq]
    dm1, xq1 = build(list(range(npatch)))
    eng1 = Engine(dm1)
    S1 = SWEqn(eng1, xq1)
    uq, hq = williamson2(torch.as_tensor(xq1, device=eng1.device), alpha=0.0)
    lam = torch.atan2(torch.as_tensor(xq1[:, 1]), torch.as_tensor(xq1[:, 0])).to(eng1.device)
    uq = uq + torch.stack([3.0 * torch.sin(2 * lam), 2.0 * torch.cos(lam)], dim=1)            # perturbed: every term active
    u0, h0 = S1.init1(uq), S1.init2(hq)
    u, h = u0, h0
    for _ in range(nsteps):
        u, h = S1.solve(u, h, 360.0, nits=2, q_exact=False)
    assert S1.fixed_iterations == 2 * nsteps
    fg = S1.fg[0].cpu().numpy(); ug0 = u0[0].cpu().numpy(); hg0 = h0[0].cpu().numpy()
    want_u, want_h = u[0].cpu().numpy(), h[0].cpu().numpy()
    dms = []
    for rank in range(world):
        dm, _ = build(patches_of_rank(npatch, world, rank))
        p0, p1 = build_plans(cs, world, rank, dm.gid0, dm.gid1)
        ranks = p1.neighbours()
        assert ranks == p0.neighbours() and len(ranks) == world - 1
        empty = np.zeros(0, np.int32)

        def lists(by_rank):
            off = np.zeros(len(ranks) + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(by_rank.get(r, empty)) for r in ranks])
            return np.concatenate([by_rank.get(r, empty) for r in ranks]).astype(np.int32), off
        arr = mesh_arrays(dm)
        g1, g1o = lists(p1.ghost_slots); m1, m1o = lists(p1.mirror_slots); g0, g0o = lists(p0.ghost_slots); m0, m0o = lists(p0.mirror_slots)
        arr.update(ranks=np.asarray(ranks, np.int32), ghost1=g1, ghost1_off=g1o, mirror1=m1, mirror1_off=m1o, ghost0=g0, ghost0_off=g0o, mirror0=m0,
                   mirror0_off=m0o, own0=p0.owned.astype(np.float64), own1=p1.owned.astype(np.float64), fg=fg[dm.gid0], u=ug0[dm.gid1], h=hg0[dm.gid2],
                   params=np.array([360.0, 2.0, 0.0]))
        write_arrays(str(tmp_path / ("rank%d.arr" % rank)), arr)
        dms.append(dm)
    out = subprocess.run([_build(str(tmp_path), "test_sw_sharded", ["-pthread"]), str(world), str(tmp_path / "rank"), str(tmp_path / "out"), str(nsteps)],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "DONE" in out.stdout
    got_u = np.full(cs.nDofs1G, np.nan); got_h = np.full(cs.nDofs2G, np.nan)
    for rank, dm in enumerate(dms):
        res = np.fromfile(str(tmp_path / ("out%d.bin" % rank)), dtype=np.float64)
        ul, hl = res[:dm.n1], res[dm.n1:]
        # every copy of a shared DoF agrees with the one-context value (ghosts included)
        assert np.linalg.norm(ul - want_u[dm.gid1]) <= 1e-10 * np.linalg.norm(want_u[dm.gid1]), rank
        got_u[dm.gid1] = ul; got_h[dm.gid2] = hl
    eu = np.linalg.norm(got_u - want_u) / np.linalg.norm(want_u); eh = np.linalg.norm(got_h - want_h) / np.linalg.norm(want_h)
    print("C++ sharded SW step, world %d: |u - u_1ctx| = %.2e  |h - h_1ctx| = %.2e" % (world, eu, eh))
    assert eu < 1e-10 and eh < 1e-11 and np.linalg.norm(want_u - ug0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_horizsolve_driven_from_cpp(tmp_path, oracle, world):
    """N2 on several ranks from C++ (round 6): mimsem_host::HorizSolve over a Shard -- every 0/1-form result completed over the halo, the ksp1
    solves fixed-length Chebyshev iterations with the exchanges inside (the executable asserts: NO all-reduce inside the evaluation, every solve
    checked) -- ranks as threads of one process (tests/cpp/test_horiz_sharded.cpp).  advection_rhs_ec + momentum_rhs_ec (viscosity on) against
    the one-context evaluation of the Python host."""
    import numpy as np
    import torch
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.horizsolve import HorizSolve
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.partition import build_plans, patches_of_rank
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import mesh_arrays, write_arrays, z_levels
    pn, ne, npatch, nk = 3, 4, 6, 3
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
        return dm, xq[dm.gidq]
    dm1, xq1 = build(list(range(npatch)))
    eng1 = Engine(dm1)
    hs1 = HorizSolve(eng1, quad_coords=xq1)
    r = np.random.default_rng(31)
    N0, N1, N2 = cs.nDofs0G, cs.nDofs1G, cs.nDofs2G
    area = float(dm1.det.mean()) * 4.0 / (pn * pn); dz = float(dm1.thick.mean()); ln = area ** 0.5
    G = dict(u1=r.standard_normal((nk, N1)) * 20.0 * ln * dz, h1=r.uniform(0.8, 1.2, (nk, N2)) * area * dz, theta=r.uniform(290, 310, (nk, N2)) * area * dz,
             Pi=r.uniform(900, 1000, (nk, N2)) * area * dz, velz=r.standard_normal((nk - 1, N2)) * area, dudz=r.standard_normal((nk - 1, N1)) * 1e-3 * ln)
    G["u2"] = G["u1"] * 1.03; G["h2"] = G["h1"] * 1.01
    t = lambda k: eng1.tensor(G[k])
    dF, dG, Fk, Gk = hs1.advection_rhs_ec(t("u1"), t("u2"), t("h1"), t("h2"), t("theta"))
    fu = hs1.momentum_rhs_ec(t("theta"), t("dudz"), t("dudz"), t("velz"), t("velz"), t("Pi"), t("u1"), t("u2"), t("h1"), t("h2"), Fx=Fk, Fk=Fk, dTheta=hs1.dTheta)
    want_fu, want_dG, want_k2i = fu.cpu().numpy(), dG.cpu().numpy(), hs1.k2i
    fg = hs1.fg.cpu().numpy()
    fg = fg if fg.shape[0] == nk else np.broadcast_to(fg, (nk, N0))
    dms = []
    for rank in range(world):
        dm, _ = build(patches_of_rank(npatch, world, rank))
        p0, p1 = build_plans(cs, world, rank, dm.gid0, dm.gid1)
        ranks = p1.neighbours()
        assert ranks == p0.neighbours() and len(ranks) == world - 1
        empty = np.zeros(0, np.int32)

        def lists(by_rank):
            off = np.zeros(len(ranks) + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(by_rank.get(q, empty)) for q in ranks])
            return np.concatenate([by_rank.get(q, empty) for q in ranks]).astype(np.int32), off
        arr = mesh_arrays(dm)
        g1, g1o = lists(p1.ghost_slots); m1, m1o = lists(p1.mirror_slots); g0, g0o = lists(p0.ghost_slots); m0, m0o = lists(p0.mirror_slots)
        arr.update(ranks=np.asarray(ranks, np.int32), ghost1=g1, ghost1_off=g1o, mirror1=m1, mirror1_off=m1o, ghost0=g0, ghost0_off=g0o, mirror0=m0,
                   mirror0_off=m0o, own0=p0.owned.astype(np.float64), own1=p1.owned.astype(np.float64), fg=np.ascontiguousarray(fg[:, dm.gid0]),
                   params=np.array([float(N0)]))
        for k, gid in (("u1", dm.gid1), ("u2", dm.gid1), ("dudz", dm.gid1), ("h1", dm.gid2), ("h2", dm.gid2), ("theta", dm.gid2), ("Pi", dm.gid2), ("velz", dm.gid2)):
            arr[k] = np.ascontiguousarray(G[k][:, gid])
        write_arrays(str(tmp_path / ("rank%d.arr" % rank)), arr)
        dms.append(dm)
    out = subprocess.run([_build(str(tmp_path), "test_horiz_sharded", ["-pthread"]), str(world), str(tmp_path / "rank"), str(tmp_path / "out")],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "DONE" in out.stdout
    got_fu = np.full((nk, N1), np.nan); got_dG = np.full((nk, N2), np.nan)
    for rank, dm in enumerate(dms):
        res = np.fromfile(str(tmp_path / ("out%d.bin" % rank)), dtype=np.float64)
        s1, s2 = nk * dm.n1, nk * dm.n2
        fl, gl, k2i = res[:s1].reshape(nk, dm.n1), res[s1:s1 + s2].reshape(nk, dm.n2), res[s1 + s2]
        assert np.linalg.norm(fl - want_fu[:, dm.gid1]) <= 1e-9 * np.linalg.norm(want_fu[:, dm.gid1]), rank          # ghosts included
        assert abs(k2i - want_k2i) <= 1e-9 * abs(want_k2i), (rank, k2i, want_k2i)
        got_fu[:, dm.gid1] = fl; got_dG[:, dm.gid2] = gl
    e1 = np.linalg.norm(got_fu - want_fu) / np.linalg.norm(want_fu); e2 = np.linalg.norm(got_dG - want_dG) / np.linalg.norm(want_dG)
    print("C++ sharded HorizSolve, world %d: |fu - fu_1ctx| = %.2e  |dG - dG_1ctx| = %.2e" % (world, e1, e2))
    assert e1 < 1e-9 and e2 < 1e-9
