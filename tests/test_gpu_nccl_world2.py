"""The device-tensor RCCL branch of HaloExchanger (all_to_all_single on GPU buffers over xGMI), the C ABI's RCCL transport
(mimsem_halo_set_rccl on a communicator made by RcclComm, with the boundary | exchange | interior split) and bench.py's N = 2 launch: needs TWO
visible GPUs, so it is skipped on the one-GPU test box (where the same control flow runs over gloo, tests/test_gpu_multiproc.py) and
runs wherever a multi-GPU node executes the suite."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ngpu():
    import torch
    return torch.cuda.device_count()


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from mimsem_amd.device import DeviceMesh, Engine
        from mimsem_amd.geom import Geom
        from mimsem_amd.mesh import CubedSphere, sphere_coords
        from mimsem_amd.partition import HaloExchanger, build_plans, patches_of_rank
        from mimsem_amd.topo import Topo
        from mimsem_amd.workloads import SCALE, z_levels
        pn, ne, npatch, nk = 3, 4, 24, 3
        cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)
        xg = np.random.default_rng(123).standard_normal((nk, cs.nDofs1G))

        def build(pids, dev):
            topos = [Topo(cs, p, nk) for p in pids]
            geoms = [Geom(t, cs, coords, nk) for t in topos]
            for g in geoms:
                g.set_levels(z_levels(nk, g.n0))
            dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
            return dm, Engine(dm, device=dev)
        dm, eng = build(patches_of_rank(npatch, world, rank), rank)
        plan1 = build_plans(cs, world, rank, dm.gid0, dm.gid1)[1]
        y = eng.apply("UMAT", eng.tensor(xg[:, dm.gid1]), lev0=0, scale=SCALE, flags=1)
        halo = HaloExchanger(plan1, engine=eng)
        halo.reverse_add(y); halo.forward_insert(y)
        dm1, eng1 = build(list(range(npatch)), rank)
        want = eng1.apply("UMAT", eng1.tensor(xg), lev0=0, scale=SCALE, flags=1).cpu().numpy()
        err = np.linalg.norm(y.cpu().numpy() - want[:, dm.gid1]) / np.linalg.norm(want)
        ok = bool(err < 1e-12)
        # the C ABI's own transport: an ncclComm_t made as a C++ host would (RcclComm), mimsem_halo_set_rccl, grouped ncclSend/ncclRecv on
        # the plan's stream, the apply split boundary | exchange | interior -- bit for bit the exchanger's result
        from mimsem_amd.distributed import DistEngine
        dm2, eng2 = build(patches_of_rank(npatch, world, rank), rank)
        de = DistEngine(eng2, cs, world, rank, overlap=True, transport="auto")
        ok = ok and de.transport == "rccl"
        y3 = de.apply("UMAT", eng2.tensor(xg[:, dm2.gid1]), lev0=0, scale=SCALE, flags=1)
        ok = ok and bool(torch.equal(y3, y))
        x0 = np.random.default_rng(5).standard_normal((nk, cs.nDofs0G))
        p3 = de.apply("PMAT", eng2.tensor(x0[:, dm2.gid0]), lev0=0, scale=SCALE, flags=0)          # 0-forms: reverse_add + forward_insert plans
        p1 = eng1.apply("PMAT", eng1.tensor(x0), lev0=0, scale=SCALE, flags=0).cpu().numpy()
        ok = ok and bool(np.linalg.norm(p3.cpu().numpy() - p1[:, dm2.gid0]) / np.linalg.norm(p1) < 1e-12)
        de.close()
        q.put((rank, ok))
    except Exception as exc:                            # report instead of leaving the parent to time out
        q.put((rank, "%s: %s" % (type(exc).__name__, exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_halo_over_rccl_two_gpus():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:                                            # a rank that raised leaves its peer waiting in a collective: end exactly these two
        for p in procs:
            if p.is_alive():
                p.terminate(); p.join(timeout=10)
            if p.is_alive():
                p.kill()
    assert all(ok is True for _, ok in res), res
