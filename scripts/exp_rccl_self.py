#!/usr/bin/env python3
"""mimsem_halo_set_rccl on ONE rank: a communicator of size 1 (ncclCommInitRank through ctypes on the librccl PyTorch already loaded)
and a plan whose neighbour is the rank itself -- the grouped ncclSend/ncclRecv path of the C ABI executed once on hardware."""
import ctypes as C, os, sys, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.partition import CHalo
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.geom import Geom
from mimsem_amd.topo import Topo
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # makes torch load its librccl
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
try:
    R = C.CDLL("librccl.so", mode=os.RTLD_NOLOAD | os.RTLD_NOW)
except OSError:
    R = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
class UID(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UID()
assert R.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
rc = R.ncclCommInitRank(C.byref(comm), 1, uid, 0)
assert rc == 0, rc
print("comm", comm.value, flush=True)
cs = CubedSphere(3, 4, 6); coords = sphere_coords(3, 4)
topo = Topo(cs, 2, 2); geom = Geom(topo, cs, coords, 2); geom.set_levels(np.stack([np.zeros(geom.n0), np.ones(geom.n0), 2*np.ones(geom.n0)]))
eng = Engine(DeviceMesh([topo], [geom], nk=2, numbering="local"))
n = eng.sizes[1]
p = types.SimpleNamespace(gids=np.arange(n), ghost_slots={0: np.arange(0, 30, dtype=np.int32)}, mirror_slots={0: np.arange(100, 130, dtype=np.int32)})
p.neighbours = lambda: [0]
h = CHalo(p, eng, max_nlev=2, transport=comm.value)
v0 = np.random.default_rng(0).standard_normal((2, n)); v = eng.tensor(v0)
h.reverse_add(v); torch.cuda.synchronize()
want = v0.copy(); want[:, 100:130] += v0[:, 0:30]
print("rccl self exchange ok:", np.array_equal(v.cpu().numpy(), want), flush=True)
h.forward_insert(v); torch.cuda.synchronize()
want[:, 0:30] = want[:, 100:130]
print("forward ok:", np.array_equal(v.cpu().numpy(), want), flush=True)
h.close()
R.ncclCommDestroy.argtypes = [C.c_void_p]; R.ncclCommDestroy(comm)
dist.destroy_process_group()
