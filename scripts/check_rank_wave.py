"""Does a rank's sub-mesh (compacted global numbering of its patches, partition.rank_mesh) get the wave-level plan?  Prints the
plan statistics for rank 0 of world = 1, 2, 4, 8 on the config-4 sphere (p = 3, 24 x 24 x 6 elements, 24 patches)."""
import ctypes as C
import sys

from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.partition import rank_mesh

pn, ne, npr, nk = 3, 24, 24, 8
cs = CubedSphere(pn, ne, npr); coords = sphere_coords(pn, ne)
for world in (1, 2, 3, 4, 6, 8, 12, 24):
    for rank in sorted({0, world - 1}):
        pids, topos, geoms = rank_mesh(cs, None, world, rank, nk, coords)
        eng = Engine(DeviceMesh(topos, geoms, nk=nk, numbering="global"))
        st = (C.c_int * 5)()
        has = eng.L.mimsem_op_wave_stats(eng.ctx, nk, st)
        print("world %2d rank %2d: %2d patches, wave plan %d %s" % (world, rank, len(pids), has, list(st)), flush=True)
        del eng
