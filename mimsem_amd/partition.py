"""Sharding of cubed-sphere patches over GPUs and the halo exchange that replaces the reference's
VecScatter gtol_0 / gtol_1 (eul/Topo.cpp:145-155).

One process per GPU.  The 6*npx^2 patches of the reference decomposition are dealt contiguously to the
ranks (24 patches -> 24/12/6/3 per GPU for 1/2/4/8 GPUs, so 8 GPUs are legal although 8 MPI ranks are not,
SURVEY F9).  Inside a rank all its patches share one compacted numbering, so patch boundaries interior to a
GPU need no communication at all; only slots whose OWNER patch lives on another rank are exchanged.

Ownership follows the reference: a patch owns its west/south-inclusive nodes and edges, ghosts sit on its
east/north side (scr/Proc2.py:89-130); the two hanging nodes are owned by face 0's SE-corner patch and
face 1's NW-corner patch.  2-forms never communicate.

Transport of HaloExchanger: ONE torch.distributed all_to_all_single per exchange (backend "nccl" = RCCL over xGMI on a GPU node: a
grouped send/recv; "gloo" in the CPU tests and in the several-ranks-on-one-GPU rehearsals, staged through host memory).  Messages
are kB-sized and latency bound, so all levels of a field travel in ONE message per neighbour rank.  Pack/unpack run on the device
through the C ABI (mimsem_halo_segments).  The device-tensor RCCL branch of HaloExchanger has NOT run on hardware yet (no multi-GPU
box in development; tests/test_gpu_nccl_world2.py runs it when two GPUs are visible).  CHalo is the same exchange driven entirely
by the C ABI (mimsem_halo_create / _begin / _end) -- its RCCL transport has run on one rank (self send/recv), its host-callback
transport on 2 and 3 ranks.
"""
import functools
import os

import numpy as np
import torch
import torch.distributed as dist

from .topo import Topo


def _cut(adj, part):
    """number of patch adjacencies (weighted by shared edges) that cross parts"""
    n = len(part)
    return int(sum(adj[i, j] for i in range(n) for j in range(i + 1, n) if part[i] != part[j]))


@functools.lru_cache(maxsize=None)
def rank_of_patch(n_patches, world, layout=None):
    """patch -> rank map (tuple of length n_patches), equal patch counts per rank.

    "contiguous": ranges of patch ids (24 patches -> 24/12/6/3 per GPU) -- whole faces for 1/2/3/6 ranks, but for 4 and 8 ranks the
    ranges straddle faces (8 ranks: patches {3, 4, 5} = one corner of face 0 and the bottom row of face 1).
    "compact" (default): geometry-aware -- parts grown over the patch adjacency graph of the cubed sphere (a patch joins the part
    it shares the most edges with), kept only when it cuts fewer patch adjacencies than the contiguous ranges, i.e. fewer halo
    slots and fewer neighbour ranks per GPU.  MIMSEM_PATCH_MAP=contiguous selects the ranges."""
    if n_patches % world:
        raise ValueError(f"{n_patches} patches do not divide over {world} ranks")
    per = n_patches // world
    contiguous = tuple(p // per for p in range(n_patches))
    layout = layout or os.environ.get("MIMSEM_PATCH_MAP", "compact")
    npx = int(round((n_patches / 6.0) ** 0.5)) if n_patches % 6 == 0 else 0
    if layout == "contiguous" or world == 1 or per == 1 or 6 * npx * npx != n_patches:
        return contiguous
    from .mesh import CubedSphere
    cs = CubedSphere(1, npx, n_patches)                     # one element per patch: its four edges are the patch adjacencies
    owners = {}
    for p in cs.patches:
        for g in np.unique(p.loc1):
            owners.setdefault(int(g), []).append(p.pid)
    adj = np.zeros((n_patches, n_patches), dtype=np.int64)
    for ps in owners.values():
        for a in ps:
            for b in ps:
                if a != b:
                    adj[a, b] += 1
    part = [-1] * n_patches
    for r in range(world):
        seed = min(p for p in range(n_patches) if part[p] < 0)
        members = [seed]; part[seed] = r
        while len(members) < per:
            # the free patch sharing the most edges with the part; among equals the one with the fewest free neighbours (leaves
            # no orphans behind), then the lowest id
            cand = [(-int(adj[q, members].sum()), int(sum(1 for t in range(n_patches) if adj[q, t] and part[t] < 0)), q)
                    for q in range(n_patches) if part[q] < 0]
            q = min(cand)[2]
            members.append(q); part[q] = r
    return tuple(part) if _cut(adj, part) < _cut(adj, contiguous) else contiguous


def patches_of_rank(n_patches, world, rank):
    m = rank_of_patch(n_patches, world)
    return [p for p in range(n_patches) if m[p] == rank]


def owner_tables(sphere):
    """owner patch of every global node / edge id (int32 arrays of length nDofs0G, nDofs1G)"""
    D = sphere.D
    own0 = np.full(sphere.nDofs0G, -1, dtype=np.int32)
    own1 = np.full(sphere.nDofs1G, -1, dtype=np.int32)
    for p in sphere.patches:
        l0 = p.loc0.reshape(D + 1, D + 1)
        own0[l0[:D, :D].ravel()] = p.pid
        own1[p.loc1x.reshape(D, D + 1)[:, :D].ravel()] = p.pid
        own1[p.loc1y.reshape(D + 1, D)[:D, :].ravel()] = p.pid
    npx = sphere.npx
    own0[sphere.hang0] = 0 * npx * npx + 0 * npx + (npx - 1)          # face 0, SE-corner patch
    own0[sphere.hang1] = 1 * npx * npx + (npx - 1) * npx + 0          # face 1, NW-corner patch
    assert (own0 >= 0).all() and (own1 >= 0).all()
    return own0, own1


class HaloPlan:
    """send/recv slot lists of one form for one rank; both sides order every list by global id"""

    def __init__(self, gids, owner_patch, world, rank, n_patches):
        rmap = np.asarray(rank_of_patch(n_patches, world), dtype=np.int64)
        owner_rank = rmap[owner_patch[gids]]
        self.rank, self.world = rank, world
        self.owned = owner_rank == rank
        # ghosts I hold, grouped by owning rank (gids are sorted, so each group is sorted by gid)
        self.ghost_slots = {int(r): np.nonzero(owner_rank == r)[0].astype(np.int32)
                            for r in np.unique(owner_rank) if r != rank}
        self.ghost_gids = {r: gids[s] for r, s in self.ghost_slots.items()}
        self.mirror_slots = {}      # filled by exchange_layout(): my owned slots that rank r holds as ghosts
        self.gids = gids

    def neighbours(self):
        return sorted(set(self.ghost_slots) | set(self.mirror_slots))


def build_plans(sphere, world, rank, gid0, gid1):
    """HaloPlans for 0- and 1-forms.  mirror lists are derived locally (every rank can compute every
    other rank's ghost set from the global mesh), so plan construction needs no communication."""
    own0, own1 = owner_tables(sphere)
    n_patches = len(sphere.patches)
    plans = []
    for form, gids, owner in ((0, gid0, own0), (1, gid1, own1)):
        plan = HaloPlan(gids, owner, world, rank, n_patches)
        plan.form = form
        rmap = np.asarray(rank_of_patch(n_patches, world), dtype=np.int64)
        for r in range(world):
            if r == rank:
                continue
            # global ids touched by rank r's patches
            pids = patches_of_rank(n_patches, world, r)
            if form == 0:
                touched = np.unique(np.concatenate([sphere.patches[p].loc0 for p in pids]))
            else:
                touched = np.unique(np.concatenate([sphere.patches[p].loc1 for p in pids]))
            mine = touched[rmap[owner[touched]] == rank]            # sorted by gid
            if mine.size:
                plan.mirror_slots[r] = np.searchsorted(gids, mine).astype(np.int32)
        plans.append(plan)
    return plans


class HaloExchanger:
    """REVERSE/ADD (partial sums at ghost slots -> owner adds) and FORWARD/INSERT (owner -> ghosts).

    One message per neighbour rank carrying every level; the messages of all neighbours live in ONE send and ONE receive
    buffer (segment-major [rank][level][slot]) that are packed / unpacked by a single kernel launch and travel in a single
    `all_to_all_single` (RCCL: one grouped send/recv launch).  Buffers are cached per level count.  `engine` supplies the
    device pack/unpack (mimsem_halo_segments); with engine=None plain torch indexing does the same (CPU/gloo tests).
    An ADD whose target slots repeat between neighbours (cube-corner nodes) is unpacked in rank-ordered ranges without
    repeats, so the sums are formed in a fixed order."""

    def __init__(self, plan, engine=None, device=None):
        self.plan, self.engine = plan, engine
        self.device = device if device is not None else (engine.device if engine is not None else torch.device("cpu"))
        self.world = plan.world
        self.sides = {"ghost": self._side(plan.ghost_slots), "mirror": self._side(plan.mirror_slots)}
        # every slot this rank shares with rank r (its ghosts owned by r and its own slots that r holds as ghosts), ordered by
        # global id on both sides: the lists of the symmetric exchange sum_all()
        pair = {}
        for r in sorted(set(plan.ghost_slots) | set(plan.mirror_slots)):
            sl = np.concatenate([plan.ghost_slots.get(r, np.zeros(0, np.int32)), plan.mirror_slots.get(r, np.zeros(0, np.int32))])
            pair[r] = np.sort(sl).astype(np.int32)          # local slots are numbered in gid order, so this is the gid order too
        self.sides["pair"] = self._side(pair)
        allp = np.concatenate(list(pair.values())) if pair else np.zeros(0, np.int32)
        # the one-exchange form needs every shared slot to have exactly one partner rank ON EVERY RANK (the ranks must agree on
        # which collective they run): guaranteed for edges (an edge borders at most two patches), not for nodes
        self.pairwise = getattr(plan, "form", None) == 1
        if self.pairwise:
            assert np.unique(allp).size == allp.size
        self._bufs = {}
        self._plans = {}
        self._stage_host = None

    def _side(self, slots_by_rank):
        ranks = sorted(slots_by_rank)
        counts = np.zeros(self.world, dtype=np.int64)
        for r in ranks:
            counts[r] = len(slots_by_rank[r])
        off = np.zeros(len(ranks) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(slots_by_rank[r]) for r in ranks])
        cat = np.concatenate([slots_by_rank[r] for r in ranks]).astype(np.int32) if ranks else np.zeros(0, dtype=np.int32)
        # ranges of neighbours whose slot sets are pairwise disjoint (greedy, rank order)
        ranges, seen, start = [], set(), 0
        for i, r in enumerate(ranks):
            sl = set(int(x) for x in slots_by_rank[r])
            if seen & sl:
                ranges.append((start, i)); start, seen = i, set()
            seen |= sl
        if ranks:
            ranges.append((start, len(ranks)))
        return dict(ranks=ranks, counts=counts, off=off, ranges=ranges,
                    idx=torch.as_tensor(cat, dtype=torch.int32, device=self.device),
                    idx_long=torch.as_tensor(cat, dtype=torch.long, device=self.device))

    def _buffers(self, key, nlev, dtype, role="send"):
        k = (key, nlev, dtype, role)               # send and receive sides never share storage (sum_all uses one list for both)
        if k not in self._bufs:
            total = int(self.sides[key]["off"][-1])
            self._bufs[k] = torch.zeros(max(total * nlev, 1), dtype=dtype, device=self.device)
        return self._bufs[k]

    def _move(self, side, mode, buf, v2, s_begin=None, s_end=None):
        """mode 0: v -> buf (pack); 1: buf -> v (insert); 2: v += buf"""
        nseg = len(side["ranks"])
        s_begin = 0 if s_begin is None else s_begin
        s_end = nseg if s_end is None else s_end
        if nseg == 0 or s_begin == s_end:
            return
        nlev = v2.shape[0]
        if self.engine is not None:
            self.engine.halo_segments(side["idx"], side["off"], s_begin, s_end, mode, buf, v2)
            return
        off = side["off"]
        for s in range(s_begin, s_end):
            a, b = int(off[s]), int(off[s + 1])
            seg = buf[a * nlev:b * nlev].view(nlev, b - a)
            ix = side["idx_long"][a:b]
            if mode == 0:
                seg.copy_(v2[:, ix])
            elif mode == 1:
                v2[:, ix] = seg
            else:
                v2[:, ix] += seg

    def _exchange(self, v, send_key, recv_key, add):
        v2 = v if v.dim() == 2 else v.unsqueeze(0)
        nlev = v2.shape[0]
        snd, rcv = self.sides[send_key], self.sides[recv_key]
        key = (send_key, recv_key, nlev, v2.dtype)
        plan = self._plans.get(key)
        if plan is None:                          # buffers, views and split lists are fixed per (direction, level count): build once
            sbuf, rbuf = self._buffers(send_key, nlev, v2.dtype, "send"), self._buffers(recv_key, nlev, v2.dtype, "recv")
            ns, nr = int(snd["off"][-1]) * nlev, int(rcv["off"][-1]) * nlev
            plan = (sbuf, rbuf, sbuf[:ns], rbuf[:nr], (rcv["counts"] * nlev).tolist(), (snd["counts"] * nlev).tolist())
            self._plans[key] = plan
        sbuf, rbuf, sview, rview, out_splits, in_splits = plan
        self._move(snd, 0, sbuf, v2)
        if self._stage_host is None:
            self._stage_host = sview.is_cuda and dist.get_backend() == "gloo"
        if self._stage_host:                      # gloo moves host memory only: rehearsals of the device path on CPU transport
            rh = torch.empty(rview.shape, dtype=rview.dtype)
            dist.all_to_all_single(rh, sview.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
            rview.copy_(rh)
        else:
            dist.all_to_all_single(rview, sview, output_split_sizes=out_splits, input_split_sizes=in_splits)
        if add:
            for (a, b) in rcv["ranges"]:                 # fixed order => reproducible sums
                self._move(rcv, 2, rbuf, v2, a, b)
        else:
            self._move(rcv, 1, rbuf, v2)

    def sum_all(self, v):
        """every copy of a shared slot := the sum of all its partial values, in ONE exchange: each rank sends its partial sums to
        the other sharer and adds what it receives (a + b on one side, b + a on the other: bitwise identical).  Needs every shared
        slot to have exactly one partner rank -- true for 1-forms (an edge borders at most two patches); otherwise falls back
        to reverse_add + forward_insert."""
        if not self.pairwise:
            self.reverse_add(v); self.forward_insert(v)
            return
        self._exchange(v, "pair", "pair", add=True)

    def reverse_add(self, v):
        """VecScatter(gtol, vl, vg, ADD_VALUES, SCATTER_REVERSE): owners accumulate the ghosts' partial sums"""
        self._exchange(v, "ghost", "mirror", add=True)

    def forward_insert(self, v):
        """VecScatter(gtol, vg, vl, INSERT_VALUES, SCATTER_FORWARD): ghosts receive the owners' values"""
        self._exchange(v, "mirror", "ghost", add=False)


def rank_mesh(sphere, geoms_or_none, world, rank, nk, coords=None):
    """(topos, geoms) of the patches this rank holds; geometry is built locally from the coordinate table"""
    from .geom import Geom
    pids = patches_of_rank(len(sphere.patches), world, rank)
    topos = [Topo(sphere, p, nk) for p in pids]
    geoms = [Geom(t, sphere, coords, nk) for t in topos] if geoms_or_none is None else geoms_or_none
    return pids, topos, geoms


class CHalo:
    """The same exchanges through the C ABI (mimsem_halo_create / _begin / _end, include/mimsem_hip.h): what a C++ host binds instead
    of VecScatterBegin/End.  Pack, transport and unpack are driven by the library; this class only builds the slot lists from a
    HaloPlan and supplies a transport:
      "dist"      a host callback over torch.distributed, staged through host memory (how a host with a plain MPI would do it;
                  the test rehearsal of several ranks on one GPU over gloo uses it),
      "loopback"  every neighbour is the rank itself (single-rank tests),
      an int      an ncclComm_t of the caller: grouped ncclSend/ncclRecv on the plan's communication stream (xGMI),
      "peer"      ONE-SIDED (round 6): the receive buffers are exported through hipIpc and opened by the neighbour ranks; the pack kernel
                  writes straight into the neighbour's buffer and publishes a sequence flag, the unpack waits for the flags -- kernels only
                  (mimsem_halo_peer_export / _set_peer; the handles travel once, over the process group, at construction).
    begin()/end() are split so that interior work can be enqueued in between."""

    def __init__(self, plan, engine, max_nlev, transport="dist"):
        import ctypes as C
        from ._lib import HALO_TRANSPORT, check
        self.eng, self.L, self.C = engine, engine.L, C
        self._check = check
        nslots = len(plan.gids)
        ranks = plan.neighbours()
        empty = np.zeros(0, np.int32)

        def lists(by_rank):
            off = np.zeros(len(ranks) + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(by_rank.get(r, empty)) for r in ranks])
            idx = np.concatenate([by_rank.get(r, empty) for r in ranks]).astype(np.int32) if ranks else empty
            return np.ascontiguousarray(idx), off
        gi, go = lists(plan.ghost_slots)
        mi, mo = lists(plan.mirror_slots)
        self.ranks = np.asarray(ranks, dtype=np.int32)
        # the symmetric exchange of sum_all: every slot shared with a rank, in gid order on both sides (as HaloExchanger's "pair")
        pair = {r: np.sort(np.concatenate([plan.ghost_slots.get(r, empty), plan.mirror_slots.get(r, empty)])).astype(np.int32) for r in ranks}
        pi, po = lists(pair)
        self.shared = np.unique(pi)
        self._keep = (gi, go, mi, mo, pi, po)
        self.handles = {}
        for name, (si, so, ri, ro) in (("reverse", (gi, go, mi, mo)), ("forward", (mi, mo, gi, go)), ("pair", (pi, po, pi, po))):
            h = C.c_void_p()
            check(self.L.mimsem_halo_create(engine.ctx, len(ranks), self.ranks.ctypes.data, si.ctypes.data, so.ctypes.data,
                                            ri.ctypes.data, ro.ctypes.data, nslots, max_nlev, C.byref(h)), "halo_create")
            self.handles[name] = h
        if transport == "peer":
            from ._lib import HALO_PEER_BLOB
            me = dist.get_rank()
            mine = {}
            for name, h in self.handles.items():
                blob = C.create_string_buffer(HALO_PEER_BLOB)
                check(self.L.mimsem_halo_peer_export(h, me, blob), "halo_peer_export")
                mine[name] = bytes(blob.raw)
            everyone = [None] * dist.get_world_size()
            dist.all_gather_object(everyone, mine)                     # (the one collective of this transport: the handles, once)
            for name, h in self.handles.items():
                blobs = b"".join(everyone[int(r)][name] for r in ranks)
                check(self.L.mimsem_halo_set_peer(h, me, blobs if blobs else None), "halo_set_peer")
        elif transport == "loopback":
            for h in self.handles.values():
                check(self.L.mimsem_halo_set_loopback(h), "halo_set_loopback")
        elif transport == "dist":
            self._cb = HALO_TRANSPORT(self._dist_transport)          # keep the callback object alive
            for h in self.handles.values():
                check(self.L.mimsem_halo_set_transport(h, C.cast(self._cb, C.c_void_p), None), "halo_set_transport")
        else:
            for h in self.handles.values():
                check(self.L.mimsem_halo_set_rccl(h, C.c_void_p(int(transport))), "halo_set_rccl")

    def _dist_transport(self, user, send, send_off, recv, recv_off, nneigh, ranks, stream):
        """host-staged exchange: wait for the pack, copy out, all_to_all over the process group, copy in"""
        try:
            C = self.C
            torch.cuda.ExternalStream(stream).synchronize()
            world = dist.get_world_size()
            ns, nr = send_off[nneigh], recv_off[nneigh]
            sh = np.empty(max(ns, 1)); rh = np.empty(max(nr, 1))
            if ns:
                self._check(self.L.mimsem_memcpy_d2h(self.eng.ctx, sh.ctypes.data, send, ns * 8), "memcpy_d2h")
            ins, outs = [0] * world, [0] * world
            for i in range(nneigh):
                ins[ranks[i]] = send_off[i + 1] - send_off[i]
                outs[ranks[i]] = recv_off[i + 1] - recv_off[i]
            # neighbours are listed in rank order, so the segment order IS the all_to_all order
            dist.all_to_all_single(torch.from_numpy(rh[:nr]), torch.from_numpy(sh[:ns]), output_split_sizes=outs, input_split_sizes=ins)
            if nr:
                self._check(self.L.mimsem_memcpy_h2d(self.eng.ctx, recv, rh.ctypes.data, nr * 8), "memcpy_h2d")
            return 0
        except Exception:                                            # noqa: BLE001 -- reported to the library as a failed transport
            import traceback
            traceback.print_exc()
            return 1

    def begin(self, name, v, add):
        v2 = v if v.dim() == 2 else v.unsqueeze(0)
        self._check(self.L.mimsem_halo_begin(self.handles[name], 1 if add else 0, v2.shape[0], v2.data_ptr(), v2.stride(0)), "halo_begin")
        return name

    def end(self, name):
        self._check(self.L.mimsem_halo_end(self.handles[name]), "halo_end")

    def reverse_add(self, v):
        self.end(self.begin("reverse", v, True))

    def forward_insert(self, v):
        self.end(self.begin("forward", v, False))

    def sum_all(self, v):
        """1-forms: every sharer sends its partial sums to the other and adds what it receives (a + b = b + a bitwise)"""
        self.end(self.begin("pair", v, True))

    def peer_timeouts(self):
        """one-sided transport: sequence numbers of exchanges whose wait gave up (a neighbour that never published), per plan; {} = none"""
        C = self.C
        out = {}
        for name, h in self.handles.items():
            v = C.c_ulonglong(0)
            self._check(self.L.mimsem_halo_peer_status(h, C.byref(v)), "halo_peer_status")
            if v.value:
                out[name] = v.value
        return out

    def close(self):
        for h in self.handles.values():
            self.L.mimsem_halo_destroy(h)
        self.handles = {}
