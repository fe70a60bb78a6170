#!/usr/bin/env python3
"""bench.py -- element operator-applies/s of the MiMSEM hot path on MI355X (BASELINE.json metric).

A "step" = one application of the 1-form mass operator (Umat, SURVEY row B1: gather -> interpolate ->
Jacobian/thickness-weighted scale -> project -> deterministic scatter-add) to EVERY (element, level) pair of
the p=3, 24x24x6 cubed sphere with 30 levels (BASELINE config 4 grid, 103 680 pairs), inputs resident in HBM.
N>1: the 24 patches (12x12 elements) are dealt to the ranks (strong scaling) and each step ends with the
halo reduce (RCCL send/recv over xGMI) that replaces the reference's VecScatter REVERSE/ADD.

At N = 1 the same line also carries "sw": shallow-water time steps/s (the second half of BASELINE's metric; --no-sw skips it)
and "column": HEVI column Schur solves/s on the same grid (--no-column skips it).
Prints ONE compact JSON line (rank 0, < 4 KB: compact_record -- the contract's keys, roofline, roofline_cold, cpu_baseline(+_column,
+_sw) and one-number summaries); the full object with every extra goes to bench_extras.json (repo root and gpurun_out/).
`python bench.py --gpus N` started without WORLD_SIZE launches its own N ranks as a child torch.distributed.run.  roofline: dominant kernel k_elem_apply<3,UMAT>, HIP-event timed inside the timed
region, against the launch's COMPULSORY bytes (b1_launch_bytes: frac <= 1 by construction; SURVEY 8(d)'s per-unit
figure rides along as `algorithmic_reference` only).  The headline workload is Infinity-Cache resident
(`cache_resident: true`); `roofline_cold` repeats the step on 8 independent spheres (829 440 units, ~1 GB working
set) -- that one is the HBM statement.  cpu_baseline: the oracle's reference-structure (assemble CSR + SpMV) path on
host cores; cpu_baseline_column: the oracle's solve_schur_column_eta column by column; cpu_baseline_sw: the assemble + MatMult work of one
shallow-water Picard iteration (Krylov solves excluded: an upper bound of the CPU's steps/s).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PN, NE, NPATCH, NK = 3, 24, 24, 30
SCALE = 1.0e8
# SURVEY 8(d)'s per-unit figures, p=3: they charge the level-invariant metric (J, det: 640 B) to EVERY (element, level) unit
# although one launch shares it between all levels of an element -- kept as `algorithmic_reference` only (round-1 VERDICT: a
# fraction computed from them exceeds 1 and is not a roofline fraction).
BYTES_OP_B1 = 1440          # whole B1 apply incl. y read-modify-write and 96 B of indices
BYTES_K1_B1 = 1248          # the element kernel alone: 120 dbl in + 24 dbl out + 96 B indices
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: FP64 vector (= matrix) peak
# executed FP64 work of the column solves per (column, level), p = 3, nk = 30: SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 of every kernel of the
# solve x 64 lanes (FMA = 2 flop), rocprofv3 --pmc over scripts/prof_column.py / prof_column3.py (profiles/r02_column_pmc.txt,
# r03_column_pmc.txt; round 4: profiles/r04_column_pmc_eta.txt, r04_column_pmc_s3.txt -- the three-launch schur_3 executes 1.66e5 where
# round 3's 30 launches executed 2.63e5).  Padding lanes of the 16-lane DPP rows (9 of 16 rows carry data) are executed work and
# count: the figure is what the ALUs did, not the algorithm's minimum (~3 700 flop per level).
SCHUR_ETA_FLOP_PER_COLUMN_LEVEL = 1.45e5
SCHUR_3_FLOP_PER_COLUMN_LEVEL = 1.66e5


def b1_launch_bytes(nEl, n1, nlev, lch, pn=PN):
    """Bytes ONE launch pair of the B1 (Umat) apply must move, from the mesh sizes.
    compulsory: every byte once per launch -- the x vector (n1 doubles per level; an edge shared by two elements is ONE value),
                thickInv per unit, the element-local results written per unit, metric + determinant + gather slots once per element.
                This is what `roofline.achieved` is computed from: it cannot exceed the HBM peak.
    requested:  what the kernel's loads ask of the memory system: x gathered per element (24 doubles per unit), metric / determinant /
                slots once per (element, level chunk of `lch`); the excess over compulsory is served by L2 / Infinity Cache or re-read."""
    mp12, n1e = (pn + 1)**2, pn*(pn + 1)
    units, nchunk = nEl*nlev, -(-nlev//lch)
    geom = (4*mp12 + mp12)*8 + 2*n1e*4                      # J, det, i1x + i1y per element
    k1_c = nlev*n1*8 + units*mp12*8 + units*2*n1e*8 + nEl*geom
    k1_r = units*2*n1e*8 + units*mp12*8 + units*2*n1e*8 + nEl*nchunk*geom
    k2 = units*2*n1e*8 + nlev*n1*8 + n1*2*4                # pass 2: element-local results read, y written, plan (2 ints per slot)
    return {"k1_compulsory": k1_c, "k1_requested": k1_r, "k2": k2, "units": units}


def b1_wave_bytes(nEl, n1, nlev, st, pn=PN):
    """Bytes of the B1 (Umat) apply in its wave-level fused form (k_apply_wave + k_wave_perim, the default for orders <= 4), from the
    mesh sizes and the plan statistics `st` = mimsem_op_wave_stats: [wave-groups, slots written straight into y, partial sums per
    level, slots of the perimeter pass, levels per work item].
    k1_compulsory: what the element kernel MUST move: x once per level, thickInv per unit, the metric record (32 B per point) once per
                element, its two lane tables once per wave-group, and the y slots it completes.  The partial sums it leaves for the
                perimeter pass are this design's overhead, not compulsory: they only count under `requested`.
    op_compulsory: the whole operator: every input once, every output once (x, thickInv, metric, y)."""
    mp12 = (pn + 1)**2
    ng, ndirect, npart, nps, lch = st
    units, nchunk = nEl*nlev, -(-nlev//lch)
    metric, tables = ng*64*32, ng*2*64*16
    k1_c = nlev*n1*8 + units*mp12*8 + metric + tables + nlev*ndirect*8
    k1_r = nlev*ng*64*16 + units*mp12*8 + nchunk*(metric + tables) + nlev*(ndirect + npart)*8
    k2 = nlev*(npart*8 + nps*8) + nps*16                # partials read, y slots written, records
    return {"k1_compulsory": k1_c, "k1_requested": k1_r, "k2": k2, "units": units,
            "op_compulsory": nlev*n1*8*2 + units*mp12*8 + nEl*mp12*32}


def launch_bytes(eng, dmesh, nlev):
    """byte model of the B1 apply for whichever form the context runs, and the name of its dominant kernel"""
    import ctypes as C
    st = (C.c_int*5)()
    if eng.L.mimsem_op_wave_stats(eng.ctx, nlev, st) == 1:
        return b1_wave_bytes(dmesh.nEl, dmesh.n1, nlev, list(st)), "k_apply_wave<3,UMAT>", "k_wave_perim", int(st[4])
    lch = eng.L.mimsem_op_level_chunk(eng.ctx, nlev)
    bm = b1_launch_bytes(dmesh.nEl, dmesh.n1, nlev, lch)
    bm["op_compulsory"] = bm["k1_compulsory"] + bm["k2"]
    return bm, "k_elem_apply<3,UMAT>", "k_gather_sum<2>", lch


def measure_pmc_traffic(timeout=240):
    """HBM-side bytes per launch of the step's kernels, measured IN THIS RUN: two child `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE --
    separately, with --kernel-trace only, the program itself after `--`, as MI355X_MICROARCH.md prescribes) over
    scripts/pmc_traffic.py: a calibration launch of known byte count (k_halo_pack over an identity index list: pins the gfx950
    FETCH_SIZE x2 correction) and the same Umat step, cache-resident and on 8 spheres.  The children are separate processes (this one
    keeps its GPU context and is idle meanwhile).  Returns {kernel: [{grid_threads, read_bytes, write_bytes, total_bytes}, ...]} or
    raises; ~25 s."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="mimsem_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                                sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py")],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:]))
            rows = collections.defaultdict(list)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter:
                        rows[(row["Kernel_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]) * 1024.0)     # KiB -> bytes
            vals[counter] = {k: sum(v) / len(v) for k, v in rows.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    F, W = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    cal = [(k, v) for k, v in F.items() if "k_halo_pack" in k[0]]
    if not cal:
        raise RuntimeError("calibration launch missing from the counter file")
    n = cal[0][0][1]                                       # one thread per packed double: n*8 bytes of data + n*4 of indices read
    factor = (n * 8 + n * 4) / cal[0][1]
    out = {"fetch_correction": round(factor, 4), "kernels": {}}
    import re
    from mimsem_amd._lib import OPS
    opname = {v: k for k, v in OPS.items()}
    for (k, g), v in sorted(F.items(), key=lambda kv: kv[0][1]):
        m = re.search(r"k_(apply_wave2|apply_wave|elem_apply)<(\d+), *(\d+)", k)
        if m and m.group(1) == "apply_wave2":                      # template <int OP, int LCT, bool ACCUM>: p = 3 only, the operator comes first
            name = "k_apply_wave2<3,%s>" % opname.get(int(m.group(2)), m.group(2))
        elif m:
            name = "k_%s<%s,%s>" % (m.group(1), m.group(2), opname.get(int(m.group(3)), m.group(3)))
        elif "k_wave_perim" in k:
            name = "k_wave_perim"
        elif "k_gather_sum" in k:
            name = "k_gather_sum<2>"
        else:
            continue
        out["kernels"].setdefault(name, []).append({"grid_threads": g, "read_bytes": v * factor, "write_bytes": W.get((k, g), 0.0),
                                                    "total_bytes": v * factor + W.get((k, g), 0.0)})
    return out


def _oracle_patch(pyoracle, pn, ne, nprocs, pi, nk, seed):
    """one reference rank's patch handed to the CPU checker (cpu_baseline leg only): mesh + sphere geometry + stretched levels"""
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import z_levels
    rng = np.random.default_rng(seed)
    cs = CubedSphere(pn, ne, nprocs)
    coords = sphere_coords(pn, ne)
    topo = Topo(cs, pi, nk)
    geom = Geom(topo, cs, coords, nk)
    geom.set_levels(z_levels(nk, geom.n0, rng))
    P = pyoracle.Patch(pn, pn, cs.nel, nk)
    P.set_sphere_geometry(coords[cs.patches[pi].loc0])
    P.set_levels(geom.levs)
    return P, rng


def cpu_worker(args):
    """one host core, three legs of ~budget seconds each share:
    (1) reference-structure assemble+MatMult of Umat on a 12x12-element patch (+ the reference's own matrix-free Uvec::assemble),
    (2) solve_schur_column_eta (eul/VertSolve.cpp:677-823 restated: oracle/o_vertops.c) column after column at nk = 30,
    (3) the operator work of ONE shallow-water Picard iteration (src/SWEqn_Picard.cpp:253-318, 402-621) on the patch:
        every assemble() of the iteration through the CSR insertion path + its MatMults -- Krylov solves NOT included."""
    budget, seed, barrier = args
    from oracle import pyoracle
    P, rng = _oracle_patch(pyoracle, PN, 12, 6, 0, 2, seed)
    x = rng.standard_normal(P.n1)
    # calibrate with EVERY worker loaded (round 3 calibrated while the others were still setting up: the slowest then ran 2.6x its budget)
    if barrier is not None:
        try:
            barrier.wait(timeout=120)
        except Exception:
            pass
    sec, _ = P.bench_assemble_mult("UMAT", x, 4, lev=1, scale=SCALE, flag=1)
    reps = max(2, int(budget / (sec / 4)))
    sec, _ = P.bench_assemble_mult("UMAT", x, reps, lev=1, scale=SCALE, flag=1)
    # the reference's OWN matrix-free variant of the same product (Uvec::assemble, eul/Assembly.cpp:2124-2196): no matrix,
    # no CSR insertion -- the fairer comparison for a matrix-free GPU engine
    t0 = time.perf_counter(); P.uvec(1, SCALE, x); t1 = time.perf_counter() - t0
    mf_reps = max(2, int(0.25 * budget / max(t1, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(mf_reps):
        P.uvec(1, SCALE, x)
    mf_sec = time.perf_counter() - t0
    res = {"units": P.nEl * reps, "sec": sec, "mf_units": P.nEl * mf_reps, "mf_sec": mf_sec}

    # (3) SW Picard iteration: src flavour (scale 1, no thickness: flag 0), Galewsky-style (q upwinded at both time levels):
    # diagnose_F: 2 x M1h->assemble(h) + 4 MatMult; diagnose_Phi: 2 x K->assemble(u) + 3 MatMult + 2 M2 MatMult;
    # 2 x diagnose_q: M0h->assemble(h) + MatMult (+ M0, E01M1 MatMults); 2 x R->assemble(q) + MatMult; M2, 2 M1, 2 M2 MatMults.
    # bench_assemble_mult(op, ., r) = r x (assemble + MatMult); the extra MatMults are charged as one more assemble-free SpMV each by
    # timing Umat / Wmat once with their assembly (they ARE assembled once per step in assemble_operator) -- a slight over-count of
    # four assemblies against ~20 MatMults not counted: the total stays a LOWER bound of the iteration's cost on the CPU.
    h2 = rng.uniform(1.0, 2.0, P.n2) * 1e3; q0 = rng.standard_normal(P.n0) * 1e-8; x2 = rng.standard_normal(P.n2); x0 = rng.standard_normal(P.n0)
    legs = (("UHMAT", x, h2, 2), ("WTQUMAT", x, x, 2), ("PHMAT", x0, h2, 2), ("ROTMAT", x, q0, 2), ("UMAT", x, None, 1), ("WMAT", x2, None, 1))
    def sw_iter(r):
        tot = 0.0
        for op, xin, f, cnt in legs:
            s_, _ = P.bench_assemble_mult(op, xin, cnt * r, lev=0, scale=1.0, flag=0, f1=f)
            tot += s_
        return tot
    t_it = sw_iter(1)
    sw_reps = max(1, int(0.5 * budget / max(t_it, 1e-6)))
    sw_sec = sw_iter(sw_reps)
    res.update({"sw_elements": P.nEl * sw_reps, "sw_sec": sw_sec})

    # (2) column solves: a 2x2-element patch with 30 levels, column after column
    del P
    Pc, rc = _oracle_patch(pyoracle, PN, 2, 6, 0, NK, seed)
    nEl, nk, n2 = Pc.nEl, Pc.nk, Pc.n2e
    area = Pc.det.mean() * 4.0 / n2; dz = Pc.thick.mean()
    lev = lambda nl, lo, hi: rc.uniform(lo, hi, (nEl, nl * n2)) * area * dz
    theta, rho, eta, pi = lev(nk, 280, 320), lev(nk, 0.5, 1.2), lev(nk, 5, 6), lev(nk, 700, 1000)
    F = [rc.standard_normal((nEl, n * n2)) * 1e8 for n in (nk - 1, nk, nk, nk)]
    def col(e):
        Pc.solve_schur_column_eta(e % Pc.nElsX, e // Pc.nElsX, 75.0, theta[e], rho[e], eta[e], pi[e], F[0][e], F[1][e], F[2][e], F[3][e])
    t0 = time.perf_counter(); col(0); tc = time.perf_counter() - t0
    ncol = max(2, int(0.5 * budget / max(tc, 1e-6)))
    t0 = time.perf_counter()
    for i in range(ncol):
        col(i % nEl)
    res.update({"columns": ncol, "col_sec": time.perf_counter() - t0})
    return res


def granted_cores():
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    # the box may grant fewer CPUs than it shows (a cgroup quota: 64 visible cores, 16 granted, on the one-GPU boxes of this pool): more
    # workers than granted CPUs are time-sliced, every one of them then runs a multiple of its budget and "cores" overstates what computed
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cores = max(1, min(cores, int(float(q[0]) / float(q[1]))))
    except Exception:
        try:
            qq = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); pp = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if qq > 0:
                cores = max(1, min(cores, qq // pp))
        except Exception:
            pass
    return cores


def cpu_baseline(budget=6.0, cores=None):
    """-> (cpu_baseline, cpu_baseline_column, cpu_baseline_sw): the oracle ("port") on the box's granted host cores"""
    import multiprocessing as mp
    cores = cores or granted_cores()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        barrier = mgr.Barrier(cores)
        with ctx.Pool(cores) as pool:
            res = pool.map(cpu_worker, [(budget, s, barrier) for s in range(cores)], chunksize=1)
    units = sum(r["units"] for r in res)
    slowest = max(r["sec"] for r in res)
    mf_units = sum(r["mf_units"] for r in res)
    mf_slowest = max(r["mf_sec"] for r in res)
    b1 = {"value": units / slowest, "unit": "element operator-applies/s", "cores": cores, "kind": "port",
          "per_core": units / slowest / cores,
          "sample": f"Umat assemble(CSR)+MatMult, reference cost structure, one 12x12-element p=3 patch per core, "
                    f"{units // cores} element-applies per core in {slowest:.1f}s (gcc -O3)",
          "matrix_free_value": mf_units / mf_slowest,
          "matrix_free_note": "the reference's own matrix-free variant (Uvec::assemble restated, no matrix, no CSR) on the same cores, "
                              f"{mf_units // cores} element-applies per core in {mf_slowest:.1f}s"}
    ncol = sum(r["columns"] for r in res); cslow = max(r["col_sec"] for r in res)
    bc = {"value": ncol / cslow, "unit": "column solves/s", "cores": cores, "kind": "port", "per_core": ncol / cslow / cores,
          "sample": f"oracle solve_schur_column_eta, p=3, 30 levels, one column at a time per core, {ncol // cores} columns per core in {cslow:.1f}s",
          "note": "eul/VertSolve.cpp:677-823 restated (oracle/o_vertops.c): dense products + dense LU where the reference has MatMatMult + PCLU on "
                  "block-sparse MATSEQAIJ matrices"}
    # one Picard iteration of the 24x24x6 sphere = 3 456 elements' worth of the timed per-element work, spread over the cores
    sw_el = sum(r["sw_elements"] for r in res); sw_slow = max(r["sw_sec"] for r in res)
    el_per_s = sw_el / sw_slow                                       # elements of ONE Picard iteration's operator work per second, all cores
    nel3, nel2 = 24 * 24 * 6, 16 * 16 * 6
    bs = {"value": el_per_s / nel3 / 2.0, "unit": "SW time-steps/s (upper bound: operator work only)", "cores": cores, "kind": "port",
          "config3_galewsky_24x24x6_steps_per_s_upper_bound": el_per_s / nel3 / 2.0,
          "config2_w2_16x16x6_picard_iterations_per_s_upper_bound": el_per_s / nel2,
          "picard_iteration_operator_work_ms_24x24x6": 1e3 * nel3 / el_per_s,
          "includes": "per Picard iteration (src/SWEqn_Picard.cpp:253-318, 402-621): 2 Uhmat + 2 WtQUmat + 2 Phmat + 2 RotMat assemblies "
                      "(coefficient loops, triple products, CSR insertion with per-entry column search) each with one MatMult, + Umat and Wmat once",
          "excludes": "the four Krylov solves per iteration (ksp M1, 2 x ksp0h, kspA: PETSc GMRES), ~20 further MatMults, E10/E21 products, "
                      "upwinding of the test/trial functions, halo scatters: the real CPU rate is LOWER than this bound",
          "sample": f"one 12x12-element p=3 patch per core, {sw_el // cores} element-iterations per core in {sw_slow:.1f}s; 2 Picard iterations per step (config 3)"}
    return b1, bc, bs


def replicate(dm, R):
    """R independent copies of a DeviceMesh (disjoint slot ranges): a synthetic larger-than-cache workload"""
    import copy
    out = copy.copy(dm)
    out.inds0 = np.concatenate([dm.inds0 + r * dm.n0 for r in range(R)]).astype(np.int32)
    out.inds1x = np.concatenate([dm.inds1x + r * dm.n1 for r in range(R)]).astype(np.int32)
    out.inds1y = np.concatenate([dm.inds1y + r * dm.n1 for r in range(R)]).astype(np.int32)
    out.nEl = dm.nEl * R
    out.inds2 = np.arange(out.nEl * dm.n * dm.n, dtype=np.int32).reshape(out.nEl, -1)
    out.n0, out.n1, out.n2 = dm.n0 * R, dm.n1 * R, dm.n2 * R
    out.indsq = np.concatenate([dm.indsq + r * dm.nq for r in range(R)]).astype(np.int32); out.nq = dm.nq * R
    out.det = np.tile(dm.det, (R, 1)); out.J = np.tile(dm.J, (R, 1, 1))
    out.thick = np.tile(dm.thick, (1, R, 1)); out.thickInv = np.tile(dm.thickInv, (1, R, 1))
    return out


def column_extras(eng, dm, rng, torch):
    """untimed-for-the-headline extras: solve_schur_column_eta for all 3 456 columns x 30 levels,
    one VertOps assemble+MatMult, and the L2Vecs transposes"""
    nEl, nk, n2 = dm.nEl, NK, eng.n2e
    area = float(dm.det.mean()) * 4.0 / n2
    dz = float(dm.thick.mean())
    lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz)
    theta, rho, eta, pi = lev(nk, 280, 320), lev(nk, 0.5, 1.2), lev(nk, 5, 6), lev(nk, 700, 1000)
    F = [eng.tensor(rng.standard_normal((nEl, n * n2)) * 1e8) for n in (nk - 1, nk, nk, nk)]
    res = {}

    def timeit(fn, reps):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    def timeit_rhs(fn, reps):
        """the solves update their right-hand sides in place: every call gets its own copies, made BEFORE the timed region (rounds 1-4 cloned
        them inside it: four 5 us copy kernels per call that are the harness's, not the solve's)"""
        sets = [[f.clone() for f in F] for _ in range(reps + 1)]
        fn(sets[0]); torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(reps):
            fn(sets[i + 1])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps
    # (the default call and the same call without its pivoted fallback, timed ALTERNATELY -- W O W O, 8 solves each, the mean of each pair: timed
    #  one after the other the first of the two carried the process's first-use effects and a 3 % remedy read as 7-10 %)
    solve = lambda Fc: eng.solve_schur_eta(75.0, theta, rho, eta, pi, *Fc)
    tw, to, nb0 = [], [], 0
    for _ in range(2):
        tw.append(timeit_rhs(solve, 8))
        nb, stf, _ = eng.solve_status()
        eng.set_pivot_fallback(0)
        try:
            to.append(timeit_rhs(solve, 8))
            nb0 = int(eng.solve_status()[0])
        finally:
            eng.set_pivot_fallback(1)
    t, tf = sum(tw) / len(tw), sum(to) / len(to)
    res["schur_column_solves_per_s"] = nEl / t
    res["schur_ms_all_columns"] = t * 1e3
    # the default call re-solves the columns its unpivoted block sweep flags (status 1) by the pivoted band LU of csrc/column_pivot.inc inside
    # the call (the reference's PCLU; on by default since round 5): what is left unresolved, how many were re-solved, and what the remedy
    # costs -- the same call with mimsem_column_set_pivot_fallback(0)
    res["schur_unconverged_columns"] = int(nb)
    res["schur_columns_resolved_by_pivoted_lu"] = int((stf == 3).sum())
    res["schur_columns_accepted_on_backward_error"] = int((stf == 4).sum())
    res["schur_pivot_fallback"] = {"ms_all_columns_with_it": t * 1e3, "ms_all_columns_without_it": tf * 1e3, "cost_frac": t / tf - 1.0,
                                   "columns_flagged_by_the_block_sweep": nb0, "columns_resolved_by_pivoted_lu": int((stf == 3).sum()), "columns_accepted_on_backward_error": int((stf == 4).sum()),
                                   "unconverged_columns": int(nb)}
    # what bounds it: FP64 work counted by the SQ counters (profiles/r03_column_pmc.txt: FMA = 2 flop, MUL / ADD = 1, x 64 lanes per wave
    # instruction, all three kernels of the solve) against the 78.6 TFLOP/s vector FP64 peak, and the bytes the solve must move (the four
    # fields and four right-hand sides in, the four updated right-hand sides out, det + thickness per quadrature point) against 8 TB/s
    mp12 = (PN + 1) ** 2
    fl = SCHUR_ETA_FLOP_PER_COLUMN_LEVEL * nEl * nk
    by = nEl * 8 * (4 * nk * n2 + 2 * (4 * nk - 1) * n2 + mp12 * (1 + 2 * nk))
    res["schur_roofline"] = {"bound": "fp64 valu (latency chain of the block-Thomas sweep)", "flop_per_solve_all_columns": fl, "TFLOPs": fl / t / 1e12,
                             "flop_frac": fl / t / 1e12 / FP64_PEAK_TFLOPS, "compulsory_bytes": by, "GBs": by / t / 1e9, "hbm_frac": by / t / 1e9 / HBM_PEAK_GBS}
    # solve_schur_column_3: theta and velz live on interfaces -- cell integrals WITHOUT the thickness (the scaling of tests/test_gpu_column.py's
    # _col_fields).  Rounds 1-3 scaled theta by dz as well (~1e3 too large): a pathologically conditioned L_rt_rt that no physical column has;
    # harmless while the solve did a fixed amount of work, wrong for the adaptive refinement of round 4's pentadiagonal solve
    thetaI, rt = lev(nk + 1, 280, 320) / dz, lev(nk, 250, 400)
    velz = lev(nk - 1, -1.0, 1.0) / dz
    t = timeit_rhs(lambda Fc: eng.solve_schur_3(75.0, thetaI, velz, rho, rt, pi, *Fc), 3)
    res["schur3_column_solves_per_s"] = nEl / t
    res["schur3_ms_all_columns"] = t * 1e3
    res["schur3_unconverged_columns"] = int(eng.solve_status()[0])
    by3 = nEl * 8 * ((nk + 1) * n2 + (nk - 1) * n2 + 3 * nk * n2 + 2 * (4 * nk - 1) * n2 + mp12 * (1 + 2 * nk))
    fl3 = SCHUR_3_FLOP_PER_COLUMN_LEVEL * nEl * nk
    res["schur3_roofline"] = {"bound": "fp64 valu", "flop_per_solve_all_columns": fl3, "TFLOPs": fl3 / t / 1e12, "flop_frac": fl3 / t / 1e12 / FP64_PEAK_TFLOPS,
                              "compulsory_bytes": by3, "GBs": by3 / t / 1e9, "hbm_frac": by3 / t / 1e9 / HBM_PEAK_GBS}
    # the caller of the column solve: one Newton iteration of VertSolve::solve_schur_eta (residual assembly, EOS residual, entropy
    # diagnostics, the Schur solve, the updates and both theta diagnoses) for every column, from an EOS-consistent state at rest
    from mimsem_amd.geom import gll_points
    from mimsem_amd.vertsolve import VertSolve
    wd = np.diff(gll_points(PN)); wj = np.outer(wd, wd).ravel()
    cell = dm.det.mean(axis=1)[:, None, None] * dm.thick.mean(axis=2).T[:, :, None] * wj[None, None, :]       # [nEl, nk, n2]
    zl = np.mean([g.levs.mean(axis=1) for g in dm.geoms], axis=0)               # mean interface heights -> level mid-heights
    zm = 0.5 * (zl[:-1] + zl[1:])
    th_v = 300.0 + 0.004 * zm                                                    # hydrostatic column: dPi/dz = -g/theta, EOS for rho theta
    pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
    rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
    colv = lambda v: eng.tensor((cell * v[None, :, None]).reshape(nEl, nk * n2) * (1.0 + 1e-4 * rng.standard_normal((nEl, nk * n2))))
    vs = VertSolve(eng, 75.0)
    levs = np.zeros((nk + 1, dm.nq))
    for g in dm.geoms:
        levs[:, np.searchsorted(dm.gidq, g.loc0[np.arange(g.n0)])] = g.levs
    zv = vs.init_gz(levs)
    st = (eng.zeros(nEl, (nk - 1) * n2), colv(rho_v), colv(rho_v * th_v), colv(pi_v))
    vs.solve_schur_eta(*st, zv, maxit=2, tol=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vs.solve_schur_eta(*st, zv, maxit=4, tol=0.0)
    torch.cuda.synchronize(); tn = (time.perf_counter() - t0) / 4
    res["vertical_newton_iteration_ms"] = tn * 1e3
    res["vertical_newton_norms_last"] = vs.history[-1]
    # the same iterations with the HOST in C++ (mimsem_amd/host/vert_call.cpp over mimsem_vertsolve.hpp), from the same start state
    exe = os.path.join(ROOT, "mimsem_amd", "host", "vert_call")
    if os.path.exists(exe):
        import subprocess
        import tempfile
        from mimsem_amd.workloads import mesh_arrays, write_arrays
        with tempfile.TemporaryDirectory() as tmp:
            case = os.path.join(tmp, "case.arr")
            arr = mesh_arrays(dm)
            cpu = lambda tt: tt.cpu().numpy()
            arr.update(dt=np.array([75.0]), zv=cpu(zv), velz=cpu(st[0]), rho=cpu(st[1]), rt=cpu(st[2]), exner=cpu(st[3]))
            write_arrays(case, arr)
            try:
                r_ = subprocess.run([exe, case, "4"], capture_output=True, text=True, timeout=300)
                res["vertical_newton_cpp_host"] = json.loads(r_.stdout) if r_.returncode == 0 else {"error": (r_.stderr or r_.stdout)[-300:]}
            except Exception as e_:                    # noqa: BLE001
                res["vertical_newton_cpp_host"] = {"error": repr(e_)[:300]}
    # the linear solve of that Newton iteration alone, on the state it was given (a hydrostatic column with 1e-4 noise: what the UMJS14 run
    # hands to solve_schur_column_eta) -- beside schur_ms_all_columns above, whose uniformly random fields make a handful of columns need
    # every refinement step (the tail of the Thomas launch)
    vs.keep_solve_args = True
    vs.solve_schur_eta(*st, zv, maxit=1, tol=0.0)
    vs.keep_solve_args = False
    if getattr(vs, "last_solve_args", None) is not None:
        dtm, thm, rhm, etm, pim, Fm = vs.last_solve_args
        sets = [[f.clone() for f in Fm] for _ in range(9)]
        eng.solve_schur_eta(dtm, thm, rhm, etm, pim, *sets[0]); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8):
            eng.solve_schur_eta(dtm, thm, rhm, etm, pim, *sets[i + 1])
        torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / 8
        nbm, stm, _ = eng.solve_status()
        res["schur_model_state"] = {"ms_all_columns": tm * 1e3, "column_solves_per_s": nEl / tm, "unconverged_columns": int(nbm),
                                    "columns_by_status": {str(k): int((stm == k).sum()) for k in (0, 1, 3, 4)},
                                    "state": "hydrostatic column (theta = 300 K + 4 K/km), 1e-4 relative noise, one Newton iteration in"}
    t = timeit(lambda: eng.colop_apply("CONST_RHO", theta, f1=rho, nout_slots=nk), 20)
    res["vertops_assemble_apply_columns_per_s"] = nEl / t
    vh = eng.tensor(rng.standard_normal((nk, dm.n2)))
    t = timeit(lambda: eng.l2_horiz_to_vert(vh), 50)
    res["l2_transposes_per_s"] = 1.0 / t
    res["l2_transpose_GBs"] = 2 * vh.numel() * 8 / t / 1e9
    return res


def box_column_workload(local_rank, rng):
    """BASELINE config 5's column half: p = 4, 32 x 32 periodic box x 64 levels (1 024 columns, 16 x 16 blocks), hydrostatic state"""
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import BoxGeom, gll_points
    from mimsem_amd.mesh import PeriodicBox, box_coords
    from mimsem_amd.topo import Topo
    pn, nkb = 4, 64
    bx = PeriodicBox(pn, 32, 4); bc = box_coords(pn, 32, 1000.0)
    bt = [Topo(bx, p, nkb) for p in range(4)]; bg = [BoxGeom(t, bx, bc, nkb, 1000.0) for t in bt]
    dz = 1500.0 / nkb
    levs = np.repeat(np.linspace(0.0, 1500.0, nkb + 1)[:, None], bg[0].n0, axis=1)
    levs[1:-1] += 0.05 * dz * rng.uniform(-1, 1, (nkb - 1, bg[0].n0))
    for g in bg:
        g.set_levels(levs)
    dmb = DeviceMesh(bt, bg, nk=nkb, numbering="global"); engb = Engine(dmb, device=local_rank)
    nEl, n2 = dmb.nEl, engb.n2e
    wd = np.diff(gll_points(pn)); wj = np.outer(wd, wd).ravel()
    detm = dmb.det.mean(axis=1); thm = dmb.thick.mean(axis=2).T
    zi = levs.mean(axis=1); zm = 0.5 * (zi[1:] + zi[:-1])
    th_v = 300.0 + 0.004 * zm; thI_v = 300.0 + 0.004 * zi
    pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
    rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
    pert = lambda nl: 1.0 + 1e-2 * rng.standard_normal((nEl, nl * n2))
    lev = lambda v: engb.tensor((detm[:, None, None] * thm[:, :, None] * v[None, :, None] * wj[None, None, :]).reshape(nEl, nkb * n2) * pert(nkb))
    itf = lambda v, nl: (detm[:, None, None] * v[None, :nl, None] * wj[None, None, :]).reshape(nEl, nl * n2) * pert(nl)
    fld = dict(rho=lev(rho_v), rt=lev(rho_v * th_v), pi=lev(pi_v), thetaL=lev(th_v), eta=lev(np.log(th_v)),
               theta=engb.tensor(itf(thI_v, nkb + 1)),
               velz=engb.tensor(itf(np.ones(nkb + 1), nkb - 1) * 0.5 * rng.standard_normal((nEl, (nkb - 1) * n2))))
    F = [engb.tensor(rng.standard_normal((nEl, n * n2)) * 1e8) for n in (nkb - 1, nkb, nkb, nkb)]
    return engb, dmb, levs, fld, F, (rho_v, th_v, pi_v, detm, thm, wj)


def column_box_p4_extras(local_rank, rng, torch):
    """config 5's column half in the driver line: solve_schur_column_eta, the box twin of solve_schur_column_3
    (box/VertSolve.cpp:879-1058) and one Newton iteration of VertSolve::solve_schur_eta on 1 024 columns x 64 levels at p = 4"""
    from mimsem_amd.vertsolve import VertSolve
    engb, dmb, levs, fld, F, (rho_v, th_v, pi_v, detm, thm, wj) = box_column_workload(local_rank, rng)
    nEl, nkb, n2 = dmb.nEl, 64, engb.n2e
    mp12 = 25

    def timeit(fn, reps):                                     # (own copies of the in-place right-hand sides per call, made before the timed region)
        sets = [[f.clone() for f in F] for _ in range(reps + 1)]
        fn(sets[0]); torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(reps):
            fn(sets[i + 1])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps
    res = {"workload": "p=4, 32x32 periodic box x 64 levels: 1 024 columns, 16x16 blocks (BASELINE config 5, column half)"}
    dt = 0.5
    t = timeit(lambda Fc: engb.solve_schur_eta(dt, fld["thetaL"], fld["rho"], fld["eta"], fld["pi"], *Fc), 5)
    nun, _, _ = engb.solve_status()
    # algorithmic figures of SURVEY 8(d) for C5 at p = 4: (768 + 32) x 8 x nk bytes per column; block-Thomas 2/3 n^3 + 2 n^3 + 2 n^3 flop per level
    by = nEl * (768 + 32) * 8 * nkb
    fl = nEl * nkb * (14.0 / 3.0 * 16 ** 3 + 2 * 16 * 16 * 2)
    res["schur_eta"] = {"ms_all_columns": t * 1e3, "column_solves_per_s": nEl / t, "unconverged_columns": nun,
                        "algorithmic_bytes": by, "GBs": by / t / 1e9, "hbm_frac": by / t / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_flop": fl, "TFLOPs": fl / t / 1e12, "flop_frac": fl / t / 1e12 / FP64_PEAK_TFLOPS}
    t = timeit(lambda Fc: engb.solve_schur_3(dt, fld["theta"], fld["velz"], fld["rho"], fld["rt"], fld["pi"], *Fc, flags=3), 3)
    by3 = nEl * (5 * 256 + 32) * 8 * nkb
    res["schur_3_box"] = {"ms_all_columns": t * 1e3, "column_solves_per_s": nEl / t,
                          "algorithmic_bytes": by3, "GBs": by3 / t / 1e9, "hbm_frac": by3 / t / 1e9 / HBM_PEAK_GBS}
    vs = VertSolve(engb, dt)
    lq = np.zeros((nkb + 1, dmb.nq))
    for g in dmb.geoms:
        lq[:, np.searchsorted(dmb.gidq, g.loc0[np.arange(g.n0)])] = g.levs
    zv = vs.init_gz(lq)
    st = (engb.zeros(nEl, (nkb - 1) * n2), fld["rho"], fld["rt"], fld["pi"])
    vs.solve_schur_eta(*st, zv, maxit=2, tol=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vs.solve_schur_eta(*st, zv, maxit=4, tol=0.0)
    torch.cuda.synchronize(); tn = (time.perf_counter() - t0) / 4
    res["vertical_newton_iteration_ms"] = tn * 1e3
    res["vertical_newton_norms_last"] = vs.history[-1]
    del engb
    return res


def sweep_extras(local_rank, torch):
    """SURVEY 8(d) throughput sweep: applies/s of the operator families B1, B3, B4, B8, B9, B11 at the batch sizes of the BASELINE
    configurations (384 / 1 536 / 3 456 / 103 680 element-level pairs) and ~1e6 pairs (the config 4 sphere with 290 levels),
    with the HBM GB/s the algorithmic bytes of SURVEY 8(d) imply.  Wall clock over back-to-back launches, inputs resident."""
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import z_levels
    fam_bytes = {"UMAT": 1440, "WMAT": 400, "UHMAT": 1512, "WTQUMAT": 1320, "ROTMAT": 1632, "WHMAT": 472}     # B1 B3 B4 B8 B9 B11
    rng = np.random.default_rng(20241024)
    res = {}
    for ne, nk in ((8, 1), (16, 1), (24, 1), (24, 30), (24, 290)):
        cs = CubedSphere(PN, ne, 6); coords = sphere_coords(PN, ne)
        topos = [Topo(cs, p, nk) for p in range(6)]
        geoms = [Geom(t, cs, coords, nk) for t in topos]
        for g in geoms:
            g.set_levels(z_levels(nk, g.n0))
        dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
        eng = Engine(dm, device=local_rank)
        units = dm.nEl * nk
        x1 = eng.tensor(rng.standard_normal((nk, dm.n1))); x2 = eng.tensor(rng.standard_normal((nk, dm.n2)))
        h = eng.tensor(rng.uniform(1, 2, (nk, dm.n2)) * 1e3); q0 = eng.tensor(rng.standard_normal((nk, dm.n0)) * 1e-4)
        y1, y2 = eng.zeros(nk, dm.n1), eng.zeros(nk, dm.n2)
        row = {}
        for op, xin, f, fl, out in (("UMAT", x1, None, 1, y1), ("WMAT", x2, None, 1, y2), ("UHMAT", x1, h, 1, y1),
                                    ("WTQUMAT", x1, x1, 0, y2), ("ROTMAT", x1, q0, 0, y1), ("WHMAT", x2, h, 1, y2)):
            reps = 200 if units < 500000 else 30
            for _ in range(5):
                eng.apply(op, xin, f=f, lev0=0, scale=SCALE, flags=fl, out=out)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(reps):
                eng.apply(op, xin, f=f, lev0=0, scale=SCALE, flags=fl, out=out)
            torch.cuda.synchronize(); el = (time.perf_counter() - t1) / reps
            row[op] = {"applies_per_s": units / el, "us_per_apply_call": el * 1e6, "GBs_algorithmic": units * fam_bytes[op] / el / 1e9}
        res["%dx%dx6_x%d_levels_%d_units" % (ne, ne, nk, units)] = row
        del eng, x1, x2, h, q0, y1, y2
        torch.cuda.empty_cache()
    return res


def sw_extras(local_rank, torch):
    """SW time steps/s (the second half of BASELINE's metric): SWEqn::solve as the reference drivers call it, on the
    config 2 (16x16x6, Williamson-2 steady state: q from the mean state, iterate to 1e-14) and config 3 (24x24x6, the Galewsky jet
    with its perturbation: 2 Picard iterations, upwinded potential vorticity) grids, everything resident on the device."""
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.sweqn import SWEqn, galewsky, williamson2
    from mimsem_amd.topo import Topo
    res = {}
    for name, ne, dt, nits, q_exact, nsteps in (("config2_w2_16x16x6", 16, 600.0, 99, True, 10), ("config3_galewsky_24x24x6", 24, 360.0, 2, False, 240)):
        cs = CubedSphere(PN, ne, 6); coords = sphere_coords(PN, ne)
        topos = [Topo(cs, p, 1) for p in range(6)]
        geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
        for g in geoms:
            g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
        dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
        eng = Engine(dm, device=local_rank)
        xq = np.zeros((dm.nq, 3))
        for g in geoms:
            xq[g.loc0] = coords[g.loc0]
        S = SWEqn(eng, xq[dm.gidq])
        is_w2 = "w2" in name                     # config 3: the Galewsky jet + perturbation (src/Galewsky.cpp), 2 Picard iterations, upwinded q
        uq, hq = (williamson2(torch.as_tensor(xq[dm.gidq], device=eng.device), alpha=0.0) if is_w2
                  else galewsky(torch.as_tensor(xq[dm.gidq], device=eng.device)))
        u, h = S.init1(uq), S.init2(hq)
        for _ in range(3):                                               # warm-up steps: graph captures, adaptive sweep counts settle
            u, h = S.solve(u, h, dt, nits=nits, q_exact=q_exact)
        c0 = S.conservation(u, h)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        picard = 0
        for _ in range(nsteps):
            u, h = S.solve(u, h, dt, nits=nits, q_exact=q_exact)
            picard += len(S.history)
        torch.cuda.synchronize(); el = time.perf_counter() - t1
        c1 = S.conservation(u, h)
        # the reference's own verification metric (Williamson2.cpp:138-151): [L1, L2, Linf] against the steady analytic state
        errs = None
        if is_w2:
            wq = 2.0 * 38.61068276698372 / 6371220.0 * torch.sin(S.lat)
            errs = {"vorticity": S.err0(S.curl(u), wq), "velocity": S.err1(u, uq), "depth": S.err2(h, hq)}
        res[name] = {"steps_per_s": nsteps / el, "ms_per_step": 1e3 * el / nsteps, "steps_timed": nsteps, "dt": dt, "picard_iterations_per_step": picard / nsteps,
                     "fixed_length_iterations": S.fixed_iterations, "adaptive_iterations": S.adaptive_iterations, "recalibrations": S.recalibrations,
                     "relative_drift_over_timed_steps": {k: (c1[k] - c0[k]) / abs(c0[k]) for k in ("mass", "energy", "enstrophy")},
                     "williamson2_error_norms_L1_L2_Linf": errs, "days": (nsteps + 3) * dt / 86400.0,
                     "krylov_iterations_last": dict(S.its), "elements": dm.nEl, "dofs": dm.n1 + dm.n2}
        # the same steps with the HOST in C++ (mimsem_amd/host/sw_call.cpp over mimsem_sweqn.hpp: no Python, no torch -- what a C++ driver like
        # src/Galewsky.cpp gets), from the state reached here
        res[name]["cpp_host"] = sw_cpp_host(dm, S.fg[0].cpu().numpy(), u[0].cpu().numpy(), h[0].cpu().numpy(), dt, nsteps, nits, q_exact)
        del S, eng
    return res


def sw_cpp_host(dm, fg, u, h, dt, nsteps, nits, q_exact):
    import subprocess
    import tempfile
    from mimsem_amd.workloads import write_sw_case
    exe = os.path.join(ROOT, "mimsem_amd", "host", "sw_call")
    if not os.path.exists(exe):
        return {"error": "mimsem_amd/host/sw_call not built (__graft_entry__.build())"}
    with tempfile.TemporaryDirectory() as tmp:
        case = os.path.join(tmp, "case.bin")
        write_sw_case(case, dm, fg, u, h, dt, max(nsteps, 20), min(nits, 50), q_exact)
        try:
            out = subprocess.run([exe, case, "3"], capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                return {"error": (out.stderr or out.stdout)[-300:]}
            return json.loads(out.stdout)
        except Exception as e:                     # noqa: BLE001 -- an extra must not take the bench line down
            return {"error": repr(e)[:300]}


FAMILIES = (("B1", "UMAT", 1, None, 1), ("B3", "WMAT", 2, None, 1), ("B4", "UHMAT", 1, 2, 1), ("B8", "WTQUMAT", 1, 1, 0),
            ("B9", "ROTMAT", 1, 0, 0), ("B11", "WHMAT", 2, 2, 1))          # (op, input form, coefficient form, flags)


def family_bytes(op, nEl, sizes, nlev, pn=PN):
    """COMPULSORY bytes of one launch of an operator family over all (element, level) units: every input once, every output once
    (DESIGN 4.4).  x and y per level in their spaces, the coefficient field per level, thickInv per unit, and the metric once per
    element: {gaa, gab, gbb, 1/det} = 32 B per quadrature point for the operators that contract with J^T J (Umat, Uhmat, WtQUmat),
    8 B per point (one factor: Q/det, or the rotational factor) for Wmat, Whmat, RotMat."""
    mp12 = (pn + 1)**2
    n0, n1, n2 = sizes
    sp = {"UMAT": (n1, None, n1, 32), "WMAT": (n2, None, n2, 8), "UHMAT": (n1, n2, n1, 32), "WTQUMAT": (n1, n1, n2, 32),
          "ROTMAT": (n1, n0, n1, 8), "WHMAT": (n2, n2, n2, 8)}[op]
    nin, ncf, nout, metric = sp
    return nlev*8*(nin + (ncf or 0) + nout) + nEl*nlev*mp12*8 + nEl*mp12*metric


def families_extras(eng, dm, rng, torch):
    """every operator family of the hot path on the headline grid (3 456 elements x 30 levels), kernel time from the
    context's HIP events (both launches where an operator has two), compulsory bytes -> fraction of the 8 TB/s roofline"""
    x1 = eng.tensor(rng.standard_normal((NK, dm.n1))); x2 = eng.tensor(rng.standard_normal((NK, dm.n2)))
    h = eng.tensor(rng.uniform(1, 2, (NK, dm.n2))*1e3); q0 = eng.tensor(rng.standard_normal((NK, dm.n0))*1e-4)
    y1, y2 = eng.zeros(NK, dm.n1), eng.zeros(NK, dm.n2)
    units = dm.nEl*NK
    res = {}
    for row, op, fin, fcf, fl in FAMILIES:
        xin = x1 if fin == 1 else x2
        f = {None: None, 0: q0, 1: x1, 2: h}[fcf]
        out = y1 if op in ("UMAT", "UHMAT", "ROTMAT") else y2
        call, _ = eng.prepare_apply(op, xin, f=f, lev0=0, scale=SCALE, flags=fl, out=out)
        for _ in range(5):
            call()
        torch.cuda.synchronize(); eng.set_profiling(1); t1 = time.perf_counter()
        for _ in range(40):
            call()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t1)/40
        c1, c2, cn = eng.profile_read(); eng.set_profiling(0)
        kus = (c1 + c2)/max(cn, 1)*1e3 if cn else wall*1e6
        b = family_bytes(op, dm.nEl, (dm.n0, dm.n1, dm.n2), NK)
        res[row] = {"op": op, "applies_per_s": units/wall, "kernel_us": kus, "kernels": 2 if c2 > 0 else 1, "bytes_per_launch": b,
                    "achieved_GBs": b/(kus*1e-6)/1e9, "frac": b/(kus*1e-6)/1e9/HBM_PEAK_GBS}
    return {"workload": "3 456 elements x 30 levels per launch (cache resident), compulsory bytes / HIP-event kernel time against 8 TB/s", "rows": res}


def local_layout_extras(local_rank, rng, torch):
    """What an UNCHANGED reference rank gets: ONE 12 x 12-element patch of the 24-patch config-4 sphere in the
    reference's own rank-local numbering (eul/Topo.cpp:82-86, 214-240: no slot-pair plan -> the two-pass kernels) and in the
    co-located local numbering Topo(paired=True) (the two-line Topo.cpp change of INTEGRATION 1.1 -> the wave-level kernels), as
    30-level calls and as the reference's own single-level mult() calls"""
    import ctypes as C
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import z_levels
    cs = CubedSphere(PN, NE, NPATCH); coords = sphere_coords(PN, NE)
    res = {}
    for paired in (False, True):
        t = Topo(cs, 5, NK, paired=paired)
        g = Geom(t, cs, coords, NK); g.set_levels(z_levels(NK, g.n0))
        dmp = DeviceMesh([t], [g], nk=NK, numbering="local")
        e = Engine(dmp, device=local_rank)
        st = (C.c_int*5)()
        wave = e.L.mimsem_op_wave_stats(e.ctx, NK, st) == 1
        x = e.tensor(rng.standard_normal((NK, dmp.n1))); y = e.zeros(NK, dmp.n1)
        row = {"form": "wave-level (k_apply_wave + k_wave_perim)" if wave else "two-pass (k_elem_apply + k_gather_sum)", "elements": dmp.nEl}
        for label, nl, reps in (("30_levels_per_call", NK, 200), ("1_level_per_call", 1, 400)):
            call, _ = e.prepare_apply("UMAT", x[:nl], lev0=0, scale=SCALE, flags=1, out=y[:nl])
            for _ in range(10):
                call()
            torch.cuda.synchronize(); e.set_profiling(1); t1 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize(); wall = (time.perf_counter() - t1)/reps
            c1, c2, cn = e.profile_read(); e.set_profiling(0)
            row[label] = {"applies_per_s": dmp.nEl*nl/wall, "us_per_call_wall": wall*1e6, "kernel_us": (c1 + c2)/cn*1e3}
        res["paired_local" if paired else "reference_local"] = row
        del e
    # the same patch from a C++ host over the shim (mimsem_amd/host/bench_call.cpp, built by __graft_entry__.build()): the reference's own
    # per-level loop as written, the same loop recorded in a hipGraph (mimsem_graph_*) and one 30-level call -- a child process
    exe = os.path.join(ROOT, "mimsem_amd", "host", "bench_call")
    if os.path.exists(exe):
        import subprocess
        try:
            r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            res["cpp_host"] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as ex:          # noqa: BLE001
            res["cpp_host"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return {"workload": "Umat apply on ONE 12x12-element patch (144 elements) of the config-4 sphere in a rank-LOCAL vector layout", "rows": res}


def roofline_entry(bm, k1, k12, cache_resident, note, kname="k_elem_apply<3,UMAT>", k2name="k_gather_sum<2>"):
    """roofline object of the dominant kernel from in-run HIP-event durations (seconds) and the launch's compulsory bytes; the
    whole operator (both kernels) and SURVEY 8(d)'s per-unit figures ride along as secondary entries"""
    a1 = bm["k1_compulsory"] / k1 / 1e9
    a12 = bm["op_compulsory"] / k12 / 1e9
    return {"bound": "hbm", "kernel": kname, "achieved": a1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": a1 / HBM_PEAK_GBS, "traffic": None,
            "traffic_note": "filled by the child rocprofv3 --pmc passes at the end of the run (absent with --no-pmc / N > 1: null)",
            "cache_resident": cache_resident, "note": note,
            "avg_kernel_us": k1 * 1e6, "units_per_launch": bm["units"],
            "bytes_per_launch": bm["k1_compulsory"], "bytes_per_unit": bm["k1_compulsory"] / bm["units"],
            "byte_model": ("compulsory: x once per level (n1 doubles), thickInv per unit, the metric record and lane tables once per "
                           "element / wave-group, the y slots the kernel completes; its partial sums for the perimeter pass are overhead "
                           "(under `requested` only)") if "wave" in kname else
                          "compulsory: x once per level (n1 doubles), thickInv + element-local result per unit, J/det/slots once per element",
            "requested": {"bytes_per_launch": bm["k1_requested"], "GBs": bm["k1_requested"] / k1 / 1e9,
                          "note": "what the kernel's loads ask for (x gathered per element, metric re-read per level chunk): the part above "
                                  "`achieved` is served by L2 / Infinity Cache"},
            "algorithmic_reference": {"bytes_per_unit": BYTES_K1_B1, "GBs": bm["units"] * BYTES_K1_B1 / k1 / 1e9,
                                      "note": "SURVEY 8(d) figure (metric charged to every unit): exceeds what HBM delivers because one launch "
                                              "shares the metric between the levels of an element; NOT a roofline fraction"},
            "whole_operator": {"kernels": kname + " + " + k2name, "avg_us": k12 * 1e6,
                               "bytes_per_launch": bm["op_compulsory"], "achieved": a12, "frac": a12 / HBM_PEAK_GBS,
                               "byte_model": "every input once (x, thickInv, metric), every output once (y)" if "wave" in kname else
                                             "both passes' compulsory bytes (element-local results written and read)",
                               "algorithmic_reference_bytes_per_unit": BYTES_OP_B1}}


def copy_reference_us(nbytes, torch, device, reps=30):
    """HIP-event time of a device-to-device copy that moves `nbytes` in total (half read, half written): what a kernel with no work,
    no gather and no dependent address chain needs for the same bytes at this size -- context for `frac`, never a roofline"""
    n = max(int(nbytes) // 16, 1)
    a = torch.empty(n, dtype=torch.float64, device=device).normal_(); b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(reps):
        e0.record(); b.copy_(a); e1.record(); e1.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3)
    best.sort()
    return best[len(best) // 2]


def cold_workload(dm, R, local_rank, rng, torch, steps=20):
    """the same step on R independent copies of the sphere: working set >> the 256 MiB Infinity Cache, i.e. HBM-resident"""
    from mimsem_amd.device import Engine
    dmc = replicate(dm, R)
    engc = Engine(dmc, device=local_rank)
    xc = engc.tensor(rng.standard_normal((NK, dmc.n1))); yc = engc.zeros(NK, dmc.n1)
    call, _ = engc.prepare_apply("UMAT", xc, lev0=0, scale=SCALE, flags=1, out=yc)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    engc.set_profiling(1); t1 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize(); dtc = time.perf_counter() - t1
    c1, c2, cn = engc.profile_read(); engc.set_profiling(0)
    bm, kname, k2name, lch = launch_bytes(engc, dmc, NK)
    ws = engc.L.mimsem_ctx_workspace_bytes(engc.ctx) / 1e6 + 2 * xc.numel() * 8 / 1e6
    r = roofline_entry(bm, c1 / cn * 1e-3, (c1 + c2) / cn * 1e-3, cache_resident=False,
                       note="%d independent spheres in one launch, context + vectors %.0f MB >> Infinity Cache" % (R, ws), kname=kname, k2name=k2name)
    r["whole_operator"]["copy_of_the_same_bytes_us"] = copy_reference_us(bm["op_compulsory"], torch, engc.device)
    r.update({"replicas": R, "level_chunk": lch, "working_set_MB": ws, "value": bm["units"] * steps / dtc,
              "value_unit": "element operator-applies/s (wall clock over %d back-to-back steps)" % steps})
    # every family of SURVEY 8(d) on the same HBM-resident workload: kernel time from the context's HIP events, compulsory
    # bytes (family_bytes); the PMC passes at the end of the run add traffic / traffic_over_compulsory per row
    x2 = engc.tensor(rng.standard_normal((NK, dmc.n2))); hh = engc.tensor(rng.uniform(1, 2, (NK, dmc.n2))*1e3)
    q0 = engc.tensor(rng.standard_normal((NK, dmc.n0))*1e-4); y2 = engc.zeros(NK, dmc.n2)
    rows = {}
    for row, op, fin, fcf, fl in FAMILIES:
        xin = xc if fin == 1 else x2
        f = {None: None, 0: q0, 1: xc, 2: hh}[fcf]
        o = yc if op in ("UMAT", "UHMAT", "ROTMAT") else y2
        callf, _ = engc.prepare_apply(op, xin, f=f, lev0=0, scale=SCALE, flags=fl, out=o)
        for _ in range(3):
            callf()
        torch.cuda.synchronize(); engc.set_profiling(1)
        for _ in range(10):
            callf()
        torch.cuda.synchronize()
        f1, f2, fn = engc.profile_read(); engc.set_profiling(0)
        kus = (f1 + f2)/max(fn, 1)*1e3
        b = family_bytes(op, dmc.nEl, (dmc.n0, dmc.n1, dmc.n2), NK)
        rows[row] = {"op": op, "kernel_us": kus, "kernels": 2 if f2 > 0 else 1, "bytes_per_launch": b, "achieved_GBs": b/(kus*1e-6)/1e9,
                     "frac": b/(kus*1e-6)/1e9/HBM_PEAK_GBS, "applies_per_s": dmc.nEl*NK/(kus*1e-6), "traffic": None, "traffic_over_compulsory": None}
    del engc
    return r, {"workload": "%d spheres per launch (%d units, HBM resident): compulsory bytes / HIP-event kernel time against 8 TB/s; traffic = "
                           "FETCH_SIZE x2 + WRITE_SIZE of the family's kernels from this run's PMC passes" % (R, dmc.nEl*NK), "rows": rows}


def _g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _r(v, nd=4):
    """numbers rounded to `nd` significant digits (the compact line is a record, not an archive)"""
    if isinstance(v, float):
        return float("%.*g" % (nd, v)) if v == v and abs(v) != float("inf") else None
    return v


def compact_record(out, extras_file=None):
    """The ONE line the driver parses (round 4: a 20.6 KB line came back as parsed = null): the contract's keys, `roofline`,
    `roofline_cold`, `cpu_baseline` (+ _column, _sw) and one-number summaries, < 4 KB; the full object goes to `extras_file`."""
    rec = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    rec["value"] = _r(rec["value"], 6); rec["ms_per_step"] = _r(rec["ms_per_step"], 6)
    rec["ms_per_step_median"] = _r(out.get("ms_per_step_median"), 6)
    cfg = out.get("config", {})
    rec["config"] = {"workload": "Umat (B1) matrix-free apply, p=3 24x24x6 cubed sphere x 30 levels (BASELINE config 4 grid)",
                     **{k: cfg.get(k) for k in ("order", "elements", "levels", "units_per_step", "patches_per_gpu", "form", "halo_transport", "same_step_over_torch_all_to_all_ms") if cfg.get(k) is not None}}

    def roof(r):
        if not isinstance(r, dict) or "frac" not in r:
            return r if r is None else {"error": str(r.get("error"))[:160]}
        w = r.get("whole_operator", {})
        return {"bound": "hbm", "kernel": r.get("kernel"), "achieved": _r(r.get("achieved")), "peak": r.get("peak"), "unit": "GB/s",
                "frac": _r(r.get("frac")), "avg_kernel_us": _r(r.get("avg_kernel_us")), "bytes_per_launch": r.get("bytes_per_launch"),
                "units_per_launch": r.get("units_per_launch"), "traffic": _r(r.get("traffic"), 6), "cache_resident": r.get("cache_resident"),
                "whole_operator": {"kernels": w.get("kernels"), "avg_us": _r(w.get("avg_us")), "bytes_per_launch": w.get("bytes_per_launch"),
                                   "frac": _r(w.get("frac")), "traffic_over_compulsory": _r(w.get("traffic_over_compulsory")),
                                   "copy_of_the_same_bytes_us": _r(w.get("copy_of_the_same_bytes_us"))}}
    if "roofline" in out:
        rec["roofline"] = roof(out["roofline"])
    if "roofline_cold" in out:
        rec["roofline_cold"] = roof(out["roofline_cold"])
        if isinstance(rec["roofline_cold"], dict) and "frac" in rec["roofline_cold"]:
            rec["roofline_cold"]["units_per_launch"] = _g(out, "roofline_cold", "units_per_launch")
            rec["roofline_cold"]["value"] = _r(_g(out, "roofline_cold", "value"))
    for key in ("cpu_baseline", "cpu_baseline_column", "cpu_baseline_sw"):
        c = out.get(key)
        if isinstance(c, dict):
            rec[key] = ({k: (_r(c[k]) if not isinstance(c[k], str) else c[k][:150]) for k in ("value", "unit", "cores", "kind", "matrix_free_value", "sample") if k in c}
                        if "value" in c else {"error": str(c.get("error"))[:160]})
    fam = _g(out, "families_cold", "rows") or {}
    if fam:
        rec["families_cold_frac"] = {k: _r(v.get("frac"), 3) for k, v in fam.items()}
        rec["families_cold_traffic_over_compulsory"] = {k: _r(v.get("traffic_over_compulsory"), 3) for k, v in fam.items()}
    col, cb = out.get("column") or {}, out.get("column_box_p4") or {}
    summ = {"column_solves_per_s": _r(col.get("schur_column_solves_per_s")), "schur_eta_ms": _r(col.get("schur_ms_all_columns")),
            "schur_eta_unresolved_columns": col.get("schur_unconverged_columns"),
            "schur_eta_columns_by_pivoted_lu": col.get("schur_columns_resolved_by_pivoted_lu"), "schur_eta_columns_accepted_on_backward_error": col.get("schur_columns_accepted_on_backward_error"), "schur_eta_pivot_fallback_cost_frac": _r(_g(col, "schur_pivot_fallback", "cost_frac"), 3),
            "schur_eta_ms_model_state": _r(_g(out, "column", "schur_model_state", "ms_all_columns")),
            "schur3_ms": _r(col.get("schur3_ms_all_columns")), "newton_iteration_ms": _r(col.get("vertical_newton_iteration_ms")),
            "newton_iteration_ms_cpp_host": _r(_g(out, "column", "vertical_newton_cpp_host", "ms_per_newton_iteration")),
            "box_p4_schur_eta_ms": _r(_g(cb, "schur_eta", "ms_all_columns")), "box_p4_schur3_ms": _r(_g(cb, "schur_3_box", "ms_all_columns")),
            "box_p4_umat_cold_frac": _r(_g(out, "box_p4", "roofline_cold", "frac"), 3),
            "horiz_rhs_ms": _r(_g(out, "horiz_rhs", "ms_per_evaluation_hipgraph")),
            "horiz_rhs_ms_reusing_grad_theta": _r(_g(out, "horiz_rhs", "ms_per_evaluation_hipgraph_reusing_grad_theta")),
            "horiz_m1_sweep_hbm_frac": _r(_g(out, "horiz_rhs", "m1_sweep_roofline", "frac"), 3),
            "horiz_rhs_ms_cpp_host": _r(_g(out, "horiz_rhs", "cpp_host", "ms_per_evaluation_recorded")),
            "horiz_rhs_ms_cpp_host_reusing_grad_theta": _r(_g(out, "horiz_rhs", "cpp_host", "ms_per_evaluation_recorded_reusing_grad_theta")),
            "sw_steps_per_s_config3": _r(_g(out, "sw", "config3_galewsky_24x24x6", "steps_per_s")),
            "sw_steps_per_s_config2": _r(_g(out, "sw", "config2_w2_16x16x6", "steps_per_s")),
            "sw_steps_per_s_config3_cpp_host": _r(_g(out, "sw", "config3_galewsky_24x24x6", "cpp_host", "graph", "steps_per_s")),
            "sw_steps_per_s_config2_cpp_host": _r(_g(out, "sw", "config2_w2_16x16x6", "cpp_host", "graph", "steps_per_s")),
            "reference_local_1_level_call_us": _r(_g(out, "reference_local_layout", "rows", "reference_local", "1_level_per_call", "us_per_call_wall")),
            "cpp_host_per_level_call_us": _r(_g(out, "reference_local_layout", "rows", "cpp_host", "per_level_calls_us_per_call")),
            "cpp_host_per_level_call_in_graph_us": _r(_g(out, "reference_local_layout", "rows", "cpp_host", "per_level_calls_in_a_graph_us_per_call"))}
    for k in ("weak_scaled", "umat_one_sided", "column_sharded", "horiz_sharded", "sw_sharded"):
        v = out.get(k)
        if isinstance(v, dict):
            summ[k] = {kk: _r(v[kk]) for kk in ("value", "ms_per_step", "schur_column_solves_per_s", "ms_per_evaluation", "steps_per_s", "fixed_length_iterations",
                                                "adaptive_iterations", "error") if kk in v}
            if k == "sw_sharded" and isinstance(v.get("one_sided_transport"), dict):
                o = v["one_sided_transport"]
                summ[k]["one_sided_steps_per_s"] = _r(o.get("steps_per_s")) if "steps_per_s" in o else str(o.get("error"))[:80]
    rec["summary"] = {k: v for k, v in summ.items() if v is not None}
    errs = [k for k, v in out.items() if isinstance(v, dict) and "error" in v]
    if errs:
        rec["extras_with_errors"] = errs
    if "extras_watchdog" in out:
        rec["extras_watchdog"] = out["extras_watchdog"]
    if extras_file:
        rec["extras_file"] = extras_file
    line = json.dumps(rec)
    if len(line) >= 4000:                                    # never let a long string cost the record: drop the prose first
        for key in ("cpu_baseline", "cpu_baseline_column", "cpu_baseline_sw"):
            if isinstance(rec.get(key), dict):
                rec[key].pop("sample", None)
        line = json.dumps(rec)
    return rec, line


def write_extras(out):
    """the full object (every extra, ~20 KB) next to the compact line: bench_extras.json at the repo root and under gpurun_out/"""
    written = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_extras.json"), "w") as f:
                json.dump(out, f)
            written = written or os.path.relpath(os.path.join(d, "bench_extras.json"), ROOT)
        except OSError:
            pass
    return written


def emit(out):
    ef = write_extras(out)
    _, line = compact_record(out, ef)
    sys.stdout.write(line + "\n"); sys.stdout.flush()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE): start N fresh ranks as a CHILD torch.distributed.run -- before this
    process has made any GPU call, never an exec -- relay its output (rank 0's compact line) and leave with its status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--families", action="store_true", help="(default at N = 1) every operator family with its roofline fraction, the p=4 box, the local layouts")
    ap.add_argument("--no-families", action="store_true", help="skip the per-family / box_p4 / reference_local_layout extras")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep extra (on by default at N = 1)")
    ap.add_argument("--column", action="store_true", help="also report the column (HEVI) path: Schur solves/s, transposes/s")
    ap.add_argument("--box", action="store_true", help="(default at N = 1, with the families) BASELINE config 5 grid (p=4, 32x32 periodic box x 64 levels) Umat apply")
    ap.add_argument("--pcie", action="store_true", help="extra: the same step with the input copied host->device and the result device->host "
                                                        "through the C ABI (the conservative MATSHELL binding of INTEGRATION.md section 2)")
    ap.add_argument("--horiz", action="store_true", help="extra: HorizSolve momentum_rhs_ec + advection_rhs_ec over all 30 levels (ms per evaluation; on by default at N = 1)")
    ap.add_argument("--no-horiz", action="store_true", help="skip the HorizSolve right-hand-side extra (~3 s)")
    ap.add_argument("--sweep", action="store_true", help="extra: SURVEY 8(d) batch-size sweep of six operator families (384 ... 1e6 element-level pairs)")
    ap.add_argument("--no-column", action="store_true", help="skip the column-solves/s extra (reported by default at N = 1, ~3 s)")
    ap.add_argument("--no-sw", action="store_true", help="skip the shallow-water time-steps/s extra (reported by default at N = 1, ~20 s)")
    ap.add_argument("--sw", action="store_true", help="extra: shallow-water Picard time steps/s (BASELINE configs 2 and 3 grids)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two child rocprofv3 --pmc passes that fill roofline.traffic (~25 s)")
    ap.add_argument("--no-horiz-sharded", action="store_true", help="N > 1: skip the sharded HorizSolve right-hand-side extra (~10 s)")
    ap.add_argument("--cold", type=int, default=8, metavar="R",
                    help="roofline_cold (not the headline value): the same step on R independent copies of the sphere, "
                         "working set >> the 256 MiB Infinity Cache, i.e. genuinely HBM-resident; 0 skips it")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))           # (nothing has touched the GPU yet: torch is imported below)
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    # rehearsal of the N > 1 control flow on a one-GPU box: every rank on device 0, gloo transport staged through the host
    # (RCCL refuses two ranks on one device).  Never used by the driver; numbers from such a run mean nothing.
    rehearsal = os.environ.get("MIMSEM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("MIMSEM_BENCH_FORCE_DIST") == "1"     # FORCE: rehearse the RCCL set-up on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.partition import HaloExchanger, build_plans, patches_of_rank
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import z_levels

    cs = CubedSphere(PN, NE, NPATCH)
    coords = sphere_coords(PN, NE)
    pids = patches_of_rank(NPATCH, world, rank)
    topos = [Topo(cs, p, NK) for p in pids]
    geoms = [Geom(t, cs, coords, NK) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(NK, g.n0))
    dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
    eng = Engine(dm, device=local_rank)
    rng = np.random.default_rng(20241024 + rank)
    x = eng.tensor(rng.standard_normal((NK, dm.n1)))
    y = eng.zeros(NK, dm.n1)
    deng = None
    safe_line = None
    headline_guard = None
    if use_dist:
        from mimsem_amd.distributed import DistEngine
        plans = build_plans(cs, world, rank, dm.gid0, dm.gid1)
        if world > 1:
            # SAFETY NET (round 6): no N > 1 RCCL exchange of this code has ever run on hardware.  Before the C ABI's own RCCL transport is set up
            # (an ncclComm_t made through ctypes next to torch's), the same step is measured over the most ordinary path there is -- the local
            # apply + torch.distributed.all_to_all_single (HaloExchanger) -- and kept as a minimal line; a timer prints THAT line and ends the
            # process should the set-up or the measurement below hang.  A normal run cancels the timer and never shows it.
            import threading
            d0 = DistEngine(eng, cs, world, rank, plans=plans)
            for _ in range(max(a.warmup, 2)):
                d0.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for _ in range(a.steps):
                d0.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            tsafe = torch.tensor([time.perf_counter() - ts0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tsafe, op=dist.ReduceOp.MAX)
            dts = tsafe.item()
            safe_line = {"metric": "element operator-applies/sec", "value": cs.ne * cs.ne * 6 * NK * a.steps / dts, "unit": "element operator-applies/s",
                         "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dts / a.steps, "higher_is_better": True,
                         "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                         "config": {"workload": "Umat (B1) matrix-free apply, p=3 24x24x6 cubed sphere x 30 levels (BASELINE config 4 grid)", "order": PN,
                                    "elements": cs.ne * cs.ne * 6, "levels": NK, "units_per_step": cs.ne * cs.ne * 6 * NK, "patches_per_gpu": len(pids),
                                    "halo_transport": "torch.distributed all_to_all_single (FALLBACK line: the C ABI's RCCL transport did not come up or hung)"}}
            guard_s = float(os.environ.get("MIMSEM_BENCH_HEADLINE_GUARD", "240"))

            def headline_bail():
                if rank == 0:
                    sys.stdout.write(json.dumps(safe_line) + "\n"); sys.stdout.flush()
                os._exit(4)
            headline_guard = threading.Timer(guard_s, headline_bail); headline_guard.daemon = True; headline_guard.start()
        # N > 1: the halo through the C ABI (mimsem_halo_create / _begin / _end): boundary wave-groups first, the exchange in flight on
        # the plan's communication stream -- grouped ncclSend / ncclRecv over xGMI on an ncclComm_t made here the way a C++ host would
        # (RcclComm); the one-GPU rehearsal has no RCCL between ranks of one device and uses the host-callback transport -- while the
        # interior groups are computed, then the unpack (what replaces MatMult + VecScatterBegin/End, eul/Assembly.cpp:2194-2195)
        deng = DistEngine(eng, cs, world, rank, overlap=True, transport="dist" if (rehearsal or dist.get_backend() != "nccl") else "auto",
                          plans=plans)

    apply_b1, _ = eng.prepare_apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)

    def step():
        if deng is not None:
            deng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)       # every copy of a shared edge holds the complete sum afterwards
        else:
            apply_b1()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    # the live kernel durations of the roofline entry: every n-th step of the timed region goes out with start / stop events (an event-timed
    # launch costs ~15 us of host time where a plain one costs 2: the samples perturb the region they are taken from): one step in 8 (three
    # samples at the driver's --steps 20), one in 16 from 200 steps on (a dozen samples and more)
    eng.set_profiling(int(os.environ.get('MIMSEM_BENCH_PROF_EVERY', str(max(8, min(16, a.steps // 12))))))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ms1, ms2, nl = eng.profile_read()
    eng.set_profiling(0)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # `value` is the contract's number: EXACTLY `steps` steps, once.  At the driver's --steps 20 that is a 0.4 ms sample, so the same timed
    # region is repeated a few more times (same step count, same fences, max over ranks each) and the median is reported BESIDE it
    rep_dts = [dt]
    for _ in range(max(0, int(os.environ.get("MIMSEM_BENCH_REPEATS", "7")) - 1)):
        fence()
        tr0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dr = time.perf_counter() - tr0
        if use_dist:
            t = torch.tensor([dr], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dr = t.item()
        rep_dts.append(dr)
    rep_sorted = sorted(rep_dts)
    dt_median = rep_sorted[len(rep_sorted) // 2]

    if headline_guard is not None:
        headline_guard.cancel()                     # the C ABI's transport came up and the timed regions finished: the safety net is not needed
    units_total = cs.ne * cs.ne * 6 * NK            # all ranks together
    units_rank = dm.nEl * NK
    value = units_total * a.steps / dt
    out = {
        "metric": "element operator-applies/sec", "value": value, "unit": "element operator-applies/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "ms_per_step_median": 1e3 * dt_median / a.steps, "value_median": units_total * a.steps / dt_median,
        "timed_region_repeats": {"count": len(rep_dts), "ms_per_step_min": 1e3 * rep_sorted[0] / a.steps, "ms_per_step_max": 1e3 * rep_sorted[-1] / a.steps,
                                 "note": "the first repeat is the contract's timed region (value, ms_per_step); the others repeat it unchanged"},
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Umat (B1, 1-form mass) matrix-free apply over every (element, level) pair of the "
                               "p=3 24x24x6 cubed sphere x 30 levels (BASELINE config 4 grid); N>1: patches dealt to the ranks, boundary "
                               "groups -> halo exchange through the C ABI (RCCL send/recv over xGMI) overlapped with the interior groups",
                   "order": PN, "elements": cs.ne * cs.ne * 6, "levels": NK, "units_per_step": units_total,
                   "patches": NPATCH, "patches_per_gpu": len(pids), "scale": SCALE, "level_chunk": None},
    }
    if safe_line is not None:
        out["config"]["same_step_over_torch_all_to_all_ms"] = safe_line["ms_per_step"]        # (the safety net's measurement, beside the C ABI's)
    if deng is not None:
        out["config"]["halo_transport"] = getattr(deng, "transport", None)
        if getattr(deng, "transport_note", None):
            out["config"]["halo_transport_note"] = deng.transport_note
    bm, kname, k2name, lch = launch_bytes(eng, dm, NK)
    out["config"]["level_chunk"] = lch
    out["config"]["form"] = "wave-level fused (k_apply_wave + k_wave_perim)" if "wave" in kname else "two-pass (k_elem_apply + k_gather_sum)"
    if nl:
        # N > 1: a step is TWO part launches of the operator (boundary groups, interior groups), each sampled apply one of them: per STEP
        # the kernel time is twice the per-launch average (the byte model is the rank's whole apply)
        pps = 2 if (deng is not None and deng.chalo is not None) else 1
        k1 = ms1 / nl * 1e-3 * pps
        k12 = (ms1 + ms2) / nl * 1e-3 * pps
        out["roofline"] = roofline_entry(bm, k1, k12, cache_resident=True,
                                         note="working set (~35 MB fields + metric) sits inside the 256 MiB Infinity Cache and is re-read "
                                              "every step: see roofline_cold for the HBM-resident workload", kname=kname, k2name=k2name)
        if world == 1:
            out["roofline"]["whole_operator"]["copy_of_the_same_bytes_us"] = copy_reference_us(bm["op_compulsory"], torch, eng.device)
            # NOT measured in this run: the PMC passes need rocprofv3 (scripts/pmc_traffic.py); the committed summary of the
            # same 103 680-unit launch is quoted for orientation only
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                out["roofline"]["traffic_from_committed_profile"] = {
                    "bytes_per_launch": min(pj["kernels"][kname], key=lambda r: r["grid_threads"])["total_bytes"],
                    "file": "profiles/pmc_traffic.json", "note": "rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE of an earlier run; a committed "
                    "constant, not an observation of this run"}
            except Exception:
                pass
    def box_extras():
        from mimsem_amd.geom import BoxGeom
        from mimsem_amd.mesh import PeriodicBox, box_coords
        bx = PeriodicBox(4, 32, 4); bc = box_coords(4, 32, 1000.0); nkb = 64
        bt = [Topo(bx, p, nkb) for p in range(4)]; bg = [BoxGeom(t, bx, bc, nkb, 1000.0) for t in bt]
        for g in bg:
            g.set_levels(np.repeat(np.linspace(0.0, 1500.0, nkb + 1)[:, None], g.n0, axis=1))
        dmb = DeviceMesh(bt, bg, nk=nkb, numbering="global"); engb = Engine(dmb, device=local_rank)
        xb = engb.tensor(rng.standard_normal((nkb, dmb.n1))); yb = engb.zeros(nkb, dmb.n1)
        callb, _ = engb.prepare_apply("UMAT", xb, lev0=0, scale=SCALE, flags=1, out=yb)
        for _ in range(10):
            callb()
        torch.cuda.synchronize(); engb.set_profiling(4); t1 = time.perf_counter()
        for _ in range(200):
            callb()
        torch.cuda.synchronize(); dtb = time.perf_counter() - t1
        b1, b2, bn = engb.profile_read(); engb.set_profiling(0)
        ub = dmb.nEl * nkb
        opb = family_bytes("UMAT", dmb.nEl, (dmb.n0, dmb.n1, dmb.n2), nkb, pn=4)
        kus = (b1 + b2) / bn * 1e3
        r = {"workload": "Umat apply, p=4 32x32 periodic box x 64 levels (65 536 units, BASELINE config 5 grid)", "value": ub * 200 / dtb,
             "kernel1_us": b1 / bn * 1e3, "kernel2_us": b2 / bn * 1e3,
             "roofline": {"bound": "hbm", "kernels": "k_apply_wave<4,UMAT> + k_wave_perim", "bytes_per_launch": opb, "avg_us": kus,
                          "achieved": opb / (kus * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": opb / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "byte_model": "whole operator: x and y once per level, thickInv per unit (25 points), metric record 32 B per point once per element",
                          "algorithmic_reference_bytes_per_unit": 2320}}
        # the same operator HBM resident: 8 independent boxes in one launch (524 288 units)
        del engb
        dmc = replicate(dmb, 8); engc = Engine(dmc, device=local_rank)
        xc = engc.tensor(rng.standard_normal((nkb, dmc.n1))); yc = engc.zeros(nkb, dmc.n1)
        callc, _ = engc.prepare_apply("UMAT", xc, lev0=0, scale=SCALE, flags=1, out=yc)
        for _ in range(3):
            callc()
        torch.cuda.synchronize(); engc.set_profiling(1)
        for _ in range(10):
            callc()
        torch.cuda.synchronize()
        c1, c2, cn = engc.profile_read(); engc.set_profiling(0)
        opc = family_bytes("UMAT", dmc.nEl, (dmc.n0, dmc.n1, dmc.n2), nkb, pn=4)
        kc = (c1 + c2) / cn * 1e3
        r["roofline_cold"] = {"bound": "hbm", "kernels": "k_apply_wave<4,UMAT> + k_wave_perim", "replicas": 8, "units": dmc.nEl * nkb, "bytes_per_launch": opc,
                              "avg_us": kc, "kernel1_us": c1 / cn * 1e3, "kernel2_us": c2 / cn * 1e3, "achieved": opc / (kc * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": opc / (kc * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_over_compulsory": None}
        del engc
        return r
    # N > 1: the extras below run collectives of paths that no multi-GPU hardware has executed yet.  A rank that fails alone leaves
    # its peers waiting in a collective; the headline measured above must survive that: past the budget rank 0 prints the line
    # with what is there and every rank leaves (os._exit: a blocked collective cannot be unwound).
    import threading
    printed = threading.Lock()

    in_flight = {"extra": None}

    def bail():
        if printed.acquire(blocking=False):
            out["extras_watchdog"] = {"note": "extras did not finish within %d s: line printed without the unfinished ones; exit status 3" % budget,
                                      "in_flight": in_flight["extra"]}
            if rank == 0:
                emit(out)
            os._exit(3)               # a hung collective is NOT a clean run: the headline survives, the status says what happened
    budget = int(os.environ.get("MIMSEM_BENCH_EXTRAS_BUDGET", "300"))
    watchdog = None
    if world > 1:
        watchdog = threading.Timer(budget, bail); watchdog.daemon = True; watchdog.start()

    def extra(key, fn):
        """an extra must never cost the headline line: a failure is reported under its key instead of aborting the run"""
        in_flight["extra"] = key
        try:
            out[key] = fn()
        except Exception as ex:          # noqa: BLE001 -- reported, not hidden
            import traceback
            out[key] = {"error": "%s: %s" % (type(ex).__name__, ex), "where": traceback.format_exc().strip().splitlines()[-3:]}

    if rank == 0 and world == 1 and not a.no_families:                 # SURVEY 8(d)'s per-family table: on by default at N = 1 (seconds)
        extra("families", lambda: families_extras(eng, dm, rng, torch))
        extra("box_p4", box_extras)
        extra("reference_local_layout", lambda: local_layout_extras(local_rank, rng, torch))
    if (a.column or not a.no_column) and rank == 0 and world == 1:     # the column half of the hot path: on by default at N = 1 (~3 s)
        extra("column", lambda: column_extras(eng, dm, rng, torch))
        extra("column_box_p4", lambda: column_box_p4_extras(local_rank, rng, torch))
    if a.pcie and rank == 0 and world == 1:
        import ctypes as C
        xh = np.ascontiguousarray(rng.standard_normal((NK, dm.n1))); yh = np.empty_like(xh)
        nbytes = xh.nbytes

        def step_pcie():
            eng.L.mimsem_memcpy_h2d(eng.ctx, C.c_void_p(x.data_ptr()), C.c_void_p(xh.ctypes.data), nbytes)
            apply_b1()
            eng.L.mimsem_memcpy_d2h(eng.ctx, C.c_void_p(yh.ctypes.data), C.c_void_p(y.data_ptr()), nbytes)
        for _ in range(3):
            step_pcie()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20):
            step_pcie()
        torch.cuda.synchronize(); elp = (time.perf_counter() - t1) / 20
        out["pcie_inclusive"] = {"value": units_rank / elp, "ms_per_step": 1e3 * elp, "bytes_each_way": nbytes,
                                 "GBs_each_way": nbytes / elp / 1e9 * 2 / 2,
                                 "note": "pageable host memory, synchronous hipMemcpy through mimsem_memcpy_h2d/_d2h; never the headline value"}
    def horiz_extras():
        from mimsem_amd.horizsolve import HorizSolve
        xqg = np.zeros((dm.nq, 3))
        for g in geoms:
            xqg[g.loc0] = coords[g.loc0]
        hs = HorizSolve(eng, quad_coords=xqg[dm.gidq])
        area = float(dm.det.mean()) * 4.0 / (PN * PN); dz = float(dm.thick.mean()); ln = area ** 0.5
        u1 = eng.tensor(rng.standard_normal((NK, dm.n1)) * 20.0 * ln * dz); u2 = u1 * 1.01
        h1 = eng.tensor(rng.uniform(0.8, 1.2, (NK, dm.n2)) * area * dz); h2 = h1 * 1.001
        th = eng.tensor(rng.uniform(290, 310, (NK, dm.n2)) * area * dz); Pi = eng.tensor(rng.uniform(900, 1000, (NK, dm.n2)) * area * dz)
        vz = eng.tensor(rng.standard_normal((NK - 1, dm.n2)) * area); dudz = eng.tensor(rng.standard_normal((NK - 1, dm.n1)) * 1e-3 * ln)

        def rhs():
            dF, dG, Fk, Gk = hs.advection_rhs_ec(u1, u2, h1, h2, th)
            return hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, Fx=Fk, Fk=Fk)
        ref = rhs(); torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(5):
            rhs()
        torch.cuda.synchronize(); el = (time.perf_counter() - t1) / 5
        its = hs.last_its
        hs.m1.fixed_its = 14                            # fixed-length PCG: the whole evaluation becomes one hipGraph
        graph, gout = eng.capture(rhs)
        graph.replay(); torch.cuda.synchronize()
        err = float(torch.linalg.vector_norm(gout - ref) / torch.linalg.vector_norm(ref))
        t1 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize(); elg = (time.perf_counter() - t1) / 20
        # the same evaluation with grad(theta) of advection_rhs_ec handed to momentum_rhs_ec (the reference solves that system twice per stage
        # for one answer, eul/HorizSolve.cpp:403 and :659): six 1-form mass solves instead of seven
        def rhs6():
            dF, dG, Fk, Gk = hs.advection_rhs_ec(u1, u2, h1, h2, th)
            return hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, Fx=Fk, Fk=Fk, dTheta=hs.dTheta)
        graph6, gout6 = eng.capture(rhs6)
        graph6.replay(); torch.cuda.synchronize()
        err6 = float(torch.linalg.vector_norm(gout6 - ref) / torch.linalg.vector_norm(ref))
        t1 = time.perf_counter()
        for _ in range(20):
            graph6.replay()
        torch.cuda.synchronize(); elg6 = (time.perf_counter() - t1) / 20
        hs.m1.fixed_its = 0
        checks_ok = hs.verify()                          # every fixed-length M1 solve of the replays logged its check norms: one read
        res = {"workload": "advection_rhs_ec + momentum_rhs_ec (viscosity on), 3456 elements x 30 levels per evaluation",
               "ms_per_evaluation_eager": 1e3 * el, "ms_per_evaluation_hipgraph": 1e3 * elg, "evaluations_per_s": 1.0 / elg,
               "m1_cg_iterations_eager": its, "graph_vs_eager_rel_diff": err,
               "ms_per_evaluation_hipgraph_reusing_grad_theta": 1e3 * elg6, "reusing_grad_theta_rel_diff": err6,
               "m1_checks": {"all_met": bool(checks_ok), "solves_checked": hs.m1.solves_checked, "solves_missed": hs.m1.solves_missed, "worst": hs.m1.worst_check}}
        # a roofline statement for the sweeps that are 73 % of this evaluation.  One Chebyshev sweep of the 1-form mass
        # solve = {element pass, block pass, gather epilogue}; COMPULSORY bytes: x, rhs, p read + x, p written (5 vectors of nk n1 doubles),
        # thickInv per (unit, point), the preconditioner's per-(level, element) factor, and once per launch the element blocks
        # (24 x 24 doubles per element) and the metric (32 B per element point).  Time: a whole solve under HIP events / its step count.
        if hs.m1.chebyshev and hs.m1._cheb is not None:
            b1 = hs.m1.apply(u1)
            for _ in range(2):
                hs.m1.solve(b1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nsolve = 10
            e0.record()
            for _ in range(nsolve):
                hs.m1.solve(b1)
            e1.record(); torch.cuda.synchronize()
            steps = hs.m1._cheb.steps
            us = e0.elapsed_time(e1) * 1e3 / nsolve / steps
            mp12 = (PN + 1) ** 2
            nbytes = 5 * NK * dm.n1 * 8 + dm.nEl * NK * mp12 * 8 + dm.nEl * NK * 8 + dm.nEl * (2 * PN * (PN + 1)) ** 2 * 8 + dm.nEl * mp12 * 32
            res["m1_sweep_roofline"] = {"bound": "hbm", "kernel": "k_elem_apply<3,UMAT> + k_blocks_residual<3,8> + k_gather_epilogue<2> (one Chebyshev sweep, 103 680 units)",
                                        "avg_sweep_us": us, "steps_per_solve": steps, "bytes_per_sweep": nbytes, "achieved": nbytes / (us * 1e-6) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "note": "compulsory bytes; the three launches exchange two element-local arrays (24 doubles per unit each, written once and "
                                                "read twice through the gather plan): ~2.0x the compulsory traffic by construction (DESIGN 8)"}
            hs.verify()
        # the same evaluation with the HOST in C++ (mimsem_amd/host/horiz_call.cpp over mimsem_horizsolve.hpp; the ksp1 solves are the
        # library's batched CG with its convergence test on the host: not recorded as a graph)
        import subprocess
        import tempfile
        from mimsem_amd.workloads import mesh_arrays, write_arrays
        exe = os.path.join(ROOT, "mimsem_amd", "host", "horiz_call")
        if os.path.exists(exe):
            with tempfile.TemporaryDirectory() as tmp:
                case = os.path.join(tmp, "case.arr")
                arr = mesh_arrays(dm)
                cpu = lambda t: t.cpu().numpy()
                arr.update(fg=cpu(hs.fg) if hs.fg.shape[0] == NK else np.broadcast_to(cpu(hs.fg), (NK, dm.n0)), u1=cpu(u1), u2=cpu(u2), h1=cpu(h1),
                           h2=cpu(h2), theta=cpu(th), Pi=cpu(Pi), velz=cpu(vz), dudz=cpu(dudz))
                write_arrays(case, arr)
                try:
                    r = subprocess.run([exe, case, "10"], capture_output=True, text=True, timeout=300)
                    res["cpp_host"] = json.loads(r.stdout) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
                    if "fu_l2" in res["cpp_host"]:
                        res["cpp_host"]["fu_l2_rel_diff_to_python_host"] = abs(res["cpp_host"]["fu_l2"] - float(torch.linalg.vector_norm(ref))) / float(torch.linalg.vector_norm(ref))
                except Exception as e:                     # noqa: BLE001
                    res["cpp_host"] = {"error": repr(e)[:300]}
        else:
            res["cpp_host"] = {"error": "mimsem_amd/host/horiz_call not built (__graft_entry__.build())"}
        return res
    if (a.horiz or not a.no_horiz) and rank == 0 and world == 1:      # row N2: the right-hand sides of the horizontal dynamics, on by default at N = 1
        extra("horiz_rhs", horiz_extras)
    if (a.sw or not a.no_sw) and rank == 0 and world == 1:        # the second half of BASELINE's metric: on by default at N = 1
        extra("sw", lambda: sw_extras(local_rank, torch))
    if (a.sweep or not a.no_sweep) and rank == 0 and world == 1:
        extra("sweep", lambda: sweep_extras(local_rank, torch))
    if rank == 0 and world == 1 and a.cold != 0:
        # the HBM number: on by default (--cold 0 skips it), R = 8 spheres = 829 440 units per launch
        def cold_all():
            rc_, fam = cold_workload(dm, a.cold, local_rank, rng, torch)
            out["families_cold"] = fam
            return rc_
        extra("roofline_cold", cold_all)
    if world > 1:
        # WEAK-scaled companion of the headline: every rank holds 8 spheres' worth of work whatever N is (its 24/N patches, 8 N
        # independent copies: 829 440 units and ~1 GB per rank, the size of roofline_cold), the halo of all copies in ONE exchange,
        # overlapped with the interior groups -- the strong-scaled 21 us step above is smaller than one exchange and can only measure
        # RCCL latency
        def weak_scaled():
            import types
            from mimsem_amd.partition import CHalo
            R = 8 * world
            dmw = replicate(dm, R)
            engw = Engine(dmw, device=local_rank)
            p1 = plans[1]
            rep = lambda d: {r: np.concatenate([np.asarray(v, dtype=np.int64) + k * dm.n1 for k in range(R)]).astype(np.int32) for r, v in d.items()}
            pw = types.SimpleNamespace(gids=np.arange(dmw.n1), ghost_slots=rep(p1.ghost_slots), mirror_slots=rep(p1.mirror_slots))
            pw.neighbours = p1.neighbours
            tr = "dist" if deng.transport != "rccl" else deng.rccl.comm
            chw = CHalo(pw, engw, max_nlev=NK, transport=tr)
            engw.set_halo_slots(1, chw.shared)
            xw = engw.tensor(rng.standard_normal((NK, dmw.n1))); yw = engw.zeros(NK, dmw.n1)

            def stepw():
                try:
                    engw.apply_part("UMAT", "boundary", xw, lev0=0, scale=SCALE, flags=1, out=yw)
                    tok = chw.begin("pair", yw, True)
                    engw.apply_part("UMAT", "interior", xw, lev0=0, scale=SCALE, flags=1, out=yw)
                except Exception:
                    engw.reset_parts()        # (a pending BOUNDARY part would refuse every later split apply on this context)
                    raise
                chw.end(tok)
            for _ in range(3):
                stepw()
            fence(); t1 = time.perf_counter()
            nst = 20
            for _ in range(nst):
                stepw()
            fence(); el = time.perf_counter() - t1
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            halo_bytes = int(sum(len(v) for v in pw.ghost_slots.values()) + sum(len(v) for v in pw.mirror_slots.values())) * NK * 8
            chw.close()
            return {"workload": "Umat apply, %d copies of the rank's %d patches x 30 levels per rank (fixed work per rank), halo of all copies "
                                "in one exchange overlapped with the interior groups" % (R, len(pids)),
                    "scaling": "weak", "units_per_rank": dmw.nEl * NK, "value": world * dmw.nEl * NK * nst / tt.item(),
                    "unit": "element operator-applies/s", "ms_per_step": 1e3 * tt.item() / nst, "halo_bytes_sent_per_rank_per_step": halo_bytes,
                    "transport": deng.transport}
        extra("weak_scaled", weak_scaled)

        # the HEADLINE step over the one-sided transport (hipIpc-opened receive buffers, two kernels per exchange, no library call): never run
        # between GPUs before this -- reported beside the headline (never as it), and only if every rank's result equals the headline
        # transport's bit for bit and no wait timed out
        def umat_one_sided():
            eng2 = Engine(dm, device=local_rank)
            d2 = DistEngine(eng2, cs, world, rank, overlap=True, transport="peer", plans=plans)
            y2 = eng2.zeros(NK, dm.n1)
            step()                                                   # (y: the headline transport's result for this x)
            for _ in range(3):
                d2.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y2)
            torch.cuda.synchronize()
            okl = 1.0 if (torch.equal(y2, y) and not d2.chalo.peer_timeouts()) else 0.0
            flag = torch.tensor([okl], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() != 1.0:
                return {"error": "result differs from the headline transport's or an exchange timed out (this rank: equal %s, time-outs %s)" % (bool(torch.equal(y2, y)), d2.chalo.peer_timeouts())}
            fence(); t1 = time.perf_counter()
            for _ in range(a.steps):
                d2.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y2)
            fence(); el = time.perf_counter() - t1
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bad = bool(d2.chalo.peer_timeouts())
            d2.close()
            if bad:
                return {"error": "an exchange timed out during the timed steps"}
            return {"workload": "the headline step (boundary groups | exchange | interior groups) with the exchange on the one-sided transport", "value": cs.ne * cs.ne * 6 * NK * a.steps / tt.item(),
                    "ms_per_step": 1e3 * tt.item() / a.steps, "equal_to_headline_transport": True, "headline_ms_per_step": 1e3 * dt / a.steps}
        extra("umat_one_sided", umat_one_sided)
    if world > 1 and not a.no_column:
        # the column half of the hot path sharded: all nk levels of an element live on one GPU, so the Schur solves and the Newton loop
        # need no halo at all -- only the MPI_Allreduce(MAX) of the four norms per iteration.  Aggregate = all columns / slowest rank.
        def column_sharded():
            from mimsem_amd.distributed import DistEngine
            from mimsem_amd.geom import gll_points
            from mimsem_amd.vertsolve import VertSolve
            nEl, nk, n2 = dm.nEl, NK, eng.n2e
            area = float(dm.det.mean()) * 4.0 / n2; dz = float(dm.thick.mean())
            lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz)
            theta, rho, eta, pi = lev(nk, 280, 320), lev(nk, 0.5, 1.2), lev(nk, 5, 6), lev(nk, 700, 1000)
            F = [eng.tensor(rng.standard_normal((nEl, n * n2)) * 1e8) for n in (nk - 1, nk, nk, nk)]
            for _ in range(2):
                eng.solve_schur_eta(75.0, theta, rho, eta, pi, *[f.clone() for f in F])
            fence(); t1 = time.perf_counter()
            for _ in range(5):
                eng.solve_schur_eta(75.0, theta, rho, eta, pi, *[f.clone() for f in F])
            fence(); ts = (time.perf_counter() - t1) / 5
            wd = np.diff(gll_points(PN)); wj = np.outer(wd, wd).ravel()
            cell = dm.det.mean(axis=1)[:, None, None] * dm.thick.mean(axis=2).T[:, :, None] * wj[None, None, :]
            zl = np.mean([g.levs.mean(axis=1) for g in geoms], axis=0); zm = 0.5 * (zl[:-1] + zl[1:])
            th_v = 300.0 + 0.004 * zm
            pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
            rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
            colv = lambda v: eng.tensor((cell * v[None, :, None]).reshape(nEl, nk * n2) * (1.0 + 1e-4 * rng.standard_normal((nEl, nk * n2))))
            vs = VertSolve(DistEngine(eng, cs, world, rank), 75.0)
            levs = np.zeros((nk + 1, dm.nq))
            for g in geoms:
                levs[:, np.searchsorted(dm.gidq, g.loc0[np.arange(g.n0)])] = g.levs
            zv = vs.init_gz(levs)
            st = (eng.zeros(nEl, (nk - 1) * n2), colv(rho_v), colv(rho_v * th_v), colv(pi_v))
            vs.solve_schur_eta(*st, zv, maxit=2, tol=0.0)
            fence(); t1 = time.perf_counter()
            vs.solve_schur_eta(*st, zv, maxit=4, tol=0.0)
            fence(); tn = (time.perf_counter() - t1) / 4
            tt = torch.tensor([ts, tn], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ts, tn = tt.tolist()
            ncol = cs.ne * cs.ne * 6
            return {"workload": "solve_schur_column_eta and one Newton iteration of VertSolve::solve_schur_eta, 3 456 columns x 30 levels dealt "
                                "to the ranks (no halo: columns are rank-local; one all-reduce of four norms per iteration)",
                    "schur_column_solves_per_s": ncol / ts, "schur_ms": ts * 1e3, "newton_iteration_ms": tn * 1e3,
                    "newton_column_iterations_per_s": ncol / tn, "columns_per_rank": nEl}
        extra("column_sharded", column_sharded)

        # N2 sharded: HorizSolve's right-hand sides (advection_rhs_ec + momentum_rhs_ec, viscosity on: ~40 operator applies, 7 mass solves,
        # every 0/1-form result completed over the halo) on the ranks' own patches x 30 levels -- milliseconds of work per rank, the
        # workload on which sharding the config-4 grid can pay (the 21 us Umat step cannot: it is smaller than one exchange)
        def horiz_sharded():
            from mimsem_amd.distributed import DistEngine
            from mimsem_amd.horizsolve import HorizSolve
            xqg = np.zeros((int(max(g.loc0.max() for g in geoms)) + 1, 3))
            for g in geoms:
                xqg[g.loc0] = coords[g.loc0]
            hs = HorizSolve(deng, quad_coords=xqg[dm.gidq])            # the headline's DistEngine: every completion through the C ABI's halo plans
            area = float(dm.det.mean()) * 4.0 / (PN * PN); dz = float(dm.thick.mean()); ln = area ** 0.5
            rg = np.random.default_rng(777)                                   # the same global fields on every rank
            U1 = rg.standard_normal((NK, cs.nDofs1G)) * 20.0 * ln * dz
            H1 = rg.uniform(0.8, 1.2, (NK, cs.nDofs2G)) * area * dz
            TH = rg.uniform(290, 310, (NK, cs.nDofs2G)) * area * dz; PI = rg.uniform(900, 1000, (NK, cs.nDofs2G)) * area * dz
            VZ = rg.standard_normal((NK - 1, cs.nDofs2G)) * area; DU = rg.standard_normal((NK - 1, cs.nDofs1G)) * 1e-3 * ln
            u1 = eng.tensor(U1[:, dm.gid1]); u2 = u1 * 1.01; h1 = eng.tensor(H1[:, dm.gid2]); h2 = h1 * 1.001
            th = eng.tensor(TH[:, dm.gid2]); Pi = eng.tensor(PI[:, dm.gid2]); vz = eng.tensor(VZ[:, dm.gid2]); dudz = eng.tensor(DU[:, dm.gid1])

            def rhs():
                dF, dG, Fk, Gk = hs.advection_rhs_ec(u1, u2, h1, h2, th)
                return hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, Fx=Fk, Fk=Fk)
            rhs(); fence(); t1 = time.perf_counter()
            for _ in range(3):
                rhs()
            fence(); el = (time.perf_counter() - t1) / 3
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return {"workload": "advection_rhs_ec + momentum_rhs_ec (viscosity on), 3456 elements x 30 levels dealt to the ranks, halo per operator "
                                "(C ABI plans, transport %s; Umat / Uhmat / RotMat applies split boundary | exchange | interior)" % deng.transport,
                    "ms_per_evaluation": 1e3 * tt.item(), "evaluations_per_s": 1.0 / tt.item(), "elements_per_rank": dm.nEl}
        if not a.no_horiz_sharded:
            extra("horiz_sharded", horiz_sharded)
    if world > 1 and not a.no_sw:
        # the SW step on the ranks' shards (config 3 as the reference driver runs it: the Galewsky jet on the 24x24x6 sphere, dt = 360 s, 2 Picard
        # iterations, upwinded q) in the FIXED-LENGTH mode over the halo (round 6): Chebyshev solves with the exchanges inside, no all-reduce in any
        # solve, one all-reduce of the check norms per Picard iteration.  93 312 unknowns over N GPUs: latency-bound by construction (~145
        # kB-sized exchanges per Picard iteration) -- the number says what xGMI point-to-point latency costs, not what the kernels do
        def sw_sharded():
            from mimsem_amd.distributed import DistEngine
            from mimsem_amd.sweqn import SWEqn, galewsky
            t1s = [Topo(cs, p, 1) for p in pids]
            g1s = [Geom(t, cs, coords, 1, signed_det=True) for t in t1s]
            for g in g1s:
                g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
            dms = DeviceMesh(t1s, g1s, nk=1, numbering="global")
            engs = Engine(dms, device=local_rank)
            xqs = np.zeros((int(max(g.loc0.max() for g in g1s)) + 1, 3))
            for g in g1s:
                xqs[g.loc0] = coords[g.loc0]
            uq, hq = galewsky(torch.as_tensor(xqs[dms.gidq], device=engs.device))
            nst = 5 if rehearsal else 20

            def run(transport):
                eng_t = Engine(dms, device=local_rank)
                des = DistEngine(eng_t, cs, world, rank, overlap=True, transport=transport)
                S = SWEqn(des, xqs[dms.gidq])
                us, hs_ = S.init1(uq), S.init2(hq)
                for _ in range(2):
                    us, hs_ = S.solve(us, hs_, 360.0, nits=2, q_exact=False)
                    if transport == "peer":
                        # (a neighbour that never publishes costs a bounded 2 s wait PER exchange: leave at the first sign of one, on every rank)
                        bad = torch.tensor([1.0 if (des.chalo.peer_timeouts() or des.chalo0.peer_timeouts()) else 0.0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                        if bad.item() > 0:
                            raise RuntimeError("one-sided transport: an exchange timed out during the warm-up steps")
                f0, a0 = S.fixed_iterations, S.adaptive_iterations
                fence(); t1 = time.perf_counter()
                for _ in range(nst):
                    us, hs_ = S.solve(us, hs_, 360.0, nits=2, q_exact=False)
                fence(); els = (time.perf_counter() - t1) / nst
                tt = torch.tensor([els], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                st = dict(S.its)
                rec = {"steps_per_s": 1.0 / tt.item(), "ms_per_step": 1e3 * tt.item(), "steps_timed": nst, "chebyshev_steps": st, "transport": getattr(des, "transport", "dist"),
                       "fixed_length_iterations": S.fixed_iterations - f0, "adaptive_iterations": S.adaptive_iterations - a0, "recalibrations": S.recalibrations,
                       "picard_iteration_recorded_as_a_graph": bool(S._pg is not None and getattr(S._pg, "record", False)),
                       "all_reduces_per_picard_iteration": 1, "exchanges_in_solves_per_picard_iteration_approx": 2 * st.get("A", 0) + 2 * st.get("F", 0) + 2 * st.get("q", 0)}
                if transport == "peer":
                    rec["exchanges_timed_out"] = des.chalo.peer_timeouts() or des.chalo0.peer_timeouts()
                return rec, us, des
            # the transport of the headline's exchange: the C ABI's plans on the rank's RCCL communicator (host-staged callback in the rehearsal)
            res, u_ref, des_ref = run(deng.rccl if deng.transport == "rccl" else "dist")
            res["workload"] = ("SWEqn::solve, config 3 (Galewsky jet, dt = 360 s, 2 Picard iterations, upwinded q), 24x24x6 sphere sharded over the ranks, "
                               "fixed-length Chebyshev solves over the halo")
            # ... and the ONE-SIDED transport (hipIpc-opened receive buffers, kernels only, the Picard iteration recorded as a graph per rank): verified
            # on one GPU between processes; between GPUs this run is its first A/B -- reported only if its state equals the other transport's
            try:
                rp, u_p, des_p = run("peer")
                same = float(torch.linalg.vector_norm(u_p - u_ref) / torch.linalg.vector_norm(u_ref))
                flag = torch.tensor([1.0 if (same < 1e-10 and not rp["exchanges_timed_out"]) else 0.0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                rp["state_rel_diff_to_the_other_transport"] = same
                res["one_sided_transport"] = rp if flag.item() == 1.0 else {"error": "state differs or an exchange timed out on some rank", **{k: rp[k] for k in ("exchanges_timed_out",)}, "state_rel_diff": same}
            except Exception as ex:                 # noqa: BLE001 -- the first transport's number stands
                res["one_sided_transport"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
            return res
        extra("sw_sharded", sw_sharded)
    if rank == 0 and world == 1 and not a.no_pmc and "roofline" in out:
        # roofline.traffic: HBM-side bytes per launch of the dominant kernel from PMC counters collected in THIS run (child rocprofv3
        # passes); traffic_frac = those bytes over the in-run kernel time, against the 8 TB/s peak
        try:
            pm = measure_pmc_traffic()
            kn = out["roofline"]["kernel"]
            launches = sorted(pm["kernels"].get(kn, []), key=lambda r: r["grid_threads"])
            if launches:
                for key, rec in (("roofline", launches[0]), ("roofline_cold", launches[-1])):
                    if key in out and "avg_kernel_us" in out[key] and (key == "roofline" or len(launches) > 1):
                        out[key]["traffic"] = rec["total_bytes"]
                        out[key]["traffic_GBs"] = rec["total_bytes"] / (out[key]["avg_kernel_us"] * 1e-6) / 1e9
                        out[key]["traffic_frac"] = out[key]["traffic_GBs"] / HBM_PEAK_GBS
                        out[key]["traffic_note"] = ("FETCH_SIZE x %.2f (calibrated on a launch of known byte count) + WRITE_SIZE of this launch, "
                                                    "child rocprofv3 --pmc passes of this run (scripts/pmc_traffic.py)" % pm["fetch_correction"])
                # the whole operator: both kernels' HBM-side bytes against every input and output once (<= 1.25x asked)
                k2n = out["roofline"]["whole_operator"]["kernels"].split(" + ")[-1]
                l2 = sorted(pm["kernels"].get(k2n, []), key=lambda r: r["grid_threads"])
                if l2:
                    for key, r1, r2 in (("roofline", launches[0], l2[0]), ("roofline_cold", launches[-1], l2[-1])):
                        if key in out and "whole_operator" in out[key] and (key == "roofline" or (len(launches) > 1 and len(l2) > 1)):
                            w = out[key]["whole_operator"]
                            w["traffic"] = r1["total_bytes"] + r2["total_bytes"]
                            w["traffic_over_compulsory"] = w["traffic"] / w["bytes_per_launch"]
                out["roofline"].pop("traffic_from_committed_profile", None)
                out["pmc_traffic"] = pm
            # per-family traffic on the 8-sphere workload: the family's element kernel (the launch with the larger grid) + the perimeter pass
            fam_kernel = {"UMAT": "k_apply_wave<3,UMAT>", "UHMAT": "k_apply_wave<3,UHMAT>", "ROTMAT": "k_apply_wave<3,ROTMAT>",
                          "WMAT": "k_elem_apply<3,WMAT>", "WTQUMAT": "k_apply_wave2<3,WTQUMAT>", "WHMAT": "k_apply_wave2<3,WHMAT>"}
            big = lambda name: max(pm["kernels"].get(name, []), key=lambda r_: r_["grid_threads"], default=None)
            perim = big("k_wave_perim")
            for row in (out.get("families_cold") or {}).get("rows", {}).values():
                k1r = big(fam_kernel[row["op"]])
                if k1r is None:
                    continue
                tb = k1r["total_bytes"] + (perim["total_bytes"] if (row["kernels"] == 2 and perim) else 0.0)
                row["traffic"] = tb; row["traffic_over_compulsory"] = tb / row["bytes_per_launch"]
                row["traffic_GBs"] = tb / (row["kernel_us"] * 1e-6) / 1e9
            bc = (out.get("box_p4") or {}).get("roofline_cold")
            kb = pm["kernels"].get("k_apply_wave<4,UMAT>", [])
            if bc and kb:
                kb1 = max(kb, key=lambda r_: r_["grid_threads"])
                pb = [r_ for r_ in pm["kernels"].get("k_wave_perim", []) if r_ is not perim]
                # (the box's perimeter launch: the largest k_wave_perim grid that is not the sphere's)
                pbx = max(pb, key=lambda r_: r_["grid_threads"], default=None) if pb else None
                bc["traffic"] = kb1["total_bytes"] + (pbx["total_bytes"] if pbx else 0.0)
                bc["traffic_over_compulsory"] = bc["traffic"] / bc["bytes_per_launch"]
        except Exception as ex:          # noqa: BLE001 -- the headline line never depends on the profiler being usable
            out["roofline"]["traffic_note"] = "PMC passes not available in this run (%s: %s): traffic stays null" % (type(ex).__name__, str(ex)[:200])
    if rank == 0 and world == 1 and not a.no_cpu:
        def cpu_all():
            b1, bc, bs = cpu_baseline()
            out["cpu_baseline_column"] = bc; out["cpu_baseline_sw"] = bs
            return b1
        extra("cpu_baseline", cpu_all)
    if watchdog is not None:
        watchdog.cancel()
    if not printed.acquire(blocking=False):
        time.sleep(60)                                   # the watchdog is printing: it ends the process
    if rank == 0:
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
