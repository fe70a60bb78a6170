"""Shape fuzz of the fused column solves (round 4, after a spill lost in a divergent region had corrupted one instantiation only): for many
(order, elements, levels, levels-per-task) the three-launch solve_schur_column_3 (both flavours) against round 2's chain, the three-launch
solve_schur_column_eta against the un-fused chain, each run twice (same bits), and the pivoted band LU for every column
(mimsem_column_set_pivot_fallback(2)) against the block sweep."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import make_patch, rel_l2
from tests.test_gpu_column import _col_fields

pytestmark = pytest.mark.gpu

TOL2 = 2e-9
CASES = [(2, 2, 4), (2, 3, 9), (3, 1, 4), (3, 2, 13), (3, 3, 10), (3, 2, 30), (4, 1, 4), (4, 2, 11), (4, 3, 7), (4, 1, 16)]


@pytest.mark.parametrize("chunk", [None, 2, 3, 5], ids=lambda c: "chunk_%s" % c)
@pytest.mark.parametrize("pn,ne,nk", CASES, ids=lambda v: str(v))
def test_fused_column_solves_over_shapes_and_task_splits(oracle, pn, ne, nk, chunk):
    from mimsem_amd.device import DeviceMesh, Engine
    old = os.environ.get("MIMSEM_SWEEP_CHUNK")
    if chunk:
        os.environ["MIMSEM_SWEEP_CHUNK"] = str(chunk)              # levels per task of every chunked walk (read per call)
    else:
        os.environ.pop("MIMSEM_SWEEP_CHUNK", None)
    try:
        cs, topo, geom, P, rng = make_patch(oracle, pn, ne, 6, 1, nk=nk, seed=900 + 13 * nk + pn)
        eng = Engine(DeviceMesh([topo], [geom], nk=nk, numbering="local"))
        F = _col_fields(P, seed=nk + 3 * pn)
        r = np.random.default_rng(5 + nk)
        nEl, n2 = P.nEl, P.n2e
        rhs = [r.standard_normal((nEl, n * n2)) * 1e8 for n in (nk - 1, nk, nk, nk)]
        t = eng.tensor

        def against(run, env):
            a = run(); b = run()
            assert all(torch.equal(x, y) for x, y in zip(a, b))        # run to run: the same bits
            os.environ[env[0]] = env[1]
            try:
                c = run()
            finally:
                del os.environ[env[0]]
            for x, y in zip(a, c):                                     # two solvers of the same systems: each within ~1e-10 of its own solution on
                assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < TOL2    # these rough columns (cond up to ~1e8 at nk = 30): see the error budgets of test_gpu_column.py
            return a
        for flags in (0, 3):
            against(lambda: eng.solve_schur_3(75.0, t(F["theta"]), t(F["velz"]), t(F["rho"]), t(F["rt"]), t(F["pi"]), *[t(x) for x in rhs],
                                              want_L=True, flags=flags), ("MIMSEM_SCHUR3_CHAIN", "1"))
        run = lambda: eng.solve_schur_eta(75.0, t(F["thetaL"]), t(F["rho"]), t(F["eta"]), t(F["pi"]), *[t(x) for x in rhs])
        a = against(run, ("MIMSEM_SCHUR_FUSED", "0"))
        eng.set_pivot_fallback(2)
        try:
            c = run(); st = eng.solve_status()[1]
        finally:
            eng.set_pivot_fallback(1)
        assert (st == 3).all()
        for x, y in zip(a, c):
            assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < TOL2
    finally:
        if old is None:
            os.environ.pop("MIMSEM_SWEEP_CHUNK", None)
        else:
            os.environ["MIMSEM_SWEEP_CHUNK"] = old
