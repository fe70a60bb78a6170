"""Pin the oracle (CPU restatement) against the reference: golden vectors generated from the
reference's own compiled eul/Basis.cpp + eul/LinAlg.cpp (tests/golden/basis_linalg.npz), the live
oracle/_ref library when it is built, and the known-answer identities of SURVEY 8(c)."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "basis_linalg.npz"))


def test_gll_bit_exact(oracle, G):
    for n in range(1, 8):
        x, w, rc = oracle.gll(n)
        assert rc == 0
        assert np.array_equal(x, G[f"gll_x_{n}"]) and np.array_equal(w, G[f"gll_w_{n}"])
        assert abs(w.sum() - 2.0) <= 1e-8            # eul/Basis.cpp:91-97 self check
    assert oracle.gll(8)[2] != 0                      # invalid order is flagged


def test_tables_bit_exact(oracle, G):
    for key in G.files:
        if not key.startswith("ljxi_"):
            continue
        _, n, m = key.split("_"); n, m = int(n), int(m)
        t = oracle.tables(n, m)
        assert np.array_equal(t["ljxi"], G[f"ljxi_{n}_{m}"])
        assert np.array_equal(t["ejxi"], G[f"ejxi_{n}_{m}"])


def test_point_evals_bit_exact(oracle, G):
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for n in (3, 4):
        xn, _, _ = oracle.gll(n)
        for a, x in enumerate(G["eval_pts"]):
            for i in range(n + 1):
                assert L.orc_node_eval(n, dp(xn), C.c_double(x), i) == G[f"node_eval_{n}"][a, i]
                assert L.orc_node_deriv(n, dp(xn), C.c_double(x), i) == G[f"node_deriv_{n}"][a, i]
            for i in range(n):
                assert L.orc_edge_eval(n, dp(xn), C.c_double(x), i) == G[f"edge_eval_{n}"][a, i]


def test_dense_kernels_bit_exact(oracle, G):
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for tag in ("u", "w", "p", "wu", "p4"):
        A, B, d = (np.ascontiguousarray(G[f"mm_{k}_{tag}"]) for k in ("A", "B", "d"))
        ni, nk = A.shape; nj = B.shape[1]
        Cm = np.zeros((ni, nj)); FD = np.zeros((ni, nk)); T = np.zeros((nk, ni)); y = np.zeros(ni)
        L.orc_mult(ni, nj, nk, dp(A), dp(B), dp(Cm))
        L.orc_mult_fd(ni, nk, nk, dp(A), dp(d), dp(FD))
        L.orc_tran(ni, nk, dp(A), dp(T))
        L.orc_axb(ni, nk, dp(A), dp(d), dp(y))
        assert np.array_equal(Cm, G[f"mm_C_{tag}"]) and np.array_equal(FD, G[f"mm_FD_{tag}"])
        assert np.array_equal(T, G[f"mm_T_{tag}"]) and np.array_equal(y, G[f"mm_y_{tag}"])


def test_inv_bit_exact_and_error_codes(oracle, G):
    for tag in ("4", "9", "16", "perm"):
        Ai, err = oracle.inv(G[f"inv_A_{tag}"])
        assert err == int(G[f"inv_err_{tag}"][0])
        assert np.array_equal(Ai, G[f"inv_Ai_{tag}"])
        assert np.allclose(Ai @ G[f"inv_A_{tag}"], np.eye(Ai.shape[0]), atol=1e-12)   # Inv round trip
    _, err = oracle.inv(G["inv_A_sing"])
    assert err == int(G["inv_err_sing"][0]) and err != 0


def test_known_answers(oracle):
    for n in range(1, 8):
        t = oracle.tables(n, n)
        assert np.array_equal(t["ljxi"], np.eye(n + 1))          # l_j(x_i) = delta_ij, exactly
        assert abs(t["Q"].sum() - 4.0) < 1e-12                    # sum_q w_q = 4 per element
        # histopolation int_{x_i}^{x_{i+1}} e_j = delta_ij  =>  int_{-1}^{1} e_j dx = 1 for every j
        # (GLL(n) is exact here: degree n-1 <= 2n-1)
        x, w, _ = oracle.gll(n)
        assert np.allclose(w @ t["ejxi"], np.ones(n), atol=1e-13)


def test_live_reference_library_matches(oracle):
    """If oracle/_ref is built here, the whole assembly restatement is re-run with the reference's
    compiled dense kernels plugged in and must not change by a single bit."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(7)
    P = oracle.Patch(3, 3, 2, 2)
    det = rng.uniform(0.5, 2.0, (P.nEl, P.mp12)); J = rng.standard_normal((P.nEl, P.mp12, 4))
    P.set_metric(det, J)
    P.set_levels(np.cumsum(rng.uniform(1, 2, (3, P.n0q)), axis=0))
    h2 = rng.standard_normal(P.n2); u1 = rng.standard_normal(P.n1); q0 = rng.standard_normal(P.n0)
    cases = [("UMAT", 1, None), ("WMAT", 1, None), ("UHMAT", 1, h2), ("ROTMAT", 0, q0), ("WTQUMAT", 0, u1),
             ("WHMAT", 1, h2), ("PMAT", 0, None), ("WMATINV", 0, None), ("UTQWMAT", 0, u1)]
    own = [P.op_elmats(op, 1, 1e8, fl, f) for op, fl, f in cases]
    rt = np.abs(rng.standard_normal(P.nk * P.n2e)) + 1.0
    own_col = P.colop_dense("EOS_BLOCK", 1, 0, f1=rt)
    assert oracle.use_reference_linalg(True)
    try:
        P2 = oracle.Patch(3, 3, 2, 2)      # tables rebuilt through the reference's Tran_IP
        P2.set_metric(det, J); P2.thick[:] = P.thick; P2.thickInv[:] = P.thickInv
        for (op, fl, f), a in zip(cases, own):
            assert np.array_equal(P2.op_elmats(op, 1, 1e8, fl, f), a), op
        assert np.array_equal(P2.colop_dense("EOS_BLOCK", 1, 0, f1=rt), own_col)
    finally:
        oracle.use_reference_linalg(False)
    assert own_col.shape == (P.nk * P.n2e, P.nk * P.n2e)
