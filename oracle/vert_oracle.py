"""oracle/vert_oracle.py -- numpy restatement of the vertical implicit Newton loop VertSolve::solve_schur_eta
(eul/VertSolve.cpp:1721-1973) with its residual assembly (:237-286, :432-502), column by column as the reference does,
on the dense column matrices and vector functions of the C oracle (pyoracle.Patch).  TEST INFRASTRUCTURE ONLY: the checker of
mimsem_amd/vertsolve.py::VertSolve.solve_schur_eta; nothing in the product imports it.  No horizontal wind (dFx = dGx = 0)."""
import numpy as np

RAYLEIGH = 4.0 / 120.0      # eul/VertSolve.cpp:32


def _v10(nk, n2):
    N, Nm = nk * n2, (nk - 1) * n2
    V = np.zeros((N, Nm))
    for k in range(nk):
        for i in range(n2):
            if k > 0: V[k * n2 + i, (k - 1) * n2 + i] = -1.0
            if k < nk - 1: V[k * n2 + i, k * n2 + i] = +1.0
    return V


def assemble_residual_ec(P, ex, ey, dt, theta, Pi, velz1, velz2, rho1, rho2, zv, V10):
    """:432-502 (+ diagnose_F_z :237-260, diagnose_Phi_z :262-286) -> fw, F, G, f_theta_corr"""
    D = lambda op, **kw: P.colop_dense(op, ex, ey, **kw)
    V01 = -V10.T
    VAinv = D("LINEAR_INV")
    F = VAinv @ D("LINEAR_RT", flag=1, f1=rho1) @ (velz1 / 3 + velz2 / 6) + VAinv @ D("LINEAR_RT", flag=1, f1=rho2) @ (velz1 / 6 + velz2 / 3)
    W1, W2 = D("CONLIN_W", f1=velz1), D("CONLIN_W", f1=velz2)
    Phi = (W1 @ velz1 + W1 @ velz2 + W2 @ velz2) / 6 + zv
    VA, VB = D("LINEAR"), D("CONST")
    fw = VA @ velz2 - VA @ velz1 + dt * V01 @ Phi
    tA2 = VAinv @ (V01 @ (VB @ Pi))
    VAt = D("LINEAR_RT", flag=1, f1=theta)
    fw = fw + 0.5 * dt * (VAt @ tA2)
    G = VAinv @ (VAt @ F)
    VR = D("RAYLEIGH")
    fw = fw + 0.5 * dt * RAYLEIGH * (VR @ velz2 + VR @ velz1)
    tA2 = VAinv @ (V01 @ (VB @ theta))
    VBt = D("CONST_RHO", f1=theta)
    VBA = D("CONLIN_W", f1=tA2)
    fw = fw + 0.5 * dt * (V01 @ (VBt @ Pi)) - 0.5 * dt * (VBA.T @ Pi)
    ftc = 0.5 * dt * (VBt @ (V10 @ F)) + 0.5 * dt * (VBA @ F)
    return fw, F, G, ftc


def solve_schur_eta(P, dt, velz_i, rho_i, rt_i, exner_i, zv, nits, hs_forcing=False, udwdx=None, columns=None):
    """`nits` Newton iterations of :1721-1973 for every column of the patch; arrays [nEl][slots*n2e]; returns the new state and
    the max-norm history.  hs_forcing: the Held-Suarez temperature forcing of :1831-1834; udwdx: the optional F_w term of :1809.
    columns: optional list of element indices -- only those columns are advanced (the others keep their input state and the
    max-norms run over the subset): columns are independent, which is what makes a sampled full-size comparison possible."""
    nEl, nk, n2 = P.nEl, P.nk, P.n2e
    cols = list(range(nEl)) if columns is None else [int(c) for c in columns]
    V10 = _v10(nk, n2)
    velz_j, rho_j, rt_j, exner_j = velz_i.copy(), rho_i.copy(), rt_i.copy(), exner_i.copy()
    def col(f):
        first = f(cols[0] % P.nElsX, cols[0] // P.nElsX, cols[0])
        out = np.zeros((nEl, first.size))
        for e in cols:
            out[e] = f(e % P.nElsX, e // P.nElsX, e)
        return out
    theta_l2_i = col(lambda ex, ey, e: P.diag_theta_L2(ex, ey, rho_i[e], rt_i[e]))
    theta_l2_h = theta_l2_i.copy()
    theta_i = col(lambda ex, ey, e: P.diag_theta2(ex, ey, rho_i[e], rt_i[e]))
    theta_h = theta_i.copy()
    exner_h, velz_h, rho_h, rt_h = exner_i.copy(), velz_i.copy(), rho_i.copy(), rt_i.copy()
    hist = []
    for _ in range(nits):
        mx = dict(exner=0.0, w=0.0, rho=0.0, eta=0.0)
        for e in cols:
            ex, ey = e % P.nElsX, e // P.nElsX
            D = lambda op, **kw: P.colop_dense(op, ex, ey, **kw)
            F_w, F_z, G_z, ftc = assemble_residual_ec(P, ex, ey, dt, theta_l2_h[e], exner_h[e], velz_i[e], velz_j[e], rho_i[e], rho_j[e], zv[e], V10)
            if udwdx is not None:
                F_w = F_w + dt * udwdx[e]
            F_exner = P.eos_residual(ex, ey, rt_j[e], exner_j[e])
            VB = D("CONST")
            dF_z = rho_j[e] + dt * (V10 @ F_z) - rho_i[e]
            dG_z = rt_j[e] + 0.5 * dt * (V10 @ G_z) - rt_i[e]
            F_rho = VB @ dF_z
            F_rt = VB @ dG_z + ftc
            if hs_forcing:
                F_rt = F_rt + dt * P.temp_forcing_hs(ex, ey, exner_h[e], theta_h[e], rho_h[e])
            t1 = D("CONST_RHO_INV", f1=rt_h[e]) @ F_rt - D("CONST_RHO_INV", f1=rho_h[e]) @ F_rho
            F_eta = VB @ t1
            th_w3 = D("CONST_RHO_INV", f1=rho_h[e]) @ (VB @ rt_h[e])
            VBinv = D("CONST_INV")
            eta = VBinv @ P.const_log_theta_plus_eta(ex, ey, th_w3, None)
            sol = P.solve_schur_column_eta(ex, ey, dt, th_w3, rho_h[e], eta, exner_h[e], F_w, F_rho, F_eta, F_exner)
            d_w, d_rho, d_eta, d_exner = sol["d_u"], sol["d_rho"], sol["d_eta"], sol["d_pi"]
            th_w3 = D("CONST_RHO_INV", f1=rho_j[e]) @ (VB @ rt_j[e])
            eta = VBinv @ P.const_log_theta_plus_eta(ex, ey, th_w3, d_eta)
            velz_j[e] += d_w; rho_j[e] += d_rho; exner_j[e] += d_exner
            rt_j[e] = VBinv @ P.const_rho_exp_eta(ex, ey, rho_j[e], eta)
            for k, (dx, x) in dict(exner=(d_exner, exner_j[e]), w=(d_w, velz_j[e]), rho=(d_rho, rho_j[e]), eta=(d_eta, eta)).items():
                mx[k] = max(mx[k], np.linalg.norm(dx) / np.linalg.norm(x))
            exner_h[e] = 0.5 * exner_i[e] + 0.5 * exner_j[e]; velz_h[e] = 0.5 * velz_i[e] + 0.5 * velz_j[e]
            rho_h[e] = 0.5 * rho_i[e] + 0.5 * rho_j[e]; rt_h[e] = 0.5 * rt_i[e] + 0.5 * rt_j[e]
        theta_h = 0.5 * col(lambda ex, ey, e: P.diag_theta2(ex, ey, rho_j[e], rt_j[e])) + 0.5 * theta_i
        theta_l2_j = col(lambda ex, ey, e: P.diag_theta_L2(ex, ey, rho_j[e], rt_j[e]))
        theta_l2_h = 0.5 * theta_l2_j + 0.5 * theta_l2_i
        hist.append(mx)
    return velz_j, rho_j, rt_j, exner_j, hist
