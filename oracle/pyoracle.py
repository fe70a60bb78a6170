"""ctypes doorway to the CPU oracle (oracle/liboracle.so) and, where present, to the compiled
reference kernels (oracle/_ref/libmimsem_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from mimsem_amd/ (the product path is HIP and fails loudly without it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)


def build(ref=True):
    """(Re)build liboracle.so, and oracle/_ref when /root/reference is present in this container."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref and os.path.isdir("/root/reference/eul"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _dp(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "oracle wants contiguous float64"
    return a.ctypes.data_as(_DP)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_IP)


class _Patch(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("n", "m", "np1", "mp1", "mp12", "n0e", "n1e", "n2e", "nElsX", "nEl", "nDofsX", "nk",
                 "n0", "n1x", "n1y", "n1", "n2", "nqX", "n0q")] + \
               [(k, _DP) for k in
                ("qx", "qw", "nx", "ljxi", "ejxi", "P", "U", "V", "W", "Q", "Pt", "Ut", "Vt", "Wt",
                 "det", "J", "thick", "thickInv", "xq", "sq")]


class _LinAlg(C.Structure):
    _fields_ = [("mult", C.c_void_p), ("mult_fd", C.c_void_p), ("mult_df", C.c_void_p),
                ("tran", C.c_void_p), ("axb", C.c_void_p), ("inv", C.c_void_p)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
        _lib.orc_patch_create.restype = C.POINTER(_Patch)
        _lib.orc_node_eval.restype = C.c_double
        _lib.orc_node_deriv.restype = C.c_double
        _lib.orc_edge_eval.restype = C.c_double
        _lib.orc_csr_create.restype = C.c_void_p
        _lib.orc_bench_assemble_mult.restype = C.c_double
    return _lib


def ref_lib():
    """The reference's own LinAlg.cpp/Basis.cpp compiled in place; None when not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libmimsem_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        for f in ("ref_node_eval_q", "ref_node_deriv", "ref_edge_eval"):
            getattr(_ref, f).restype = C.c_double
    return _ref


def use_reference_linalg(on=True):
    """Route every oracle assembly through the reference's compiled dense kernels (or back)."""
    L = lib()
    if not on:
        L.orc_set_linalg(None)
        return True
    R = ref_lib()
    if R is None:
        return False
    la = _LinAlg(*[C.cast(getattr(R, n), C.c_void_p).value for n in
                   ("ref_Mult_IP", "ref_Mult_FD_IP", "ref_Mult_DF_IP", "ref_Tran_IP", "ref_Ax_b", "ref_Inv")])
    L.orc_set_linalg(C.byref(la))
    return True


# ---------------------------------------------------------------------------------------------
OPS = dict(UMAT=0, WMAT=1, UHMAT=2, PMAT=3, PHMAT=4, WTQUMAT=5, ROTMAT=6, WHMAT=7, UTMAT=8,
           UTMAT_H=9, UTQWMAT=10, WTQDUDZ=11, WMATINV=12, WHMATINV=13)
COLOPS = dict(CONST=0, CONST_INV=1, CONST_RHO=2, CONST_RHO_INV=3, CONST_THETA=4, EOS_BLOCK=5,
              LINEAR=6, LINEAR_INV=7, LINEAR_RT=8, LINEAR_THETA=9, LINEAR_RHO2=10, RAYLEIGH=11,
              LINCON=12, LINCON2=13, CONLIN=14, CONLIN_W=15, CONLIN_RHODPI=16,
              LINEAR_RAYLEIGH_INV=17, EOS_BLOCK_INV=18, LINEAR_RHO2_UP=19, LINCON2_UP=20)


def gll(n):
    x = np.zeros(n + 1); w = np.zeros(n + 1)
    rc = lib().orc_gll(n, _dp(x), _dp(w))
    return x, w, rc


def tables(n, m):
    L = lib()
    mp12 = (m + 1) ** 2
    out = dict(ljxi=np.zeros((m + 1, n + 1)), ejxi=np.zeros((m + 1, n)),
               P=np.zeros((mp12, (n + 1) ** 2)), U=np.zeros((mp12, (n + 1) * n)),
               V=np.zeros((mp12, (n + 1) * n)), W=np.zeros((mp12, n * n)), Q=np.zeros(mp12))
    L.orc_node_table(n, m, _dp(out["ljxi"])); L.orc_edge_table(n, m, _dp(out["ejxi"]))
    L.orc_tab_P(n, m, _dp(out["P"])); L.orc_tab_U(n, m, _dp(out["U"])); L.orc_tab_V(n, m, _dp(out["V"]))
    L.orc_tab_W(n, m, _dp(out["W"])); L.orc_tab_Q(m, _dp(out["Q"]))
    return out


def inv(A):
    n = A.shape[0]
    A = np.ascontiguousarray(A, dtype=np.float64); out = np.zeros_like(A)
    err = lib().orc_inv(_dp(A), _dp(out), n)
    return out, err


class Patch:
    """One reference-rank worth of mesh (Topo+Geom) held by the oracle."""

    def __init__(self, n, m, nElsX, nk):
        self.L = lib()
        self.p = self.L.orc_patch_create(n, m, nElsX, nk)
        self.c = self.p.contents
        for k, _ in _Patch._fields_[:19]:
            setattr(self, k, getattr(self.c, k))

    def __del__(self):
        try:
            self.L.orc_patch_destroy(self.p)
        except Exception:
            pass

    def arr(self, name, shape):
        return np.ctypeslib.as_array(getattr(self.c, name), shape=shape)

    # geometry ---------------------------------------------------------------------------
    def set_sphere_geometry(self, coords, radius=6371220.0, abs_det=True):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        assert coords.shape == (self.n0q, 3)
        self.L.orc_patch_set_sphere_geometry(self.p, _dp(coords), C.c_double(radius), int(abs_det))

    def set_levels(self, levs):
        levs = np.ascontiguousarray(levs, dtype=np.float64)
        assert levs.shape == (self.nk + 1, self.n0q)
        self.L.orc_patch_set_levels(self.p, _dp(levs))

    def set_metric(self, det, J):
        det = np.ascontiguousarray(det, dtype=np.float64); J = np.ascontiguousarray(J, dtype=np.float64)
        assert det.shape == (self.nEl, self.mp12) and J.shape == (self.nEl, self.mp12, 4)
        self.L.orc_patch_set_metric(self.p, _dp(det), _dp(J))

    @property
    def det(self): return self.arr("det", (self.nEl, self.mp12))
    @property
    def J(self): return self.arr("J", (self.nEl, self.mp12, 4))
    @property
    def thick(self): return self.arr("thick", (self.nk, self.n0q))
    @property
    def thickInv(self): return self.arr("thickInv", (self.nk, self.n0q))
    @property
    def xq(self): return self.arr("xq", (self.n0q, 3))

    # index maps ---------------------------------------------------------------------------
    def elinds(self, kind):
        fn, cnt = dict(n0=(self.L.orc_elinds0_l, self.n0e), n1x=(self.L.orc_elinds1x_l, self.n1e),
                       n1y=(self.L.orc_elinds1y_l, self.n1e), n2=(self.L.orc_elinds2_l, self.n2e),
                       q=(self.L.orc_elindsq_l, self.mp12))[kind]
        out = np.zeros((self.nEl, cnt), dtype=np.int32)
        tmp = np.zeros(cnt, dtype=np.int32)
        for ey in range(self.nElsX):
            for ex in range(self.nElsX):
                fn(self.p, ex, ey, _ip(tmp))
                out[ey * self.nElsX + ex] = tmp
        return out

    def interp(self, kind, ex, ey, px, py, vec):
        """Geom::interp0 / interp1_l / interp2_l / interp1_g / interp2_g at one quadrature point (eul/Geom.cpp:328-417)"""
        fn = {"0": self.L.orc_interp0, "1l": self.L.orc_interp1_l, "2l": self.L.orc_interp2_l,
              "1g": self.L.orc_interp1_g, "2g": self.L.orc_interp2_g}[kind]
        val = np.zeros(2)
        fn(self.p, ex, ey, px, py, _dp(vec), _dp(val))
        return val

    # horizontal operators -----------------------------------------------------------------
    def elmat_size(self, op):
        return self.L.orc_op_elmat_size(self.p, OPS[op])

    def op_elmats(self, op, lev=0, scale=1.0, flag=0, f1=None):
        out = np.zeros((self.nEl, self.elmat_size(op)))
        rc = self.L.orc_op_elmats(self.p, OPS[op], lev, C.c_double(scale), int(flag), _dp(f1), _dp(out))
        assert rc == 0
        return out

    def op_apply(self, op, elmats, x, ny):
        y = np.zeros(ny)
        rc = self.L.orc_op_apply(self.p, OPS[op], _dp(elmats), _dp(np.ascontiguousarray(x)), _dp(y))
        assert rc == 0
        return y

    def out_size(self, op):
        return dict(UMAT=self.n1, UHMAT=self.n1, UTMAT=self.n1, UTMAT_H=self.n1, ROTMAT=self.n1,
                    UTQWMAT=self.n1, WMAT=self.n2, WHMAT=self.n2, WMATINV=self.n2, WHMATINV=self.n2,
                    WTQUMAT=self.n2, WTQDUDZ=self.n2, PMAT=self.n0, PHMAT=self.n0)[op]

    def apply(self, op, x, lev=0, scale=1.0, flag=0, f1=None):
        """assemble(...) then MatMult on local vectors: the reference's two-call idiom."""
        em = self.op_elmats(op, lev, scale, flag, f1)
        return self.op_apply(op, em, x, self.out_size(op))

    def apply_up(self, which, x, fac, dt, f1, ul):
        """Phmat::assemble_up (which=0) / RotMat_up::assemble (which=1) followed by MatMult, local vectors"""
        op = "PMAT" if which == 0 else "ROTMAT"
        em = np.zeros((self.nEl, self.elmat_size(op)))
        rc = self.L.orc_op_elmats_up(self.p, which, C.c_double(fac), C.c_double(dt), _dp(f1), _dp(ul), _dp(em))
        assert rc == 0
        return self.op_apply(op, em, x, self.out_size(op)), em

    def project_from_quad(self, which, xq):
        """WtQmat / PtQmat / UtQmat applied to a quad-grid field (B7)"""
        y = np.zeros([self.n2, self.n0, self.n1][which])
        rc = self.L.orc_project_from_quad(self.p, which, _dp(np.ascontiguousarray(xq)), _dp(y)); assert rc == 0
        return y

    def apply_testup(self, which, x, lev, scale, tau, f1, f2, transpose=False):
        """Umat::assemble_up / Uhmat::assemble_up then MatMult (or MatMult with MT), local vectors"""
        em = np.zeros((self.nEl, self.elmat_size("UMAT")))
        rc = self.L.orc_op_elmats_testup(self.p, which, lev, C.c_double(scale), C.c_double(tau), _dp(f1), _dp(f2), _dp(em))
        assert rc == 0
        if transpose:
            n = self.n1e
            b = em.reshape(self.nEl, 4, n, n)
            em = np.ascontiguousarray(np.stack([b[:, 0].transpose(0, 2, 1), b[:, 2].transpose(0, 2, 1),
                                                b[:, 1].transpose(0, 2, 1), b[:, 3].transpose(0, 2, 1)], axis=1)).reshape(self.nEl, -1)
        return self.op_apply("UMAT", em, x, self.n1)

    def umat_ray(self, x, lev, scale, dt, exner_k, exner_s):
        """Umat_ray::assemble then MatMult on local vectors; also returns the element blocks"""
        em = np.zeros((self.nEl, self.elmat_size("UMAT")))
        rc = self.L.orc_umat_ray_elmats(self.p, lev, C.c_double(scale), C.c_double(dt), _dp(exner_k), _dp(exner_s), _dp(em))
        assert rc == 0
        return self.op_apply("UMAT", em, x, self.n1), em

    def uvec_hu_up(self, lev, scale, vel, rho, fac, tau, vel2):
        v = np.zeros(self.n1)
        self.L.orc_uvec_hu_up(self.p, lev, C.c_double(scale), _dp(vel), _dp(rho), C.c_double(fac), C.c_double(tau), _dp(vel2), _dp(v))
        return v

    def bench_assemble_mult(self, op, x, reps, lev=0, scale=1.0, flag=0, f1=None):
        """seconds for `reps` x (assemble + MatMult) with the reference's CSR cost structure; also returns y"""
        y = np.zeros(self.out_size(op))
        sec = self.L.orc_bench_assemble_mult(self.p, OPS[op], lev, C.c_double(scale), int(flag), _dp(f1),
                                             _dp(np.ascontiguousarray(x)), _dp(y), int(reps))
        assert sec >= 0
        return sec, y

    def pvec(self, lev, scale):
        v = np.zeros(self.n0); self.L.orc_pvec(self.p, lev, C.c_double(scale), _dp(v)); return v

    def phvec(self, lev, scale, h2):
        v = np.zeros(self.n0); self.L.orc_phvec(self.p, lev, C.c_double(scale), _dp(h2), _dp(v)); return v

    def uvec(self, lev, scale, vel):
        v = np.zeros(self.n1); self.L.orc_uvec(self.p, lev, C.c_double(scale), 1, _dp(vel), _dp(v)); return v

    def uvec_hu(self, lev, scale, vel, rho, fac):
        v = np.zeros(self.n1)
        self.L.orc_uvec_hu(self.p, lev, C.c_double(scale), _dp(vel), _dp(rho), C.c_double(fac), _dp(v)); return v

    def uvec_wxu(self, lev, scale, vel, vort):
        v = np.zeros(self.n1); self.L.orc_uvec_wxu(self.p, lev, C.c_double(scale), _dp(vel), _dp(vort), _dp(v)); return v

    def wvec(self, lev, scale, vert_scale, rho):
        """Wvec::assemble with Wt = W^T (corrected restatement of eul/Assembly.cpp:2457-2495)"""
        v = np.zeros(self.n2); self.L.orc_wvec(self.p, lev, C.c_double(scale), int(vert_scale), _dp(rho), _dp(v)); return v

    def wvec_K(self, lev, scale, vel1, vel2):
        """Wvec::assemble_K with Wt = W^T (eul/Assembly.cpp:2497-2545)"""
        v = np.zeros(self.n2); self.L.orc_wvec_K(self.p, lev, C.c_double(scale), _dp(vel1), _dp(vel2), _dp(v)); return v

    def e10(self, x0):
        y = np.zeros(self.n1); self.L.orc_e10_apply(self.p, _dp(x0), _dp(y)); return y

    def e21(self, x1):
        y = np.zeros(self.n2); self.L.orc_e21_apply(self.p, _dp(x1), _dp(y)); return y

    # column operators -----------------------------------------------------------------------
    def colop_dims(self, colop):
        r = C.c_int(); c = C.c_int()
        self.L.orc_colop_dims_ex(self.p, COLOPS[colop], C.byref(r), C.byref(c))
        return r.value, c.value

    def colop_dense_ex(self, colop, ex, ey, param=0.0, f1=None, f2=None, uh=None, flag=0):
        r, c = self.colop_dims(colop)
        out = np.zeros((r, c))
        rc = self.L.orc_colop_dense_ex(self.p, COLOPS[colop], ex, ey, int(flag), C.c_double(param), _dp(f1), _dp(f2), _dp(uh), _dp(out))
        assert rc == 0
        return out

    def diag_theta_up(self, ex, ey, dt, rho, rt, uh):
        th = np.zeros((self.nk + 1) * self.n2e)
        rc = self.L.orc_diag_theta_up(self.p, ex, ey, C.c_double(dt), _dp(rho), _dp(rt), _dp(uh), _dp(th)); assert rc == 0; return th

    def temp_forcing_hs(self, ex, ey, exner, theta, rho):
        o = np.zeros(self.nk * self.n2e)
        self.L.orc_temp_forcing_hs(self.p, ex, ey, _dp(exner), _dp(theta), _dp(rho), _dp(o)); return o

    def solve_schur_column_3(self, ex, ey, dt, theta, velz, rho, rt, pi, F_u, F_rho, F_rt, F_pi, flags=0):
        N = self.nk * self.n2e; Nm = (self.nk - 1) * self.n2e
        F_u, F_rho, F_rt, F_pi = (np.array(a, dtype=np.float64) for a in (F_u, F_rho, F_rt, F_pi))
        d_u = np.zeros(Nm); d_rho = np.zeros(N); d_rt = np.zeros(N); d_pi = np.zeros(N); L = np.zeros((N, N))
        rc = self.L.orc_solve_schur_column_3(self.p, ex, ey, C.c_double(dt), int(flags), _dp(theta), _dp(velz), _dp(rho), _dp(rt), _dp(pi),
                                             _dp(F_u), _dp(F_rho), _dp(F_rt), _dp(F_pi),
                                             _dp(d_u), _dp(d_rho), _dp(d_rt), _dp(d_pi), _dp(L))
        assert rc == 0
        return dict(d_u=d_u, d_rho=d_rho, d_rt=d_rt, d_pi=d_pi, L=L, F_u=F_u, F_rho=F_rho, F_rt=F_rt, F_pi=F_pi)

    @property
    def sq(self): return self.arr("sq", (self.n0q, 2))

    def colop_dense(self, colop, ex, ey, flag=0, f1=None, f2=None):
        r, c = self.colop_dims(colop)
        out = np.zeros((r, c))
        rc = self.L.orc_colop_dense(self.p, COLOPS[colop], ex, ey, int(flag), _dp(f1), _dp(f2), _dp(out))
        assert rc == 0
        return out

    def eos_residual(self, ex, ey, rt, exner):
        o = np.zeros(self.nk * self.n2e); self.L.orc_eos_residual(self.p, ex, ey, _dp(rt), _dp(exner), _dp(o)); return o

    def eos_rhs(self, ex, ey, rt, factor, exponent):
        o = np.zeros(self.nk * self.n2e)
        self.L.orc_eos_rhs(self.p, ex, ey, _dp(rt), C.c_double(factor), C.c_double(exponent), _dp(o)); return o

    def const_log_theta_plus_eta(self, ex, ey, theta, eta=None):
        o = np.zeros(self.nk * self.n2e)
        self.L.orc_const_log_theta_plus_eta(self.p, ex, ey, _dp(theta), _dp(eta), _dp(o)); return o

    def const_rho_exp_eta(self, ex, ey, rho, eta):
        o = np.zeros(self.nk * self.n2e)
        self.L.orc_const_rho_exp_eta(self.p, ex, ey, _dp(rho), _dp(eta), _dp(o)); return o

    def horiz_to_vert(self, vh):
        vz = np.zeros((self.nEl, self.nk * self.n2e)); self.L.orc_horiz_to_vert(self.p, _dp(vh), _dp(vz)); return vz

    def vert_to_horiz(self, vz):
        vh = np.zeros((self.nk, self.n2)); self.L.orc_vert_to_horiz(self.p, _dp(vz), _dp(vh)); return vh

    def diag_theta_L2(self, ex, ey, rho, rt):
        th = np.zeros(self.nk * self.n2e)
        rc = self.L.orc_diag_theta_L2(self.p, ex, ey, _dp(rho), _dp(rt), _dp(th)); assert rc == 0; return th

    def diag_theta2(self, ex, ey, rho, rt):
        th = np.zeros((self.nk + 1) * self.n2e)
        rc = self.L.orc_diag_theta2(self.p, ex, ey, _dp(rho), _dp(rt), _dp(th)); assert rc == 0; return th

    def solve_schur_column_eta(self, ex, ey, dt, theta, rho, eta, pi, F_u, F_rho, F_eta, F_pi):
        N = self.nk * self.n2e; Nm = (self.nk - 1) * self.n2e
        F_u, F_rho, F_eta, F_pi = (np.array(a, dtype=np.float64) for a in (F_u, F_rho, F_eta, F_pi))
        d_u = np.zeros(Nm); d_rho = np.zeros(N); d_eta = np.zeros(N); d_pi = np.zeros(N); Lpi = np.zeros((N, N))
        velz = np.zeros(Nm)
        rc = self.L.orc_solve_schur_column_eta(self.p, ex, ey, C.c_double(dt), _dp(theta), _dp(velz), _dp(rho),
                                               _dp(eta), _dp(pi), _dp(F_u), _dp(F_rho), _dp(F_eta), _dp(F_pi),
                                               _dp(d_u), _dp(d_rho), _dp(d_eta), _dp(d_pi), _dp(Lpi))
        assert rc == 0
        return dict(d_u=d_u, d_rho=d_rho, d_eta=d_eta, d_pi=d_pi, L_pi=Lpi,
                    F_u=F_u, F_rho=F_rho, F_eta=F_eta, F_pi=F_pi)
