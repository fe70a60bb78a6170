"""Device-resident Krylov solves for the mass-matrix systems that follow almost every operator application in
the reference (KSPSolve(ksp1, b, x): GMRES + block-Jacobi on M1, rtol 1e-16; eul/HorizSolve.cpp:77-96, 224, 246).
SURVEY 8(f) row N1 -- the first "next" row after the operator engine.

The mass matrices are symmetric positive definite, so a preconditioned CG converges to the same (unique) solution
the reference's GMRES does; all levels are solved at once (one independent system per level, per-level scalars kept
on the device: no host synchronisation inside the iteration).  Mat-vecs are the matrix-free engine applies."""
import torch


def pcg(apply_A, b, minv=None, x0=None, rtol=1e-14, maxit=300, check_every=10, allreduce=None):
    """Solve A x = b for a batch of systems (rows of b).  apply_A(x)->A x on [nlev, n] tensors.
    minv: elementwise preconditioner (Jacobi) of the same shape.  allreduce(t): sums per-level scalars over ranks
    (multi-GPU: each rank holds ghost copies, the caller's dot weights handle ownership)."""
    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - apply_A(x) if x0 is not None else b.clone()
    z = r * minv if minv is not None else r
    p = z.clone()
    dot = (lambda u, v: (u * v).sum(dim=1)) if allreduce is None else (lambda u, v: allreduce((u * v).sum(dim=1)))
    rz = dot(r, z)
    bnorm = torch.sqrt(dot(b, b)).clamp_min(1e-300)
    its = 0
    for it in range(maxit):
        Ap = apply_A(p)
        alpha = rz / dot(p, Ap).clamp_min(1e-300)
        x += alpha[:, None] * p
        r -= alpha[:, None] * Ap
        z = r * minv if minv is not None else r
        rz_new = dot(r, z)
        beta = rz_new / rz.clamp_min(1e-300)
        p = z + beta[:, None] * p
        rz = rz_new
        its = it + 1
        if its % check_every == 0:                    # the only host synchronisation
            if bool((torch.sqrt(dot(r, r)) / bnorm).max() < rtol):
                break
    return x, its


class MassSolver:
    """M1 u = b on all levels (the ksp1 solves).  Jacobi preconditioner from the diagonal of the element blocks
    (built once per thickness field, reused over time steps)."""

    def __init__(self, eng, scale=1.0e8, vert_scale=True):
        self.eng, self.scale, self.flags = eng, scale, 1 if vert_scale else 0
        n1e = eng.n1e
        dm = eng.mesh
        ix = torch.as_tensor(dm.inds1x, device=eng.device).long()
        iy = torch.as_tensor(dm.inds1y, device=eng.device).long()
        diag = eng.zeros(eng.nk, dm.n1)
        for k in range(eng.nk):
            em = eng.element_matrices("UMAT", lev=k, scale=scale, flags=self.flags).view(eng.nEl, 4, n1e, n1e)
            diag[k].index_add_(0, ix.reshape(-1), torch.diagonal(em[:, 0], dim1=1, dim2=2).reshape(-1))
            diag[k].index_add_(0, iy.reshape(-1), torch.diagonal(em[:, 3], dim1=1, dim2=2).reshape(-1))
        self.minv = 1.0 / diag

    def apply(self, x, lev0=0):
        return self.eng.apply("UMAT", x, lev0=lev0, scale=self.scale, flags=self.flags)

    def solve(self, b, lev0=0, rtol=1e-14, maxit=300):
        nlev = b.shape[0]
        return pcg(lambda v: self.apply(v, lev0), b, minv=self.minv[lev0:lev0 + nlev], rtol=rtol, maxit=maxit)
