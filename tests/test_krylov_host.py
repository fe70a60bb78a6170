"""Host-side logic of the fixed-length solves (no GPU): the Lanczos interval with its residual bounds, the safety margins that follow them
(round 6), the Chebyshev coefficients of a real interval / an ellipse and the convergence they promise -- on small dense problems in numpy /
CPU torch.  The device side of the same solvers is tests/test_gpu_next_rows.py, test_gpu_sweqn.py."""
import math

import numpy as np
import pytest
import torch

from mimsem_amd.krylov import ChebyshevMass, arnoldi_ritz, chebyshev_ellipse_coefs, chebyshev_ellipse_rate, lanczos_bounds, ritz_margins


def _spd(n, kappa, seed):
    r = np.random.default_rng(seed)
    q, _ = np.linalg.qr(r.standard_normal((n, n)))
    lam = np.exp(r.uniform(0.0, math.log(kappa), n)); lam[0], lam[-1] = 1.0, kappa
    return (q * lam) @ q.T, lam


@pytest.mark.parametrize("its", [6, 12, 40])
def test_lanczos_interval_and_its_residual_bounds(its):
    """the extreme Ritz values lie inside the spectrum and an eigenvalue lies within the reported residual bound of each; with enough steps the
    bounds collapse and the interval is the spectrum's"""
    A, lam = _spd(120, 40.0, 3)
    dinv = 1.0 / np.diag(A)
    ev = np.linalg.eigvals(dinv[:, None] * A).real
    At, dt = torch.as_tensor(A), torch.as_tensor(dinv)
    b = torch.as_tensor(np.random.default_rng(5).standard_normal((2, 120)))             # two rows: the bounds are the worst over the rows
    lo, hi, elo, ehi = lanczos_bounds(lambda v: v @ At, lambda r: r * dt, b, its=its, errors=True)
    assert ev.min() * (1 - 1e-10) <= lo <= hi <= ev.max() * (1 + 1e-10)
    assert np.abs(ev - lo).min() <= elo * (1 + 1e-8) + 1e-12 and np.abs(ev - hi).min() <= ehi * (1 + 1e-8) + 1e-12
    assert (lo, hi) == lanczos_bounds(lambda v: v @ At, lambda r: r * dt, b, its=its)   # (the same interval without the bounds)
    if its == 40:
        assert abs(lo - ev.min()) < 2e-3 * ev.min() and abs(hi - ev.max()) < 1e-6 * ev.max() and ehi < 1e-3      # (kappa = 40: the lower end is the slow one)


def test_margins_follow_the_estimates_quality():
    assert ritz_margins(1.0, 2.0, 1e-4, 1e-6) == (0.99, 1.01)                           # well-known ends: the 1 % floor
    lo, hi = ritz_margins(1.0, 2.0, 0.03, 0.02)
    assert lo == pytest.approx(0.97) and hi == pytest.approx(1.02)                      # uncertain ends: as wide as the uncertainty
    lo, hi = ritz_margins(1.0, 2.0, 1e-4, 1e-6, widen=2.0)                              # a re-estimate after a missed check: round 5's 10 % / 5 % per unit
    assert lo == pytest.approx(0.90) and hi == pytest.approx(1.05)
    assert ritz_margins(1.0, 2.0, 0.9, 0.0)[0] == pytest.approx(0.6)                    # never below 60 % of the lower end


def test_chebyshev_coefficients_and_the_rate_they_promise():
    """the coefficients of the real interval (ChebyshevMass) are the ellipse formula's with c^2 = delta^2; on a diagonal system with its spectrum in the
    interval the recurrence p = z + beta p, x += alpha p reaches 2 sigma^n; an interval that misses the lower end by 2 % still converges (graceful)"""
    lmin, lmax = 0.35, 1.2
    ch = ChebyshevMass(None, None, lmin, lmax, rtol=1e-14, margin=(1.0, 1.0))
    d, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    ref = chebyshev_ellipse_coefs(d, de * de, ch.steps)
    assert np.allclose(np.array(ch.coef), np.array(ref), rtol=1e-13, atol=1e-15)
    sg = (math.sqrt(lmax / lmin) - 1) / (math.sqrt(lmax / lmin) + 1)
    assert chebyshev_ellipse_rate(d, de, 0.0) == pytest.approx(sg, rel=1e-12)
    assert ch.steps == math.ceil(math.log(2.0 / 1e-14) / math.log(1.0 / sg))
    for lam_lo, slack in ((lmin, 1.0), (0.98 * lmin, 40.0)):
        lam = np.linspace(lam_lo, lmax, 400)
        b = np.random.default_rng(1).standard_normal(400)
        x = np.zeros(400); p = np.zeros(400)
        for al, be in ch.coef:
            z = b - lam * x
            p = z + be * p
            x = x + al * p
        res = np.linalg.norm(b - lam * x) / np.linalg.norm(b)
        assert res <= slack * 2.0 * sg ** ch.steps * 1.5, (lam_lo, res)


def test_ellipse_with_imaginary_foci_converges_on_a_skew_spectrum():
    """1 +- i sigma (the upwinded lumped 0-form mass of diagnose_q under its diagonal): the recurrence stays real with c^2 < 0 and contracts at the
    ellipse's rate"""
    n, sig = 60, 0.27
    r = np.random.default_rng(2)
    K = r.standard_normal((n, n)); K = K - K.T
    K *= sig / np.abs(np.linalg.eigvals(K).imag).max()
    A = np.eye(n) + K
    a_re, a_im = 0.01, 1.05 * sig
    rate = chebyshev_ellipse_rate(1.0, a_re, a_im)
    steps = int(math.ceil(math.log(0.5e-13) / math.log(rate))) + 1
    coef = chebyshev_ellipse_coefs(1.0, a_re * a_re - a_im * a_im, steps)
    b = r.standard_normal(n); x = np.zeros(n); p = np.zeros(n)
    for al, be in coef:
        z = b - A @ x
        p = z + be * p
        x = x + al * p
    assert np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 1e-12
    ev, ev0 = arnoldi_ritz(lambda v: v @ torch.as_tensor(A).T, n, 40, torch.device("cpu"), earlier=25)
    assert abs(ev.real.min() - 1.0) < 1e-6 and abs(np.abs(ev.imag).max() - sig) < 1e-3 * sig and len(ev0) == 25
