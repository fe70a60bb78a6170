"""Initial conditions of the shallow-water drivers (src/Williamson2.cpp:20-61, src/Galewsky.cpp:24-82): the vectorised torch versions
used by bench.py against a point-by-point scalar restatement of the reference's functions (CPU only)."""
import math

import numpy as np
import torch

from mimsem_amd.sweqn import RAD_SPHERE, galewsky, williamson2


def _u_galewsky(x):
    eps, umax, phi0 = 1.0e-8, 80.0, math.pi / 7.0
    phi1 = math.pi / 2.0 - phi0
    en = math.exp(-4.0 / ((phi1 - phi0) * (phi1 - phi0)))
    phi = math.asin(x[2] / RAD_SPHERE)
    if phi0 + eps < phi < phi1 - eps:
        return (umax / en) * math.exp(1.0 / ((phi - phi0) * (phi - phi1)))
    return 0.0


def _h_galewsky(x):
    ni, phiPrime = 1000, 0.0
    phi, lam = math.asin(x[2] / RAD_SPHERE), math.atan2(x[1], x[0])
    dphi = abs(phi / ni)
    h, grav, omega = 10000.0, 9.80616, 7.292e-5
    sgn = 1 if phi > 0 else -1
    x2 = [x[0], x[1], 0.0]
    for _ in range(ni):
        phiPrime += sgn * dphi
        x2[2] = RAD_SPHERE * math.sin(phiPrime)
        u = _u_galewsky(x2)
        f = 2.0 * omega * math.sin(phiPrime)
        h -= RAD_SPHERE * u * (f + math.tan(phiPrime) * u / RAD_SPHERE) * dphi / grav
    h += 120.0 * math.cos(phi) * math.exp(-1.0 * (lam / (1.0 / 3.0)) ** 2) * math.exp(-1.0 * ((math.pi / 4.0 - phi) / (1.0 / 15.0)) ** 2)
    return h


def _points():
    r = np.random.default_rng(4)
    lat = np.concatenate([r.uniform(-1.5, 1.5, 10), [math.pi / 7.0, math.pi / 2.0 - math.pi / 7.0, 0.7, 0.78, -0.6, 0.0]])
    lon = np.concatenate([r.uniform(-3.1, 3.1, 10), [0.1, -0.2, 0.05, 0.0, 2.0, 1.0]])
    return np.stack([RAD_SPHERE * np.cos(lat) * np.cos(lon), RAD_SPHERE * np.cos(lat) * np.sin(lon), RAD_SPHERE * np.sin(lat)], axis=1)


def test_galewsky_initial_state():
    x = _points()
    (uv, h) = galewsky(torch.as_tensor(x))
    for i, xi in enumerate(x):
        assert abs(float(uv[i, 0]) - _u_galewsky(xi)) <= 1e-12 * 80.0 and float(uv[i, 1]) == 0.0
        assert abs(float(h[i]) - _h_galewsky(xi)) <= 1e-10 * 1e4
    assert float(uv[:, 0].max()) > 60.0                       # the jet core is among the points


def test_williamson2_initial_state():
    x = _points()
    uv, h = williamson2(torch.as_tensor(x), alpha=0.0)
    U0, H0, OMEGA, GRAV = 38.61068276698372, 2998.1154702758267, 7.292e-5, 9.80616
    for i, xi in enumerate(x):
        th = math.asin(xi[2] / RAD_SPHERE)
        assert abs(float(uv[i, 0]) - U0 * math.cos(th)) < 1e-12 * U0 and abs(float(uv[i, 1])) < 1e-12
        b = math.sin(th)
        assert abs(float(h[i]) - (H0 - (RAD_SPHERE * OMEGA * U0 + 0.5 * U0 * U0) * b * b / GRAV)) < 1e-10 * H0
