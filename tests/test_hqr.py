"""The Hessenberg QR eigenvalue routine behind mimsem_ksp_ritz (csrc/hqr_host.hpp, host-only C++) against numpy: random Hessenberg matrices,
spectra like the ones it is used on (real clusters, complex-conjugate pairs off the real axis), a companion matrix, trivial sizes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hqr") / "hqr_cli")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "hqr_cli.cpp"), "-o", exe])
    return exe


def _eig(cli, H):
    n = H.shape[0]
    inp = "%d\n" % n + "\n".join(" ".join("%.17g" % v for v in row) for row in H) + "\n"
    out = subprocess.run([cli], input=inp, capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == 0
    v = np.array(out[1:], dtype=float).reshape(n, 2)
    return v[:, 0] + 1j * v[:, 1]


def _match(a, b, tol):
    a = list(a)
    for z in b:
        k = int(np.argmin([abs(z - w) for w in a]))
        assert abs(z - a[k]) <= tol * max(1.0, abs(z)), (z, a[k])
        a.pop(k)


def test_hessenberg_eigenvalues_match_numpy(cli):
    rng = np.random.default_rng(8)
    for n in (1, 2, 3, 7, 30, 61):
        H = np.triu(rng.standard_normal((n, n)), -1)
        _match(_eig(cli, H), np.linalg.eigvals(H), 1e-9)
    # a real cluster in [0.35, 1.2] (the preconditioned [u|h] operator) and a vertical segment 1 +- 0.27 i (the upwinded mass)
    for n, spec in ((40, rng.uniform(0.35, 1.2, 40)), (40, np.concatenate([0.99 + 1j * rng.uniform(0, 0.27, 20), 0.99 - 1j * rng.uniform(0, 0.27, 20)]))):
        if np.iscomplexobj(spec):
            spec = np.concatenate([spec[:20], spec[:20].conj()])
            D = np.zeros((n, n))
            for k in range(20):
                D[2 * k:2 * k + 2, 2 * k:2 * k + 2] = [[spec[k].real, spec[k].imag], [-spec[k].imag, spec[k].real]]
        else:
            D = np.diag(spec)
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = Q @ D @ Q.T
        import scipy.linalg
        H = scipy.linalg.hessenberg(A)
        ev = _eig(cli, np.triu(H, -1))
        _match(ev, np.linalg.eigvals(A), 1e-10)
        assert abs(ev.real.min() - np.linalg.eigvals(A).real.min()) < 1e-10
    # companion matrix of (x - 1)(x - 2)(x - 3)(x^2 + 1)
    c = np.poly([1, 2, 3, 1j, -1j]).real
    C = np.zeros((5, 5)); C[0, :] = -c[1:]; C[1:, :-1] = np.eye(4)
    _match(_eig(cli, C), [1, 2, 3, 1j, -1j], 1e-9)
