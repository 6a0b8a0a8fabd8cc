"""Screened-Poisson oracle: the numpy/scipy restatement of fourierSolve (src/render.cpp:172-254), the C++ naive
DCT-I restatement, and the linear-system form the GPU's CG solves must agree.

Pinning: FFTW is not buildable here without its generated config (vendored tree needs ./configure), so the
reference's fourierSolve is not executed; scipy.fft.dctn(type=1) is REDFT00 by definition. SURVEY.md Appendix A.2
records the survey's measurement of the reference against that same scipy restatement (2e-16..1.3e-14)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from helpers import rel_l2


def lcg_fields(w, h, seed=12345):
    # SURVEY.md §8(d) micro-benchmark inputs: c ~ U[0,1), gx,gy ~ U[-1/2,1/2) from an LCG
    n = w * h * 3
    state = np.uint64(seed)
    out = np.empty(3 * n)
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for i in range(3 * n):
            state = state * a + c
            out[i] = float(state >> np.uint64(11)) / 9007199254740992.0
    return out[:n].reshape(h, w, 3), out[n:2 * n].reshape(h, w, 3) - 0.5, out[2 * n:].reshape(h, w, 3) - 0.5


def system_matrix(w, h, alpha):
    """W (alpha I - L) with mirror (whole-sample symmetric) boundaries, one channel."""
    def lap1(n):
        L = sp.lil_matrix((n, n))
        for i in range(n):
            L[i, i] = -2.0
            if 0 < i < n - 1:
                L[i, i - 1] = 1.0
                L[i, i + 1] = 1.0
            elif i == 0:
                L[i, 1] = 2.0
            else:
                L[i, n - 2] = 2.0
        return L.tocsr()
    Lx, Ly = lap1(w), lap1(h)
    L = sp.kron(sp.identity(h), Lx) + sp.kron(Ly, sp.identity(w))
    wx = np.where((np.arange(w) > 0) & (np.arange(w) < w - 1), 2.0, 1.0)
    wy = np.where((np.arange(h) > 0) & (np.arange(h) < h - 1), 2.0, 1.0)
    wgt = (wy[:, None] * wx[None, :]).ravel()
    A = sp.diags(wgt) @ (alpha * sp.identity(w * h) - L)
    return A.tocsc(), wgt


@pytest.mark.parametrize("w,h", [(8, 6), (9, 7), (33, 20), (2, 2), (3, 17)])
def test_scipy_restatement_equals_cpp_naive_dct(O, w, h):
    c, gx, gy = lcg_fields(w, h, seed=w * 100 + h)
    a = O.fourier_solve(c, gx, gy, 0.04)
    b = O.poisson_dct_c(c, gx, gy, 0.04)
    assert rel_l2(a, b) < 1e-12


@pytest.mark.parametrize("w,h,alpha", [(9, 7, 0.04), (33, 20, 0.04), (16, 16, 0.4), (12, 5, 4.0)])
def test_linear_system_form_reproduces_dct_solution(O, w, h, alpha):
    """(alpha I - L) f = h plus the DC shift == the DCT solve with the DC override (fp64 lambda)."""
    c, gx, gy = lcg_fields(w, h, seed=7 * w + h)
    ref = O.fourier_solve(c, gx, gy, alpha, float_lambda=False)
    A, wgt = system_matrix(w, h, alpha)
    out = np.empty_like(ref)
    for ch in range(3):
        u, px, py = c[:, :, ch], gx[:, :, ch], gy[:, :, ch]
        dx = np.empty_like(u); dy = np.empty_like(u)
        dx[:, 1:w - 1] = px[:, 2:] - px[:, 1:w - 1]; dx[:, 0] = -2 * px[:, 0]; dx[:, w - 1] = -2 * px[:, w - 1]
        dy[1:h - 1, :] = py[2:, :] - py[1:h - 1, :]; dy[0, :] = -2 * py[0, :]; dy[h - 1, :] = -2 * py[h - 1, :]
        hh = alpha * u - dx - dy
        f = spla.spsolve(A, wgt * hh.ravel())
        shift = (np.sum(wgt * u.ravel()) - np.sum(wgt * hh.ravel()) / alpha) / (4.0 * (w - 1) * (h - 1))
        out[:, :, ch] = f.reshape(h, w) + shift
    assert rel_l2(out, ref) < 1e-11
    # and A is symmetric (that is what the weights are for)
    assert abs(A - A.T).max() < 1e-14


def test_float_lambda_quirk_is_small_and_dc_invariant_holds(O):
    w, h = 64, 48
    c, gx, gy = lcg_fields(w, h)
    a = O.fourier_solve(c, gx, gy, 0.04, float_lambda=True)
    b = O.fourier_solve(c, gx, gy, 0.04, float_lambda=False)
    d = rel_l2(a, b)
    assert 0 < d < 1e-7          # 3-4e-9 at 512x512
    wx = np.where((np.arange(w) > 0) & (np.arange(w) < w - 1), 2.0, 1.0)
    wy = np.where((np.arange(h) > 0) & (np.arange(h) < h - 1), 2.0, 1.0)
    wgt = wy[:, None, None] * wx[None, :, None]
    np.testing.assert_allclose((wgt * a).sum(axis=(0, 1)), (wgt * c).sum(axis=(0, 1)), rtol=1e-12)


def test_assemble_matches_reference_pairing(O):
    # cx = cx0(x,y) + cx1(x-1,y), cy = cy0(x,y) + cy1(x,y-1)  (src/render.cpp:345-349)
    rng = np.random.default_rng(3)
    bufs = {k: rng.standard_normal((5, 7, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
    c, cx, cy = O.assemble(bufs)
    assert np.array_equal(c, bufs["img"])
    want_cx = bufs["cx0"].copy(); want_cx[:, 1:] += bufs["cx1"][:, :-1]
    want_cy = bufs["cy0"].copy(); want_cy[1:, :] += bufs["cy1"][:-1, :]
    assert np.array_equal(cx, want_cx) and np.array_equal(cy, want_cy)
