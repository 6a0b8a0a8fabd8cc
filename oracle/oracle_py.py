"""ctypes front end of the CPU oracle (oracle/liboracle.so) + the numpy/scipy Poisson restatement.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleVertex(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("geometric_normal", C.c_double * 3),
                ("frame_x", C.c_double * 3), ("frame_y", C.c_double * 3), ("frame_n", C.c_double * 3),
                ("st", C.c_double * 2), ("uv", C.c_double * 2),
                ("uv_screen_size", C.c_double), ("mean_curvature", C.c_double), ("ray_radius", C.c_double),
                ("shape_id", C.c_int32), ("primitive_id", C.c_int32), ("material_id", C.c_int32), ("gid", C.c_int32),
                ("t", C.c_double)]


class OracleSampleRecord(C.Structure):
    _fields_ = [("radiance", C.c_double * 3), ("contrib", C.c_double * 3),
                ("contribX0", C.c_double * 3), ("contribX1", C.c_double * 3),
                ("contribY0", C.c_double * 3), ("contribY1", C.c_double * 3),
                ("prob", C.c_double), ("wX0", C.c_double), ("wY0", C.c_double), ("wX1", C.c_double), ("wY1", C.c_double),
                ("bounces", C.c_int32), ("primary_miss", C.c_int32), ("valid0", C.c_int32 * 4), ("rng_draws", C.c_int32)]


class OracleStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("bounces", C.c_uint64),
                ("primary_misses", C.c_uint64), ("x0_valid_initial", C.c_uint64), ("nonfinite_samples", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("seconds", C.c_double)]


def build(force=False):
    """Compile liboracle.so with the host compiler if it is missing (seconds). GDPT_ORACLE_SO names another build of the
    same source instead (bench.py's cpu_baseline leg times one compiled with -O3 -march=native on the box it runs on)."""
    alt = os.environ.get("GDPT_ORACLE_SO")
    if alt and os.path.exists(alt):
        return alt
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_int]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_intersection_epsilon.restype = C.c_double
        L.oracle_intersection_epsilon.argtypes = [C.c_void_p]
        L.oracle_pcg_init.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oracle_pcg_next.restype = C.c_uint32
        L.oracle_pcg_next.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.oracle_pcg_next_double.restype = C.c_double
        L.oracle_pcg_next_double.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.oracle_sample_primary.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, dp]
        L.oracle_filter_sample.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.oracle_intersect.restype = C.c_int
        L.oracle_intersect.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double, dp, C.POINTER(OracleVertex)]
        L.oracle_shading_info_tri.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_void_p, dp, dp, C.POINTER(OracleVertex), dp]
        L.oracle_bsdf_pdf.restype = C.c_double
        L.oracle_bsdf_pdf.argtypes = [C.c_void_p, C.c_void_p, dp, dp, C.POINTER(OracleVertex)]
        L.oracle_bsdf_sample.restype = C.c_int
        L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_void_p, dp, C.POINTER(OracleVertex), dp, C.c_double, dp, dp, dp]
        L.oracle_texture_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, C.c_double, dp]
        L.oracle_grad_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(OracleSampleRecord)]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.POINTER(OracleStats)]
        L.oracle_light_table.argtypes = [C.c_void_p, dp, dp]
        L.oracle_sample_point_on_shape.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_double, dp]
        L.oracle_pdf_point_on_shape.restype = C.c_double
        L.oracle_pdf_point_on_shape.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
        L.oracle_occluded.restype = C.c_int
        L.oracle_occluded.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double]
        L.oracle_path_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint64, dp,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_reconnect_render.restype = C.c_int
        L.oracle_reconnect_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.POINTER(OracleStats)]
        L.oracle_path_render.restype = C.c_int
        L.oracle_path_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.POINTER(OracleStats)]
        L.oracle_table2d.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int, dp, dp, dp]
        L.oracle_assemble.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
        L.oracle_poisson_dct.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


class OracleScene:
    """CPU oracle scene built from a GdptSceneDesc pointer (the data at the C-ABI boundary)."""

    def __init__(self, desc_ptr, use_bvh=False):
        self._desc = desc_ptr  # keep the owner alive
        addr = C.cast(desc_ptr, C.c_void_p)
        self.handle = C.c_void_p(lib().oracle_scene_create(addr, int(use_bvh)))
        d = desc_ptr.contents if hasattr(desc_ptr, "contents") else desc_ptr
        self.width, self.height = d.camera.width, d.camera.height

    def close(self):
        if self.handle:
            lib().oracle_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def intersection_epsilon(self):
        return lib().oracle_intersection_epsilon(self.handle)

    def sample_primary(self, sx, sy):
        o, d = (C.c_double * 3)(), (C.c_double * 3)()
        lib().oracle_sample_primary(self.handle, sx, sy, o, d)
        return np.array(o), np.array(d)

    def intersect(self, org, dirv, tnear=0.0, tfar=float("inf"), ray_diff=(0.0, 0.0)):
        v = OracleVertex()
        hit = lib().oracle_intersect(self.handle, _vec(org), _vec(dirv), tnear, tfar, _vec(ray_diff), C.byref(v))
        return v if hit else None

    def shading_info_tri(self, gid, st, gn):
        out = (C.c_double * 13)()
        lib().oracle_shading_info_tri(self.handle, gid, _vec(st), _vec(gn), out)
        return np.array(out)

    def bsdf_eval(self, mat, dir_in, dir_out, vertex):
        f = (C.c_double * 3)()
        lib().oracle_bsdf_eval(self.handle, C.addressof(mat), _vec(dir_in), _vec(dir_out), C.byref(vertex), f)
        return np.array(f)

    def bsdf_pdf(self, mat, dir_in, dir_out, vertex):
        return lib().oracle_bsdf_pdf(self.handle, C.addressof(mat), _vec(dir_in), _vec(dir_out), C.byref(vertex))

    def bsdf_sample(self, mat, dir_in, vertex, rnd_uv, rnd_w):
        d = (C.c_double * 3)()
        eta, rough = C.c_double(), C.c_double()
        ok = lib().oracle_bsdf_sample(self.handle, C.addressof(mat), _vec(dir_in), C.byref(vertex), _vec(rnd_uv), rnd_w,
                                      d, C.byref(eta), C.byref(rough))
        return (np.array(d), eta.value, rough.value) if ok else None

    def texture_eval(self, tex, uv, footprint, channels=3):
        out = (C.c_double * 3)()
        lib().oracle_texture_eval(self.handle, C.addressof(tex), channels, _vec(uv), footprint, out)
        return np.array(out)

    def grad_sample(self, x, y, state, inc):
        st = C.c_uint64(state)
        rec = OracleSampleRecord()
        lib().oracle_grad_sample(self.handle, x, y, C.byref(st), inc, C.byref(rec))
        return rec, st.value

    def render(self, spp, rng_scheme, rows=(0, 0), threads=0):
        """Returns (dict of five HxWx3 float64 buffers, OracleStats)."""
        shape = (self.height, self.width, 3)
        bufs = {k: np.zeros(shape, dtype=np.float64) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        st = OracleStats()
        rc = lib().oracle_render(self.handle, spp, rng_scheme, rows[0], rows[1], threads,
                                 _dp(bufs["img"]), _dp(bufs["cx0"]), _dp(bufs["cy0"]), _dp(bufs["cx1"]), _dp(bufs["cy1"]),
                                 C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle_render failed")
        return bufs, st


def _path_methods():
    """Integrator::Path additions to OracleScene (kept together; see oracle.h)."""
    def light_table(self, num_lights):
        pmf, cdf = np.zeros(num_lights), np.zeros(num_lights + 1)
        lib().oracle_light_table(self.handle, _dp(pmf), _dp(cdf))
        return pmf, cdf

    def sample_point_on_shape(self, shape_id, ref_point, uv, w):
        out = np.zeros(6)
        lib().oracle_sample_point_on_shape(self.handle, int(shape_id), _vec(ref_point), _vec(uv), float(w), _dp(out))
        return out[:3].copy(), out[3:].copy()

    def pdf_point_on_shape(self, shape_id, point, normal, ref_point):
        return lib().oracle_pdf_point_on_shape(self.handle, int(shape_id), _vec(point), _vec(normal), _vec(ref_point))

    def occluded(self, org, dirv, tnear, tfar):
        return bool(lib().oracle_occluded(self.handle, _vec(org), _vec(dirv), float(tnear), float(tfar)))

    def path_sample(self, x, y, state, inc):
        st = C.c_uint64(state)
        rad = np.zeros(3)
        b, sh = C.c_int32(), C.c_int32()
        lib().oracle_path_sample(self.handle, int(x), int(y), C.byref(st), C.c_uint64(inc), _dp(rad), C.byref(b), C.byref(sh))
        return rad, st.value, b.value, sh.value

    def path_render(self, spp, rng_scheme, rows=(0, 0), threads=0):
        """path_render (src/render.cpp:74-117): HxWx3 image + OracleStats."""
        img = np.zeros((self.height, self.width, 3))
        st = OracleStats()
        rc = lib().oracle_path_render(self.handle, int(spp), int(rng_scheme), int(rows[0]), int(rows[1]), int(threads), _dp(img), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"oracle_path_render failed ({rc}): scenes with an environment map / without lights are not restated")
        return img, st

    def reconnect_render(self, spp, rows=(0, 0), threads=0):
        """GDPT_SHIFT_RECONNECT (include/gdpt.h), sample-stream RNG: five HxWx3 buffers + OracleStats."""
        bufs = {k: np.zeros((self.height, self.width, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        st = OracleStats()
        rc = lib().oracle_reconnect_render(self.handle, int(spp), int(rows[0]), int(rows[1]), int(threads),
                                           *[_dp(bufs[k]) for k in ("img", "cx0", "cy0", "cx1", "cy1")], C.byref(st))
        if rc != 0:
            raise RuntimeError(f"oracle_reconnect_render failed ({rc})")
        return bufs, st

    for f in (light_table, sample_point_on_shape, pdf_point_on_shape, occluded, path_sample, path_render, reconnect_render):
        setattr(OracleScene, f.__name__, f)


_path_methods()


def table2d(f, rnd):
    """TableDist2D (src/table_dist.cpp:40-150) over the HxW array f: returns (uv samples for rnd (Nx2), their pdf, total)."""
    f = np.ascontiguousarray(f, dtype=np.float64)
    rnd = np.ascontiguousarray(rnd, dtype=np.float64).reshape(-1, 2)
    uv, pdf, total = np.zeros_like(rnd), np.zeros(len(rnd)), np.zeros(1)
    lib().oracle_table2d(_dp(f), f.shape[1], f.shape[0], _dp(rnd), len(rnd), _dp(uv), _dp(pdf), _dp(total))
    return uv, pdf, float(total[0])


def pcg_init(stream):
    s, i = C.c_uint64(), C.c_uint64()
    lib().oracle_pcg_init(stream, C.byref(s), C.byref(i))
    return s.value, i.value


def pcg_next(state, inc):
    s = C.c_uint64(state)
    r = lib().oracle_pcg_next(C.byref(s), inc)
    return r, s.value


def pcg_next_double(state, inc):
    s = C.c_uint64(state)
    r = lib().oracle_pcg_next_double(C.byref(s), inc)
    return r, s.value


def filter_sample(filter_type, param, u0, u1):
    out = (C.c_double * 2)()
    lib().oracle_filter_sample(filter_type, param, u0, u1, out)
    return np.array(out)


def assemble(bufs):
    """src/render.cpp:340-350 on HxWx3 arrays -> (c, cx, cy)."""
    h, w, _ = bufs["img"].shape
    c, cx, cy = (np.empty((h, w, 3)) for _ in range(3))
    a = [np.ascontiguousarray(bufs[k], dtype=np.float64) for k in ("img", "cx0", "cy0", "cx1", "cy1")]
    lib().oracle_assemble(w, h, *[_dp(x) for x in a], _dp(c), _dp(cx), _dp(cy))
    return c, cx, cy


def poisson_dct_c(c, gx, gy, alpha):
    """fourierSolve with the oracle's naive DCT-I (C++). Small sizes only (O(N^1.5))."""
    h, w, _ = c.shape
    out = np.empty((h, w, 3))
    c, gx, gy = (np.ascontiguousarray(a, dtype=np.float64) for a in (c, gx, gy))
    lib().oracle_poisson_dct(w, h, _dp(c), _dp(gx), _dp(gy), alpha, _dp(out))
    return out


def fourier_solve(c, gx, gy, alpha=0.04, float_lambda=True):
    """numpy/scipy restatement of fourierSolve (src/render.cpp:172-254).

    scipy.fft.dctn(type=1) is FFTW's REDFT00 (same unnormalised definition). `float_lambda` keeps the
    reference's fp32 rounding of the Laplacian eigenvalue (src/render.cpp:233).
    Arrays are HxWx3 float64 (row-major Image3 layout).
    """
    from scipy.fft import dctn
    c = np.asarray(c, dtype=np.float64)
    gx = np.asarray(gx, dtype=np.float64)
    gy = np.asarray(gy, dtype=np.float64)
    h, w, _ = c.shape
    xs, ys = np.arange(w), np.arange(h)
    lap_x = 2.0 * np.cos(np.pi * xs / (w - 1))
    lap_y = -4.0 + 2.0 * np.cos(np.pi * ys / (h - 1))
    lam = lap_y[:, None] + lap_x[None, :]
    if float_lambda:
        lam = lam.astype(np.float32).astype(np.float64)
    wx = np.where((xs > 0) & (xs < w - 1), 2.0, 1.0)
    wy = np.where((ys > 0) & (ys < h - 1), 2.0, 1.0)
    out = np.empty_like(c)
    for ch in range(3):
        u, px, py = c[:, :, ch], gx[:, :, ch], gy[:, :, ch]
        dc = float(np.sum(wy[:, None] * wx[None, :] * u))
        hh = alpha * u
        dx = np.empty_like(u)
        dx[:, 1:w - 1] = px[:, 2:w] - px[:, 1:w - 1]
        dx[:, 0] = -2.0 * px[:, 0]
        dx[:, w - 1] = -2.0 * px[:, w - 1]
        dy = np.empty_like(u)
        dy[1:h - 1, :] = py[2:h, :] - py[1:h - 1, :]
        dy[0, :] = -2.0 * py[0, :]
        dy[h - 1, :] = -2.0 * py[h - 1, :]
        hh = (hh - dx) - dy
        H = dctn(hh, type=1)
        F = H / (alpha - lam)
        F[0, 0] = dc
        f = dctn(F, type=1) / (4.0 * (w - 1) * (h - 1))
        out[:, :, ch] = f
    return out
