"""gdpt_amd — MI355X-native drop-in for LaJolla's Integrator::GradPath hot path.

Thin Python mirror of the reference's operator-level entry points over the C ABI in include/gdpt.h
(libgdpt.so: host ingest in C++, kernels in HIP for gfx950). Names follow the reference:

    parse_scene(path)                     src/parsers/parse_scene.h:9
    Scene(desc) / render(...)             gradient_path_render tile loop, src/render.cpp:257-333
    fourierSolve(w, h, c, gx, gy, alpha)  src/render.cpp:172-254
    gradient_path_render(scene, ...)      src/render.cpp:257-370
    imwrite(filename, image)              src/image.cpp:135-173

There is no CPU fallback: every compute call goes to the HIP library and raises if it is missing
or no GPU is visible. (The directory name contains hyphens; import it through the top-level
`gdpt_amd.py` shim.)
"""
import ctypes as C
import os

import numpy as np

from ._ctypes_defs import *  # noqa: F401,F403
from . import _ctypes_defs as defs

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgdpt.so")
_LIB = None


class GdptError(RuntimeError):
    """Raised for a non-zero status from the C ABI (the reference throws fl_exception, src/flexception.h)."""


def library_path():
    return LIB_PATH


def lib():
    """Loads libgdpt.so; fails loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GdptError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                            f"(make -C gradient-based-path-tracing_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.gdpt_last_error.restype = C.c_char_p
        L.gdpt_build_arch.restype = C.c_char_p
        L.gdpt_parse_scene.argtypes = [C.c_char_p, C.POINTER(C.POINTER(defs.GdptSceneDesc))]
        L.gdpt_parse_scene_film.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.POINTER(defs.GdptSceneDesc))]
        L.gdpt_free_scene_desc.argtypes = [C.POINTER(defs.GdptSceneDesc)]
        L.gdpt_scene_upload.argtypes = [C.POINTER(defs.GdptSceneDesc), C.c_int, C.POINTER(vp)]
        L.gdpt_scene_free.argtypes = [vp]
        L.gdpt_scene_info.argtypes = [vp] + [C.POINTER(C.c_int32)] * 4
        L.gdpt_render.argtypes = [vp, C.POINTER(defs.GdptRenderParams), dp, dp, dp, dp, dp, C.POINTER(defs.GdptRenderStats)]
        L.gdpt_render_device.argtypes = [vp, C.POINTER(defs.GdptRenderParams), vp, vp, vp, vp, vp, vp, C.POINTER(defs.GdptRenderStats)]
        L.gdpt_path_render.argtypes = [vp, C.POINTER(defs.GdptRenderParams), dp, C.POINTER(defs.GdptRenderStats)]
        L.gdpt_path_render_device.argtypes = [vp, C.POINTER(defs.GdptRenderParams), vp, vp, C.POINTER(defs.GdptRenderStats)]
        L.gdpt_assemble_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gdpt_poisson_solve.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, dp]
        L.gdpt_poisson_solve_ex.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, dp, C.c_int, C.c_double, C.c_int,
                                            C.POINTER(defs.GdptPoissonStats)]
        L.gdpt_poisson_solve_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, C.c_int, C.c_double, C.c_int,
                                                vp, C.POINTER(defs.GdptPoissonStats)]
        L.gdpt_assemble_solve_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, C.c_int, C.c_double, C.c_int,
                                                 vp, C.POINTER(defs.GdptPoissonStats)]
        L.gdpt_gradient_path_render.argtypes = [vp, C.POINTER(defs.GdptRenderParams), C.c_double, dp, dp, dp, dp, dp, dp,
                                                C.POINTER(defs.GdptRenderStats), C.POINTER(defs.GdptPoissonStats)]
        L.gdpt_imwrite.argtypes = [C.c_char_p, C.c_int, C.c_int, dp]
        L.gdpt_imread.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(dp)]
        L.gdpt_image_free.argtypes = [dp]
        L.gdpt_bvh_check.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32)]
        L.gdpt_sbvh_check.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int32)]
        L.gdpt_assemble_rows_device.argtypes = [C.c_int] * 4 + [vp] * 9
        L.gdpt_band_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gdpt_band_rows_weighted.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gdpt_tile_row_costs.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.gdpt_poisson_forget_stream.argtypes = [vp]
        L.gdpt_band_rows_from_row_costs.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gdpt_multi_rebalance.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
        L.gdpt_multi_create.argtypes = [C.POINTER(defs.GdptSceneDesc), C.POINTER(defs.GdptMultiConfig), C.POINTER(vp)]
        L.gdpt_multi_free.argtypes = [vp]
        L.gdpt_multi_free.restype = None
        L.gdpt_multi_gradient_path_render.argtypes = [vp, C.POINTER(defs.GdptRenderParams), C.c_double, dp, dp, dp, dp, dp, dp,
                                                      C.POINTER(defs.GdptRenderStats), C.POINTER(defs.GdptMultiStats)]
        L.gdpt_debug_knob_set.argtypes = [C.c_char_p, C.c_double]
        L.gdpt_debug_knobs_reset.restype = None
        L.gdpt_debug_chunk_plan.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(C.c_int32), C.c_int]
        L.gdpt_debug_get_stamps.argtypes = [C.POINTER(C.c_double)]
        L.gdpt_debug_get_stamps.restype = None
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise GdptError(lib().gdpt_last_error().decode("utf-8", "replace"))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class SceneDesc:
    """Host-side flattened scene (owner of a GdptSceneDesc*), the result of parse_scene()."""

    def __init__(self, ptr):
        self.ptr = ptr

    @property
    def desc(self):
        return self.ptr.contents

    @property
    def width(self):
        return self.ptr.contents.camera.width

    @property
    def height(self):
        return self.ptr.contents.camera.height

    def close(self):
        if self.ptr:
            lib().gdpt_free_scene_desc(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_scene(filename, film=(0, 0)):
    """Mitsuba-0.x XML subset -> SceneDesc (reference: parse_scene, src/parsers/parse_scene.cpp:1615-1630).
    `film` = (width, height) replaces the <film> extent of the file (0 keeps it)."""
    p = C.POINTER(defs.GdptSceneDesc)()
    _check(lib().gdpt_parse_scene_film(os.fsencode(filename), int(film[0]), int(film[1]), C.byref(p)))
    return SceneDesc(p)


def _params(spp, rng_scheme, rows, max_depth_override=0, shift=0, plan_rows=0):
    p = defs.GdptRenderParams()
    p.shift_mode = int(shift)
    p.plan_rows = int(plan_rows)
    p.spp, p.rng_scheme = int(spp), int(rng_scheme)
    p.row_begin, p.row_end = int(rows[0]), int(rows[1])
    p.max_depth_override = int(max_depth_override)
    return p


class Scene:
    """Device-resident scene: own BVH2 + fp32 traversal records + fp64 shading tables in HBM
    (replaces Scene::Scene's Embree build, src/scene.cpp:4-53)."""

    def __init__(self, scene_desc, device=0):
        self.desc = scene_desc
        self.width, self.height = scene_desc.width, scene_desc.height
        h = C.c_void_p()
        _check(lib().gdpt_scene_upload(scene_desc.ptr, int(device), C.byref(h)))
        self.handle = h

    def info(self):
        v = [C.c_int32() for _ in range(4)]
        _check(lib().gdpt_scene_info(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("num_nodes", "num_tris", "num_spheres", "bvh_depth"), [x.value for x in v]))

    def render(self, spp=0, rng_scheme=defs.RNG_SAMPLE, rows=(0, 0), shift=defs.SHIFT_REFERENCE, plan_rows=0):
        """Five-buffer render to host arrays (HxWx3 float64). Returns (buffers, GdptRenderStats).
        `shift`: SHIFT_REFERENCE (the reference's offsets) or SHIFT_RECONNECT (include/gdpt.h).
        `plan_rows`: GdptRenderParams.plan_rows (0 = work items cut for the whole film)."""
        shape = (self.height, self.width, 3)
        bufs = {k: np.zeros(shape, dtype=np.float64) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        st = defs.GdptRenderStats()
        p = _params(spp, rng_scheme, rows, shift=shift, plan_rows=plan_rows)
        _check(lib().gdpt_render(self.handle, C.byref(p), _dp(bufs["img"]), _dp(bufs["cx0"]), _dp(bufs["cy0"]),
                                 _dp(bufs["cx1"]), _dp(bufs["cy1"]), C.byref(st)))
        return bufs, st

    def render_device(self, ptrs, spp=0, rng_scheme=defs.RNG_SAMPLE, rows=(0, 0), stream=None, want_stats=False,
                      shift=defs.SHIFT_REFERENCE, plan_rows=0):
        """Five-buffer render into device memory; `ptrs` = 5 device addresses (e.g. torch tensor.data_ptr())."""
        st = defs.GdptRenderStats() if want_stats else None
        p = _params(spp, rng_scheme, rows, shift=shift, plan_rows=plan_rows)
        _check(lib().gdpt_render_device(self.handle, C.byref(p), *[C.c_void_p(int(x)) for x in ptrs],
                                        C.c_void_p(int(stream) if stream else 0), C.byref(st) if st is not None else None))
        return st

    def path_render(self, spp=0, rng_scheme=defs.RNG_SAMPLE, rows=(0, 0), want_counts=False):
        """Integrator::Path (path_render, src/render.cpp:74-117): HxWx3 float64 image + GdptRenderStats."""
        img = np.zeros((self.height, self.width, 3), dtype=np.float64)
        st = defs.GdptRenderStats()
        if want_counts:
            st.nodes_visited = 2 ** 64 - 1
        p = _params(spp, rng_scheme, rows)
        _check(lib().gdpt_path_render(self.handle, C.byref(p), _dp(img), C.byref(st)))
        return img, st

    def path_render_device(self, ptr, spp=0, rng_scheme=defs.RNG_SAMPLE, rows=(0, 0), stream=None):
        p = _params(spp, rng_scheme, rows)
        _check(lib().gdpt_path_render_device(self.handle, C.byref(p), C.c_void_p(int(ptr)), C.c_void_p(int(stream) if stream else 0), None))

    def gradient_path_render(self, spp=0, rng_scheme=defs.RNG_SAMPLE, alpha=0.04, return_buffers=False,
                             shift=defs.SHIFT_REFERENCE, plan_rows=0):
        """Whole Integrator::GradPath: render + assembly + screened-Poisson solve (src/render.cpp:257-370)."""
        shape = (self.height, self.width, 3)
        out = np.zeros(shape, dtype=np.float64)
        bufs = {k: np.zeros(shape, dtype=np.float64) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        rs, ps = defs.GdptRenderStats(), defs.GdptPoissonStats()
        p = _params(spp, rng_scheme, (0, 0), shift=shift, plan_rows=plan_rows)
        _check(lib().gdpt_gradient_path_render(self.handle, C.byref(p), float(alpha), _dp(out),
                                               _dp(bufs["img"]), _dp(bufs["cx0"]), _dp(bufs["cy0"]), _dp(bufs["cx1"]), _dp(bufs["cy1"]),
                                               C.byref(rs), C.byref(ps)))
        return (out, bufs, rs, ps) if return_buffers else out

    def tile_row_costs(self, spp=1):
        """Rays of a pilot render of every 16-pixel tile row (gdpt_tile_row_costs): what sharding.bands_weighted balances by."""
        T = (self.height + 15) // 16
        buf = (C.c_double * T)()
        _check(lib().gdpt_tile_row_costs(self.handle, int(spp), buf, T))
        return [float(x) for x in buf]

    def close(self):
        if getattr(self, "handle", None):
            lib().gdpt_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fourierSolve(width, height, imgData, imgGradX, imgGradY, dataCost=0.04, solver=defs.SOLVER_DEFAULT, tol=0.0,
                 max_iters=0, return_stats=False):
    """Screened-Poisson reconstruction on the GPU; arguments as the reference's fourierSolve
    (src/render.cpp:172-175). Inputs HxWx3 (or flat W*H*3) float64; returns HxWx3."""
    a = [np.ascontiguousarray(x, dtype=np.float64).reshape(height, width, 3) for x in (imgData, imgGradX, imgGradY)]
    out = np.empty((height, width, 3), dtype=np.float64)
    st = defs.GdptPoissonStats()
    _check(lib().gdpt_poisson_solve_ex(int(width), int(height), _dp(a[0]), _dp(a[1]), _dp(a[2]), float(dataCost), _dp(out),
                                       int(solver), float(tol), int(max_iters), C.byref(st)))
    return (out, st) if return_stats else out


def assemble_device(width, height, src_ptrs, dst_ptrs, stream=None, rows=(0, 0)):
    """c, cx, cy from the five accumulation buffers (src/render.cpp:340-350); `rows` = the band this rank owns."""
    _check(lib().gdpt_assemble_rows_device(int(width), int(height), int(rows[0]), int(rows[1]),
                                           *[C.c_void_p(int(x)) for x in src_ptrs],
                                           *[C.c_void_p(int(x)) for x in dst_ptrs], C.c_void_p(int(stream) if stream else 0)))


def band_rows(height, num_bands, band):
    """Rows [r0, r1) of one band of the sharded tile loop, as the C host computes them (gdpt_band_rows)."""
    r0, r1 = C.c_int32(), C.c_int32()
    _check(lib().gdpt_band_rows(int(height), int(num_bands), int(band), C.byref(r0), C.byref(r1)))
    return r0.value, r1.value


def band_rows_weighted(height, num_bands, band, costs):
    """Rows [r0, r1) of one band of a cost-balanced sharding, as the C host computes them (gdpt_band_rows_weighted)."""
    r0, r1 = C.c_int32(), C.c_int32()
    arr = (C.c_double * len(costs))(*[float(c) for c in costs])
    _check(lib().gdpt_band_rows_weighted(int(height), int(num_bands), int(band), arr, len(costs), C.byref(r0), C.byref(r1)))
    return r0.value, r1.value


def band_rows_from_row_costs(height, num_bands, band, row_costs, granularity=1):
    """Rows [r0, r1) of one band of a sharding balanced on per-row costs, as the C host computes them (gdpt_band_rows_from_row_costs;
    mirror: sharding.bands_from_row_costs)."""
    r0, r1 = C.c_int32(), C.c_int32()
    arr = (C.c_double * len(row_costs))(*[float(c) for c in row_costs])
    if len(row_costs) != int(height):
        raise ValueError("one cost per pixel row expected")
    _check(lib().gdpt_band_rows_from_row_costs(int(height), int(num_bands), int(band), arr, int(granularity), C.byref(r0), C.byref(r1)))
    return r0.value, r1.value


class MultiScene:
    """The scene uploaded to several devices of one node, tile loop sharded into row bands (include/gdpt.h,
    gdpt_multi_*): replaces the reference's thread pool over tiles (src/parallel.cpp:183-256)."""

    def __init__(self, scene_desc, devices, exchange=defs.EXCHANGE_RCCL, balance=False):
        self.desc = scene_desc
        self.width, self.height = scene_desc.width, scene_desc.height
        cfg = defs.GdptMultiConfig()
        cfg.num_devices, cfg.exchange, cfg.balance = len(devices), int(exchange), int(bool(balance))
        for i, d in enumerate(devices):
            cfg.devices[i] = int(d)
        h = C.c_void_p()
        _check(lib().gdpt_multi_create(scene_desc.ptr, C.byref(cfg), C.byref(h)))
        self.handle = h

    def gradient_path_render(self, spp=0, rng_scheme=defs.RNG_SAMPLE, alpha=0.04, return_buffers=False, shift=defs.SHIFT_REFERENCE):
        shape = (self.height, self.width, 3)
        out = np.zeros(shape, dtype=np.float64)
        bufs = {k: np.zeros(shape, dtype=np.float64) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        rs, ms = defs.GdptRenderStats(), defs.GdptMultiStats()
        p = _params(spp, rng_scheme, (0, 0), shift=shift)
        _check(lib().gdpt_multi_gradient_path_render(self.handle, C.byref(p), float(alpha), _dp(out),
                                                     _dp(bufs["img"]), _dp(bufs["cx0"]), _dp(bufs["cy0"]), _dp(bufs["cx1"]), _dp(bufs["cy1"]),
                                                     C.byref(rs), C.byref(ms)))
        return (out, bufs, rs, ms) if return_buffers else out

    def rebalance(self, band_ms, granularity=1):
        """Feedback between frames (gdpt_multi_rebalance): `band_ms` = GdptMultiStats.render_ms of the last call; the bands are cut
        again on the corrected cost model, at multiples of `granularity` rows."""
        arr = (C.c_double * len(band_ms))(*[float(t) for t in band_ms])
        _check(lib().gdpt_multi_rebalance(self.handle, arr, int(granularity)))

    def close(self):
        if getattr(self, "handle", None):
            lib().gdpt_multi_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def poisson_forget_stream(stream):
    """Drops the solver scratch kept for `stream` on the current device (gdpt_poisson_forget_stream): call before destroying a
    stream the solver has run on."""
    _check(lib().gdpt_poisson_forget_stream(C.c_void_p(int(stream) if stream else 0)))


def poisson_solve_device(width, height, c_ptr, gx_ptr, gy_ptr, out_ptr, alpha=0.04, solver=defs.SOLVER_DEFAULT, tol=0.0,
                         max_iters=0, stream=None, want_stats=False):
    st = defs.GdptPoissonStats() if want_stats else None
    _check(lib().gdpt_poisson_solve_device(int(width), int(height), C.c_void_p(int(c_ptr)), C.c_void_p(int(gx_ptr)),
                                           C.c_void_p(int(gy_ptr)), float(alpha), C.c_void_p(int(out_ptr)), int(solver),
                                           float(tol), int(max_iters), C.c_void_p(int(stream) if stream else 0),
                                           C.byref(st) if st is not None else None))
    return st


def assemble_solve_device(width, height, src_ptrs, dst_ptrs, out_ptr, alpha=0.04, solver=defs.SOLVER_DEFAULT, tol=0.0, max_iters=0,
                          stream=None, want_stats=False):
    """assemble_device over the whole film + poisson_solve_device on its outputs as one call (gdpt_assemble_solve_device): the
    assembly and the solver's right-hand side are one pass over the film. `src_ptrs` = img, cx0, cy0, cx1, cy1; `dst_ptrs` = c, cx, cy."""
    st = defs.GdptPoissonStats() if want_stats else None
    _check(lib().gdpt_assemble_solve_device(int(width), int(height), *[C.c_void_p(int(x)) for x in src_ptrs], *[C.c_void_p(int(x)) for x in dst_ptrs],
                                            float(alpha), C.c_void_p(int(out_ptr)), int(solver), float(tol), int(max_iters),
                                            C.c_void_p(int(stream) if stream else 0), C.byref(st) if st is not None else None))
    return st


def imwrite(filename, image):
    """.pfm (fp32) / .exr (fp16) by suffix, like src/image.cpp:135-173."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    h, w, _ = img.shape
    _check(lib().gdpt_imwrite(os.fsencode(filename), w, h, _dp(img)))


def imread(filename, channels=3):
    """imread3 / imread1 of the reference (src/image.cpp:26-133): HxWxC float64 texels."""
    w, h = C.c_int(), C.c_int()
    p = C.POINTER(C.c_double)()
    _check(lib().gdpt_imread(os.fsencode(filename), int(channels), C.byref(w), C.byref(h), C.byref(p)))
    try:
        a = np.ctypeslib.as_array(p, shape=(h.value, w.value, int(channels))).copy()
    finally:
        lib().gdpt_image_free(p)
    return a


def bvh_check(bounds):
    """Builds + verifies the traversal trees over n fp32 boxes (n x 6: min xyz, max xyz); host only.
    Returns the stats dict of gdpt_bvh_check (include/gdpt.h)."""
    b = np.ascontiguousarray(bounds, dtype=np.float32).reshape(-1, 6)
    st = (C.c_int32 * 8)()
    _check(lib().gdpt_bvh_check(b.ctypes.data_as(C.POINTER(C.c_float)), b.shape[0], st))
    return dict(zip(("bvh2_nodes", "bvh2_depth", "wide_nodes", "wide_arity", "wide_stack_need", "leaves", "max_leaf_prims", "bvh8_nodes"), list(st)[:8]))


def sbvh_check(tri_verts, budget=0.3, samples_per_tri=16):
    """Builds the BVH with spatial splits over n fp32 triangles (n x 3 x 3) and verifies it incl. coverage by point sampling;
    host only. Returns the stats dict of gdpt_sbvh_check (include/gdpt.h)."""
    t = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
    st = (C.c_int32 * 8)()
    _check(lib().gdpt_sbvh_check(t.ctypes.data_as(C.POINTER(C.c_float)), t.shape[0], C.c_double(budget), int(samples_per_tri), st))
    d = dict(zip(("bvh2_nodes", "bvh2_depth", "references", "wide_nodes", "wide_stack_need", "leaves"), list(st)[:6]))
    d["sah_nodes"] = st[6] / 1000.0; d["sah_prims"] = st[7] / 1000.0
    return d


def shape_triangles(scene_desc):
    """fp32 vertices of every triangle of a SceneDesc (T x 3 x 3), in shape order."""
    d = scene_desc.desc
    out = []
    for i in range(d.num_shapes):
        sh = d.shapes[i]
        if sh.num_triangles <= 0:
            continue
        pos = np.ctypeslib.as_array(sh.positions, shape=(sh.num_vertices, 3)).astype(np.float32)
        idx = np.ctypeslib.as_array(sh.indices, shape=(sh.num_triangles, 3))
        out.append(pos[idx])
    return np.concatenate(out, axis=0) if out else np.zeros((0, 3, 3), np.float32)


def shape_triangle_bounds(scene_desc):
    """fp32 boxes of every triangle of a SceneDesc, in shape order (what gdpt_scene_upload hands its BVH builder)."""
    d = scene_desc.desc
    out = []
    for i in range(d.num_shapes):
        sh = d.shapes[i]
        if sh.num_triangles <= 0:
            continue
        pos = np.ctypeslib.as_array(sh.positions, shape=(sh.num_vertices, 3)).astype(np.float32)
        idx = np.ctypeslib.as_array(sh.indices, shape=(sh.num_triangles, 3))
        tri = pos[idx]                                   # T x 3 x 3
        out.append(np.concatenate([tri.min(axis=1), tri.max(axis=1)], axis=1))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 6), np.float32)


class debug_knobs:
    """Test instrument (include/gdpt_debug.h): `with debug_knobs(force_eager=1, log2k=0): ...` forces an alternative
    schedule for the calls inside the block and restores the product path afterwards. The library reads no environment
    variable; manual sweep scripts under tests/ that take their settings from GDPT_* variables translate them through
    debug_knobs.from_env() explicitly."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        for k, v in self.knobs.items():
            _check(lib().gdpt_debug_knob_set(k.encode(), float(v)))
        return self

    def __exit__(self, *exc):
        lib().gdpt_debug_knobs_reset()
        return False

    @staticmethod
    def set(**knobs):
        for k, v in knobs.items():
            _check(lib().gdpt_debug_knob_set(k.encode(), float(v)))

    @staticmethod
    def reset():
        lib().gdpt_debug_knobs_reset()

    @staticmethod
    def chunk_plan(spp, film_pixels, resident_lanes=256 * 2 * 256, force_log2k=-1):
        """Sample ranges of a pixel's work items: [begin[c], begin[c+1]) (include/gdpt_debug.h)."""
        buf = (C.c_int32 * 80)()
        n = lib().gdpt_debug_chunk_plan(int(spp), int(force_log2k), int(film_pixels), int(resident_lanes), buf, 80)
        if n < 0:
            raise GdptError("gdpt_debug_chunk_plan: capacity")
        return list(buf)[:n + 1]

    @staticmethod
    def stamps():
        """Per-segment wave cycles of the last render made under debug_knobs(stamps=1) (include/gdpt_debug.h)."""
        v = (C.c_double * 16)()
        lib().gdpt_debug_get_stamps(v)
        d = dict(zip(("loop_head", "trace", "vertex", "consume", "bsdf", "finish", "camera", "wave_steps", "publish", "take", "item"), list(v)))
        d["busy_us"] = (v[13] - v[12]) / 100.0      # first wave started -> first wave found the queue empty
        d["drain_us"] = (v[14] - v[13]) / 100.0     # ... -> last wave ended
        return d

    @staticmethod
    def from_env(environ=None):
        """GDPT_FORCE_EAGER=1 -> force_eager=1 ... for the manual sweep scripts (tests/sweep_*.py, tune_render.py)."""
        environ = os.environ if environ is None else environ
        names = ("force_eager", "log2k", "keep_frac", "search_frac", "blocks_per_cu", "no_lds_scene", "lds_wide",
                 "no_twosided_machine", "presplit", "presplit_floor", "bvh_leaf_max", "bvh_leaf_factor", "sbvh", "sbvh_alpha", "plan_rounds", "plan_shrink", "plan_digits", "bvh_collapse_dp", "stamps", "wavefront", "wf_slots", "wf_sort", "multi_fail_band", "multi_fail_stage", "dct_bk", "dct_bm", "full_material_switch", "no_plain_kernel", "replay_per_step")
        lib().gdpt_debug_knobs_reset()
        for n in names:
            v = environ.get("GDPT_" + n.upper())
            if v is not None:
                _check(lib().gdpt_debug_knob_set(n.encode(), float(v)))


def build_arch():
    return lib().gdpt_build_arch().decode()
