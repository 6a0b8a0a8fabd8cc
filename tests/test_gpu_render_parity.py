"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same PCG streams.

Tolerance: BASELINE.json north_star asks for <= 1e-4 relative L2 on the output; the five accumulation buffers are
compared at 1e-9 (fp64 on both sides; the kernels may contract a*b+c into FMAs, the oracle never does), hit
records (fp32 traversal) are expected bit-identical, so discrete path decisions agree."""
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT, SCENES, rel_l2, scene_variant, DescBuilder

pytestmark = pytest.mark.gpu
BUFS = ("img", "cx0", "cy0", "cx1", "cy1")
TOL = 1e-9


def check_buffers(got, want, tol=TOL):
    for k in BUFS:
        assert np.isfinite(got[k]).all(), k
        err = rel_l2(got[k], want[k])
        assert err < tol, f"{k}: rel L2 {err}"


@pytest.mark.parametrize("w,h,spp", [(64, 64, 8), (48, 32, 5), (33, 17, 3), (16, 16, 1)])
def test_cbox_sample_stream(G, O, scene_tmp, w, h, spp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=w, height=h)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.render(spp, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr).render(spp, G.RNG_SAMPLE, threads=8)
    check_buffers(got, want)
    assert st.samples == w * h * spp == ost.samples
    assert st.bounces == ost.bounces and st.nonfinite_samples == 0
    again, _ = sc.render(spp, G.RNG_SAMPLE)
    for k in BUFS:      # one writer per pixel, fixed reduction order: run-to-run bit-identical
        assert np.array_equal(got[k], again[k])


def test_cbox_tile_stream_reference_rng_order(G, O, scene_tmp):
    """The reference's own RNG order (one PCG stream per 16x16 tile, src/render.cpp:281-309), one lane per tile."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=48, height=40)    # ragged last tile row
    sd = G.parse_scene(xml)
    got, st = G.Scene(sd).render(3, G.RNG_TILE)
    want, ost = O.OracleScene(sd.ptr).render(3, G.RNG_TILE, threads=4)
    check_buffers(got, want)
    assert st.bounces == ost.bounces


def test_work_item_granularity_does_not_change_the_result(G, scene_tmp, monkeypatch):
    """The persistent kernel cuts a pixel's samples into work items whose partial sums are merged in chunk order;
    1, 2 or 8 items per pixel (and the eager evaluator) must agree to summation-order rounding."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=24)
    sc = G.Scene(G.parse_scene(xml))
    ref, rst = sc.render(16, G.RNG_SAMPLE)
    for env in ({"log2k": 0}, {"log2k": 1}, {"log2k": 3}, {"force_eager": 1}, {"no_lds_scene": 1}):
        with G.debug_knobs(**env):
            got, st = sc.render(16, G.RNG_SAMPLE)
        assert st.bounces == rst.bounces
        for k in BUFS:
            assert rel_l2(got[k], ref[k]) < 1e-12, (env, k)


def test_row_bands_compose_to_the_whole_image(G, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=64)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    whole, _ = sc.render(4, G.RNG_SAMPLE)
    # whole tile rows, and bands cut at any row (the persistent kernels anchor their 16x16 work items at the band's first row; what
    # bench.py's cost-balanced bands rely on)
    for cuts in (((0, 16), (16, 48), (48, 64)), ((0, 13), (13, 37), (37, 38), (38, 64))):
        acc = None
        for rows in cuts:
            part, st = sc.render(4, G.RNG_SAMPLE, rows=rows)
            assert st.samples == 40 * (rows[1] - rows[0]) * 4
            for k in BUFS:
                assert not part[k][:rows[0]].any() and not part[k][rows[1]:].any()     # rows outside the band untouched
            acc = part if acc is None else {k: acc[k] + part[k] for k in BUFS}
        for k in BUFS:
            assert np.array_equal(acc[k], whole[k]), (k, cuts)


@pytest.mark.parametrize("max_depth", [1, 2, 3, 6])
def test_max_depth_variants(G, O, scene_tmp, max_depth):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=32, height=32, max_depth=max_depth)
    sd = G.parse_scene(xml)
    got, st = G.Scene(sd).render(4, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr).render(4, G.RNG_SAMPLE, threads=4)
    check_buffers(got, want)
    assert st.bounces == ost.bounces


def test_sphere_scene_small_pt_compare(G, O, scene_tmp):
    """Sphere primitives (fp64 quadratic on the fp32 ray) incl. 1e5-radius walls and a sphere light."""
    xml = scene_variant(scene_tmp, "cbox/small_pt_compare.xml", width=48, height=32)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    assert sc.info()["num_spheres"] >= 6 and sc.info()["num_tris"] == 0
    got, st = sc.render(6, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr).render(6, G.RNG_SAMPLE, threads=8)
    check_buffers(got, want, 1e-7)      # acos/atan2/sincos differ by an ulp between libm and the device library
    assert st.bounces == ost.bounces


@pytest.mark.parametrize("scene", ["disney_diffuse", "disney_metal", "disney_clearcoat", "disney_sheen", "disney_glass", "disney_bsdf"])
def test_disney_lobes_gradpath(G, O, scene_tmp, scene):
    """BASELINE config 5 geometry (61 600 triangles, checkerboard plane) with the integrator switched to gradpath.
    Lit only by an envmap, which GradPath ignores: the primal image is zero, the gradient buffers are not."""
    xml = scene_variant(scene_tmp, f"disney_bsdf_test/{scene}.xml", width=48, height=36, integrator="gradpath")
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.render(4, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).render(4, G.RNG_SAMPLE, threads=8)
    assert st.nonfinite_samples == ost.nonfinite_samples
    assert st.bounces == ost.bounces
    assert not got["img"].any() and not want["img"].any()
    for k in BUFS[1:]:
        assert np.abs(want[k]).max() > 0
        assert rel_l2(got[k], want[k]) < 1e-7, k


def test_sponza_textures_sphere_light_big_bvh(G, O, scene_tmp):
    """BASELINE config 4 geometry: 66 445 triangles in 37 meshes with vertex normals + 1 sphere light, 10 image
    textures (mip levels at the primary vertex), BVH walked from HBM. Textures are decoded from the scene's JPEGs by the build's own decoder."""
    xml = scene_variant(scene_tmp, "sponza/sponza.xml", width=96, height=72)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    info = sc.info()
    assert info["num_tris"] == 66445 and info["num_spheres"] == 1 and info["bvh_depth"] <= 32
    got, st = sc.render(4, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).render(4, G.RNG_SAMPLE, threads=8)
    assert st.bounces == ost.bounces and st.nonfinite_samples == ost.nonfinite_samples
    for k in BUFS[1:]:
        assert np.abs(want[k]).max() > 0
    for k in BUFS:
        err = rel_l2(got[k], want[k])
        assert err < 1e-7, (k, err)
    tile, st2 = sc.render(2, G.RNG_TILE)
    wtile, ost2 = O.OracleScene(sd.ptr, use_bvh=True).render(2, G.RNG_TILE, threads=8)
    assert st2.bounces == ost2.bounces
    for k in BUFS:
        assert rel_l2(tile[k], wtile[k]) < 1e-7, k


def test_textured_two_sided_synthetic_scene(G, O):
    """Image texture with mip levels on the primary vertex, checkerboard, glass (two-sided: offsets survive past the
    first bounce under A-semantics), an emitter — built directly as a GdptSceneDesc."""
    b = DescBuilder(G)
    rng = np.random.default_rng(5)
    tex = b.image_tex(list(rng.random(32 * 16 * 3)), 32, 16, 3, 2.0, 2.0, 0.1, 0.2)
    ct = lambda v: DescBuilder.const_tex(G, v)
    m_floor = b.material(G.MAT_LAMBERTIAN, [tex])
    m_glass = b.material(G.MAT_DISNEY_GLASS, [ct((0.9, 0.8, 0.7)), ct(0.2), ct(0.4)], eta=1.5)
    m_bsdf = b.material(G.MAT_DISNEY_BSDF, [b.checker_tex((0.8, 0.3, 0.2), (0.2, 0.3, 0.8), 4, 4, 0, 0)] +
                        [ct(v) for v in (0.6, 0.1, 0.2, 0.5, 0.3, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7)], eta=1.4)
    m_light = b.material(G.MAT_LAMBERTIAN, [ct(0.0)])
    quad = lambda p: (sum(p, []), [0, 1, 2, 0, 2, 3])
    pos, idx = quad([[-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3]])
    b.mesh(pos, idx, m_floor, uvs=[0, 0, 1, 0, 1, 1, 0, 1], normals=[0, 1, 0] * 4)
    pos, idx = quad([[-1, 0.2, 0.5], [0.2, 0.2, 0.5], [0.2, 1.6, 0.2], [-1, 1.6, 0.2]])
    b.mesh(pos, idx, m_glass)
    pos, idx = quad([[0.4, 0.1, -0.5], [1.8, 0.1, -0.2], [1.8, 1.4, -0.2], [0.4, 1.4, -0.5]])
    b.mesh(pos, idx, m_bsdf, uvs=[0, 0, 1, 0, 1, 1, 0, 1])
    pos, idx = quad([[-1, 3, -1], [1, 3, -1], [1, 3, 1], [-1, 3, 1]])     # wound to face down (-y)
    b.mesh(pos, idx, m_light, light=(12.0, 10.0, 8.0))
    b.sphere((-1.5, 0.6, -1.0), 0.6, m_bsdf)
    # camera: reuse the loader's camera maths through a tiny scene file is overkill; look down -z from (0,1.2,5)
    import math
    cam = b.desc.camera
    cam.width, cam.height, cam.filter_type, cam.filter_param = 40, 30, G.FILTER_GAUSSIAN, 0.5
    c2w = np.eye(4); c2w[:3, 0] = [-1, 0, 0]; c2w[:3, 1] = [0, 1, 0]; c2w[:3, 2] = [0, 0, -1]; c2w[:3, 3] = [0, 1.2, 5]
    aspect = 40 / 30
    cot = 1.0 / math.tan(math.radians(45.0 / 2))
    persp = np.array([[cot, 0, 0, 0], [0, cot, 0, 0], [0, 0, 1, -1], [0, 0, 1, 0]], dtype=float)
    c2s = np.diag([-0.5, -0.5 * aspect, 1, 1]) @ np.array([[1, 0, 0, -1], [0, 1, 0, -1 / aspect], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=float) @ persp
    s2c = np.linalg.inv(c2s)
    for i in range(16):
        cam.sample_to_cam[i] = s2c.ravel()[i]
        cam.cam_to_world[i] = c2w.ravel()[i]
    b.desc.max_depth, b.desc.rr_depth = 6, 3
    desc = b.finish()

    class Holder:      # duck-typed SceneDesc for G.Scene
        ptr = desc
        width, height = 40, 30
    sc = G.Scene(Holder)
    got, st = sc.render(8, G.RNG_SAMPLE)
    want, ost = O.OracleScene(desc).render(8, G.RNG_SAMPLE, threads=8)
    assert st.bounces == ost.bounces and st.nonfinite_samples == ost.nonfinite_samples
    check_buffers(got, want, 1e-7)
    assert np.abs(want["img"]).max() > 0 and np.abs(want["cx0"]).max() > 0


def test_traversal_counters_and_bvh_info(G, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=32, height=32)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    info = sc.info()
    assert info["num_tris"] == 38 and 0 < info["bvh_depth"] <= 32 and info["num_nodes"] >= 9
    import ctypes as C
    st = G.GdptRenderStats()
    st.nodes_visited = 2 ** 64 - 1        # ask for the counting build
    p = G._params(2, G.RNG_SAMPLE, (0, 0))
    bufs = {k: np.zeros((32, 32, 3)) for k in BUFS}
    G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[bufs[k].ctypes.data_as(C.POINTER(C.c_double)) for k in BUFS], C.byref(st)))
    assert st.rays > 0 and st.nodes_visited >= st.rays and st.tris_tested > 0
    plain, st2 = sc.render(2, G.RNG_SAMPLE)
    assert st2.nodes_visited == 0 and st2.rays == st.rays
    for k in BUFS:
        assert np.array_equal(plain[k], bufs[k])       # counting build computes the same image


def test_argument_errors(G, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=32, height=32)
    sc = G.Scene(G.parse_scene(xml))
    with pytest.raises(G.GdptError):
        sc.render(4, G.RNG_SAMPLE, rows=(8, 4))
    with pytest.raises(G.GdptError):
        sc.render(4, G.RNG_TILE, rows=(8, 32))          # tile-stream bands must be whole tile rows
    with pytest.raises(G.GdptError):
        sc.render(4, 99)


def test_roughplastic_and_roughdielectric_boxes(G, O, scene_tmp):
    """the cbox with its two boxes switched to RoughPlastic (one-sided: lane machine with lazy
    offsets) and RoughDielectric (two-sided: eager evaluator), GradPath and Path against the oracle."""
    for variant, tol in ((("roughplastic",), 1e-9), (("roughplastic", "roughdielectric"), 1e-7)):
        xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=32)
        text = open(xml).read()
        text = text.replace('<bsdf type="diffuse" id="box">', '<bsdf type="roughplastic" id="box"><float name="alpha" value="0.1"/>', 1)
        if len(variant) > 1:
            # the first shape that references "box" becomes a rough glass block with its own material
            text = text.replace('</bsdf>', '</bsdf>\n<bsdf type="roughdielectric" id="glassbox"><float name="alpha" value="0.05"/><float name="intIOR" value="1.4"/></bsdf>', 1)
            text = text.replace('<ref id="box"/>', '<ref id="glassbox"/>', 1)
        open(xml, "w").write(text)
        sd = G.parse_scene(xml)
        types = sorted({sd.desc.materials[i].type for i in range(sd.desc.num_materials)})
        assert G.MAT_ROUGHPLASTIC in types and ((G.MAT_ROUGHDIELECTRIC in types) == (len(variant) > 1))
        sc = G.Scene(sd)
        got, st = sc.render(6, G.RNG_SAMPLE)
        want, ost = O.OracleScene(sd.ptr).render(6, G.RNG_SAMPLE, threads=8)
        assert st.bounces == ost.bounces
        check_buffers(got, want, tol)
        img, pst = sc.path_render(4, G.RNG_SAMPLE)
        pwant, post = O.OracleScene(sd.ptr).path_render(4, G.RNG_SAMPLE, threads=8)
        assert rel_l2(img, pwant) < tol and pst.bounces == post.bounces


@pytest.mark.parametrize("scene", ["disney_glass.xml", "disney_bsdf.xml"])
def test_two_sided_lane_machine_replays_offsets_exactly(G, O, scene_tmp, scene, monkeypatch):
    """Two-sided lobes: the lane machine logs the base path's (material, p2) per bounce and replays the four offsets from
    the log (render_twosided.h); the straight-loop evaluator carries them along. Same arithmetic, two schedules — and
    both equal the oracle, rays and bounces included."""
    xml = scene_variant(scene_tmp, "disney_bsdf_test/" + scene, width=64, height=48, integrator="gradpath")
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.render(6, G.RNG_SAMPLE)
    with G.debug_knobs(no_twosided_machine=1):
        eager, est = sc.render(6, G.RNG_SAMPLE)
    with G.debug_knobs(full_material_switch=1):       # the kernel with every lobe vs the one built for this scene's material set
        full, fst = sc.render(6, G.RNG_SAMPLE)
    for k in BUFS:
        assert rel_l2(got[k], full[k]) < 1e-12, k
    assert (fst.rays, fst.bounces) == (st.rays, st.bounces)
    # offsets replay up to four logged iterations per wave step (render_twosided.h: kReplayPerStep): the budget is a
    # schedule, not arithmetic — one iteration per step (as first built) or the whole log at once give the same bits
    for budget in (1, 2, 1000):
        with G.debug_knobs(replay_per_step=budget):
            other, ost2 = sc.render(6, G.RNG_SAMPLE)
        for k in BUFS:
            assert np.array_equal(got[k], other[k], equal_nan=True), (budget, k)
        assert (ost2.rays, ost2.bounces) == (st.rays, st.bounces)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).render(6, G.RNG_SAMPLE, threads=8)
    for k in BUFS:
        assert rel_l2(got[k], eager[k]) < 1e-12, k
    check_buffers(got, want, 1e-7)
    # (samples whose BSDF sampling fails return the zero record: the lane machine never traces their offset rays)
    assert st.bounces == est.bounces == ost.bounces and st.rays <= est.rays
    assert np.abs(want["cx0"]).max() > 0


@pytest.mark.parametrize("rel, integ", [("sponza/sponza.xml", None), ("disney_bsdf_test/disney_metal.xml", "gradpath"), ("cbox/cbox_gdpt.xml", None)])
def test_closest_hit_does_not_depend_on_the_tree(G, scene_tmp, rel, integ, monkeypatch):
    """The hit a ray reports is defined without reference to the BVH (fp32 Möller–Trumbore, smallest t, lowest primitive id
    on ties), so different trees over the same triangles must give bit-identical images. Four trees: the product's (SAH with
    spatial splits, host/sbvh.cpp: triangles referenced from every leaf that holds a part of them), the same with another leaf
    policy, the object-split build (the product's for small scenes such as cbox), the object-split build over pre-split triangles
    (host/presplit.cpp), and spatial splits under a tight reference budget — a traversal that
    skipped a box it should have entered, or a clipped box that lost part of its triangle, shows up here as a differing pixel."""
    xml = scene_variant(scene_tmp, rel, width=160, height=96, integrator=integ)
    sd = G.parse_scene(xml)
    a, sa = G.Scene(sd).render(4, G.RNG_SAMPLE)
    for knobs in ({"bvh_leaf_max": 2}, {"sbvh": 0.0}, {"sbvh": 0.0, "presplit": 0.7, "bvh_leaf_max": 2}, {"sbvh": 0.15, "sbvh_alpha": 0.0}, {"sbvh": 1.0}):
        with G.debug_knobs(**knobs):
            b, sb = G.Scene(sd).render(4, G.RNG_SAMPLE)
        for k in BUFS:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), (k, knobs)
        assert sa.rays == sb.rays and sa.bounces == sb.bounces


def test_kernel_without_sphere_and_texture_code_equals_the_general_one(G, scene_tmp):
    """A scene of triangles with constant textures (cbox) runs a lane machine compiled without the sphere test and the
    texture lookups (device_trace.h: PLAIN); the general Lambertian kernel must give the same bits, counters included —
    LDS-resident and walked from HBM, with ragged edge tiles and items of several samples."""
    for film, spp, extra in (((96, 80), 7, {}), ((200, 120), 24, {}), ((96, 80), 7, {"no_lds_scene": 1})):
        sc = G.Scene(G.parse_scene(scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=film[0], height=film[1])))
        with G.debug_knobs(**extra):
            plain, ps = sc.render(spp, G.RNG_SAMPLE)
        with G.debug_knobs(no_plain_kernel=1, **extra):
            general, gs = sc.render(spp, G.RNG_SAMPLE)
        for k in BUFS:
            assert np.array_equal(plain[k], general[k]), (film, k)
        assert (ps.rays, ps.bounces, ps.samples, ps.nonfinite_samples) == (gs.rays, gs.bounces, gs.samples, gs.nonfinite_samples)
        assert np.abs(plain["cx0"]).max() > 0


@pytest.mark.parametrize("rel, integ, film, spp", [("sponza/sponza.xml", None, (200, 112), 6), ("disney_bsdf_test/disney_metal.xml", "gradpath", (96, 80), 5),
                                                   ("sponza/sponza.xml", None, (33, 17), 3), ("sponza/sponza.xml", None, (640, 360), 24)])
def test_wavefront_pipeline_equals_the_lane_machine(G, O, scene_tmp, rel, integ, film, spp):
    """Scenes walked from HBM can run as a wavefront pipeline (render_wavefront.h: a step kernel over path slots whose state
    lives in HBM, a counting sort of the generation's rays by (octant, origin cell), and a trace kernel of <= 64 VGPRs at 8 waves
    per SIMD, one generation per ray) instead of the lane machine. Same per-sample program,
    same streams, same per-item summation order: the five buffers must be bit-identical, counters included, and equal
    the oracle."""
    xml = scene_variant(scene_tmp, rel, width=film[0], height=film[1], integrator=integ)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    with G.debug_knobs(wavefront=0):
        lane, ls = sc.render(spp, G.RNG_SAMPLE)
    with G.debug_knobs(wavefront=1):
        wave, ws = sc.render(spp, G.RNG_SAMPLE)
        band, bs = sc.render(spp, G.RNG_SAMPLE, rows=(16, 48) if film[1] >= 48 else (0, 16))
    with G.debug_knobs(wavefront=1, wf_slots=max(512, film[0] * film[1] // 4)):        # far fewer slots than work items: every slot runs many items in turn
        few, fs = sc.render(spp, G.RNG_SAMPLE)
    for sort in (0, 2):                           # the ray queue in slot order / sorted cell-major (the default above: octant-major):
        with G.debug_knobs(wavefront=1, wf_sort=sort):       # which lane walks which ray cannot change a hit
            other, os_ = sc.render(spp, G.RNG_SAMPLE)
        for k in BUFS:
            assert np.array_equal(lane[k], other[k]), (k, sort)
        assert (os_.rays, os_.bounces) == (ls.rays, ls.bounces)
    for k in BUFS:
        assert np.array_equal(lane[k], few[k]), k
    assert (fs.rays, fs.bounces) == (ls.rays, ls.bounces)
    for k in BUFS:
        assert np.array_equal(lane[k], wave[k]), k
    assert (ls.rays, ls.bounces, ls.samples, ls.nonfinite_samples) == (ws.rays, ws.bounces, ws.samples, ws.nonfinite_samples)
    r0, r1 = (16, 48) if film[1] >= 48 else (0, 16)
    for k in BUFS:
        assert np.array_equal(band[k][r0:r1], lane[k][r0:r1]), k          # a row band through the same pipeline
    if film[0] * film[1] * spp > 400000:      # (640x360x24: items of up to 10 samples — the accumulation across an item's samples
        return                                # must round alike in both pipelines; too large for the oracle in a test)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).render(spp, G.RNG_SAMPLE, threads=8)
    check_buffers(wave, want, 1e-7)
    assert ws.bounces == ost.bounces


@pytest.mark.parametrize("name", ["disney_diffuse", "disney_metal", "disney_clearcoat", "disney_sheen"])
def test_kernel_built_for_the_material_set_equals_the_full_switch(G, scene_tmp, name):
    """A triangles-only scene whose materials are {Lambertian, one one-sided Disney lobe} runs a lane machine whose material
    switch holds those two arms only (render_phases_general_sets.h; 0-44 instead of 88 spilled VGPRs). No arithmetic differs
    on any path such a scene can take: the kernel with every one-sided lobe (knob full_material_switch) must give the same
    bits and counters."""
    sc = G.Scene(G.parse_scene(scene_variant(scene_tmp, f"disney_bsdf_test/{name}.xml", width=96, height=80, integrator="gradpath")))
    a, sa = sc.render(6, G.RNG_SAMPLE)
    with G.debug_knobs(full_material_switch=1):
        b, sb = sc.render(6, G.RNG_SAMPLE)
    for k in BUFS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k
    assert (sa.rays, sa.bounces, sa.nonfinite_samples) == (sb.rays, sb.bounces, sb.nonfinite_samples)
    assert np.abs(np.asarray(a["cx0"])).max() > 0
