"""Integrator::Path: path_tracing with next-event estimation + MIS (src/path_tracing.h:13-348)
under the path_render tile loop (src/render.cpp:74-117).

CPU part: the oracle's restatement against statistics of renders the reference repository ships (cbox_path/*.exr; the
oracle's emitter-sampling pieces are pinned against the reference's own functions in test_oracle_vs_reference_kat.py).
GPU part: the HIP kernels (render_path.hip) through the C ABI against the oracle on the same PCG streams."""
import json
import os

import numpy as np
import pytest

from helpers import ROOT, rel_l2, scene_variant, DescBuilder

TOL = 1e-9


def _ref_stats(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_images.json")))["reference_renders"][name]


def test_oracle_path_agrees_with_the_reference_own_renders(G, O, scene_tmp):
    """cbox_path/cb_1000.exr (512x512, written by the reference): channel means within 1 % of an independent oracle
    render at 128x128x32 spp (image means do not depend on the resolution in expectation)."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=128, height=128, integrator="path")
    sd = G.parse_scene(xml)
    img, st = O.OracleScene(sd.ptr, use_bvh=True).path_render(32, G.RNG_SAMPLE, threads=8)
    assert st.samples == 128 * 128 * 32 and st.nonfinite_samples == 0
    gold = _ref_stats("cbox_path/cb_1000.exr")
    ratio = img.mean(axis=(0, 1)) / np.array(gold["mean"])
    assert np.all(np.abs(ratio - 1) < 0.01), ratio
    assert (img >= 0).all() and gold["negative_fraction"] == 0.0


def test_oracle_path_tile_and_sample_streams_and_bands(G, O, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=48, integrator="path")
    sd = G.parse_scene(xml)
    sc = O.OracleScene(sd.ptr)
    whole, _ = sc.path_render(2, G.RNG_SAMPLE, threads=4)
    top, _ = sc.path_render(2, G.RNG_SAMPLE, rows=(0, 16), threads=2)
    bottom, _ = sc.path_render(2, G.RNG_SAMPLE, rows=(16, 48), threads=2)
    assert np.array_equal(top + bottom, whole)                      # bands do not interact
    with_bvh, _ = O.OracleScene(sd.ptr, use_bvh=True).path_render(2, G.RNG_SAMPLE, threads=4)
    assert np.array_equal(with_bvh, whole)                          # closest hit / occlusion do not depend on the walk
    tile, _ = sc.path_render(2, G.RNG_TILE, threads=4)
    assert tile.shape == whole.shape and not np.array_equal(tile, whole)
    assert abs(tile.mean() / whole.mean() - 1) < 0.1


def test_oracle_path_with_environment_map(G, O, scene_tmp):
    """Envmap light (src/lights/envmap.inl): importance table over luminance*sin(elevation), lat-long lookup, MIS with the
    BSDF-sampled miss. The Disney test scenes are lit by scenes/matpreview/envmap.exr (PIZ) only."""
    xml = scene_variant(scene_tmp, "disney_bsdf_test/disney_diffuse.xml", width=48, height=48, integrator="path")
    sd = G.parse_scene(xml)
    assert sd.desc.has_envmap == 1 and sd.desc.num_lights == 1 and sd.desc.lights[0].shape_id == -1
    assert sd.desc.envmap.light_id == 0 and sd.desc.envmap.scale == 3.0
    sc = O.OracleScene(sd.ptr, use_bvh=True)
    pmf, cdf = sc.light_table(1)
    assert pmf[0] == 1.0 and cdf[1] > 0            # power = pi r^2 * table total / (w h), src/lights/envmap.inl:1-5
    img, st = sc.path_render(4, G.RNG_SAMPLE, threads=8)
    assert np.isfinite(img).all() and img.mean() > 0.05 and st.nonfinite_samples == 0
    corner = img[:4, :4].mean(axis=(0, 1))          # camera rays that miss everything see the environment directly
    assert np.all(corner > 0)


# ---------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("w,h,spp", [(64, 64, 8), (33, 17, 3), (16, 16, 1)])
def test_gpu_cbox_path_sample_stream(G, O, scene_tmp, w, h, spp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=w, height=h, integrator="path")
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.path_render(spp, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr).path_render(spp, G.RNG_SAMPLE, threads=8)
    assert np.isfinite(got).all() and rel_l2(got, want) < TOL
    # (the persistent kernel skips shadow rays whose contribution is zero anyway, so only the bounce count is comparable)
    assert st.samples == w * h * spp == ost.samples and st.bounces == ost.bounces and st.rays <= ost.rays
    again, _ = sc.path_render(spp, G.RNG_SAMPLE)
    assert np.array_equal(got, again)


@pytest.mark.gpu
def test_gpu_cbox_path_tile_stream_and_bands(G, O, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=48, height=40, integrator="path")
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.path_render(3, G.RNG_TILE)
    want, ost = O.OracleScene(sd.ptr).path_render(3, G.RNG_TILE, threads=4)
    assert rel_l2(got, want) < TOL and st.bounces == ost.bounces and st.rays == ost.rays      # straight loop: every ray of the reference
    whole, _ = sc.path_render(4, G.RNG_SAMPLE)
    band = np.zeros_like(whole)
    for rows in ((0, 16), (16, 40)):
        part, _ = sc.path_render(4, G.RNG_SAMPLE, rows=rows)
        band[rows[0]:rows[1]] = part[rows[0]:rows[1]]
    assert np.array_equal(band, whole)


@pytest.mark.gpu
def test_gpu_sponza_path_sphere_light_and_textures(G, O, scene_tmp):
    """Sphere emitter (cone sampling), image textures, 66k triangles: shadow rays are any-hit walks of the BVH4."""
    xml = scene_variant(scene_tmp, "sponza/sponza.xml", width=64, height=48, integrator="path")
    sd = G.parse_scene(xml)
    got, st = G.Scene(sd).path_render(2, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).path_render(2, G.RNG_SAMPLE, threads=8)
    assert rel_l2(got, want) < 1e-7
    assert st.bounces == ost.bounces and st.rays <= ost.rays


@pytest.mark.gpu
def test_gpu_path_mesh_and_sphere_emitters_with_disney_lobes(G, O):
    """Two emitters with different powers (light selection table), a mesh with vertex normals as emitter, glossy and
    refractive lobes (eta_scale in the roulette), max_depth bound."""
    b = DescBuilder(G)
    c = DescBuilder.const_tex
    lam = b.material(G.MAT_LAMBERTIAN, [c(G, [0.6, 0.5, 0.4])])
    metal = b.material(G.MAT_DISNEY_METAL, [c(G, [0.9, 0.7, 0.3]), c(G, 0.3), c(G, 0.2)])
    glass = b.material(G.MAT_DISNEY_GLASS, [c(G, [0.9, 0.95, 1.0]), c(G, 0.2), c(G, 0.1)], eta=1.4)
    b.mesh([-3, -1, -3, 3, -1, -3, 3, -1, 3, -3, -1, 3], [0, 2, 1, 0, 3, 2], lam)                  # floor
    b.mesh([-3, -1, -3, 3, -1, -3, 3, 3, -3, -3, 3, -3], [0, 1, 2, 0, 2, 3], metal)               # back wall
    b.sphere([0.0, -0.2, 0.0], 0.8, glass)
    up = [0, -1, 0] * 4
    b.mesh([-0.7, 2.5, -0.7, 0.7, 2.5, -0.7, 0.7, 2.5, 0.7, -0.7, 2.5, 0.7], [0, 1, 2, 0, 2, 3], lam, normals=up, light=[12.0, 11.0, 10.0])
    b.sphere([2.0, 1.0, 1.5], 0.3, lam, light=[30.0, 10.0, 5.0])
    import math
    cam = b.desc.camera
    cam.width, cam.height, cam.filter_type, cam.filter_param = 48, 32, G.FILTER_GAUSSIAN, 0.5
    c2w = np.eye(4); c2w[:3, 0] = [-1, 0, 0]; c2w[:3, 1] = [0, 1, 0]; c2w[:3, 2] = [0, 0, -1]; c2w[:3, 3] = [0, 0.8, 6]
    aspect = 48 / 32
    cot = 1.0 / math.tan(math.radians(50.0 / 2))
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
        width, height = 48, 32
    sc = G.Scene(Holder)
    got, st = sc.path_render(6, G.RNG_SAMPLE)
    want, ost = O.OracleScene(desc).path_render(6, G.RNG_SAMPLE, threads=8)
    assert want.mean() > 0.01
    assert rel_l2(got, want) < 1e-7
    assert st.bounces == ost.bounces and st.rays <= ost.rays


@pytest.mark.gpu
def test_gpu_path_persistent_and_straight_loop_agree(G, scene_tmp, monkeypatch):
    """The lane machine and the straight per-sample loop are two schedules of the same arithmetic."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=24, integrator="path")
    sc = G.Scene(G.parse_scene(xml))
    ref, rst = sc.path_render(16, G.RNG_SAMPLE)
    for env in ({"force_eager": 1}, {"no_lds_scene": 1}, {"log2k": 0}, {"keep_frac": 0, "search_frac": 0}):
        with G.debug_knobs(**env):
            got, st = sc.path_render(16, G.RNG_SAMPLE)
        assert rel_l2(got, ref) < 1e-12 and st.bounces == rst.bounces, env


@pytest.mark.gpu
def test_gpu_path_output_agrees_with_the_reference_own_render(G):
    """cbox_path/cb_1000.exr written by the reference: channel means within 1 %, 32x32 block means within 3 % rel. L2."""
    gold = _ref_stats("cbox_path/cb_1000.exr")
    import tempfile
    xml = scene_variant(tempfile.mkdtemp(), "cbox/cbox_gdpt.xml", integrator="path")
    sc = G.Scene(G.parse_scene(xml))
    img, st = sc.path_render(64, G.RNG_SAMPLE)
    h, w, _ = img.shape
    assert (w, h) == (gold["width"], gold["height"]) and st.nonfinite_samples == 0
    ratio = img.mean(axis=(0, 1)) / np.array(gold["mean"])
    assert np.all(np.abs(ratio - 1) < 0.01), ratio
    bs = 32
    thumb = img.reshape(h // bs, bs, w // bs, bs, 3).mean(axis=(1, 3))
    ref = np.array(gold["block_mean_32"])
    assert np.linalg.norm(thumb - ref) / np.linalg.norm(ref) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["disney_diffuse.xml", "disney_metal.xml", "disney_glass.xml", "disney_bsdf.xml"])
def test_gpu_path_environment_map_scenes(G, O, scene_tmp, scene):
    """Envmap-lit Disney scenes through Integrator::Path: sampling of the lat-long importance table, any-hit shadow
    rays to infinity, MIS-weighted environment lookups of BSDF-sampled misses; both schedules against the oracle."""
    xml = scene_variant(scene_tmp, "disney_bsdf_test/" + scene, width=64, height=48, integrator="path")
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    got, st = sc.path_render(4, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).path_render(4, G.RNG_SAMPLE, threads=8)
    assert np.isfinite(got).all() and want.mean() > 0.05
    assert rel_l2(got, want) < 1e-6 and st.bounces == ost.bounces
    tile, tst = sc.path_render(2, G.RNG_TILE)
    twant, tost = O.OracleScene(sd.ptr, use_bvh=True).path_render(2, G.RNG_TILE, threads=8)
    assert rel_l2(tile, twant) < 1e-6 and tst.bounces == tost.bounces and tst.rays == tost.rays


@pytest.mark.gpu
def test_gpu_cli_renders_path_scenes(G, scene_tmp, tmp_path):
    """lajolla dispatches on the scene's integrator like render() (src/render.cpp:374-392) and writes an EXR the
    build's reader (and the reference's, see test_image_io.py) can load."""
    import subprocess
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=48, height=32, integrator="path")
    exe = os.path.join(ROOT, "gradient-based-path-tracing_amd", "lajolla")
    out = tmp_path / "p.exr"
    r = subprocess.run([exe, "-o", str(out), "--spp", "4", xml], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = G.imread(str(out), 3)
    want, _ = G.Scene(G.parse_scene(xml)).path_render(4, G.RNG_SAMPLE)
    assert img.shape == (32, 48, 3)
    assert np.allclose(img, want, rtol=2e-3, atol=1e-4)             # fp16 storage


@pytest.mark.gpu
@pytest.mark.parametrize("rel,filt", [("veach_mi/mi.xml", None), ("pixel_filter_test/pixel_filter_test.xml", None),
                                      ("pixel_filter_test/pixel_filter_test.xml", "box"), ("pixel_filter_test/pixel_filter_test.xml", "tent")])
def test_gpu_reference_test_scenes(G, O, scene_tmp, rel, filt):
    """The reference's own test scenes for this integrator: veach_mi (five sphere emitters of very different power over
    RoughPlastic plates, `direct` = path with max_depth 2) and pixel_filter_test (checkerboard of scale 1000 under
    Gaussian / box / tent pixel filters, src/filters/*.inl), Path and GradPath, against the oracle."""
    xml = scene_variant(scene_tmp, rel, width=96, height=64)
    if filt is not None:
        text = open(xml).read()
        import re
        text = re.sub(r'<rfilter type="gaussian">.*?</rfilter>', f'<rfilter type="{filt}"/>', text, flags=re.S)
        open(xml, "w").write(text)
    sd = G.parse_scene(xml)
    if filt is not None:
        assert sd.desc.camera.filter_type == {"box": G.FILTER_BOX, "tent": G.FILTER_TENT}[filt]
    sc = G.Scene(sd)
    osc = O.OracleScene(sd.ptr, use_bvh=True)
    got, st = sc.path_render(6, G.RNG_SAMPLE)
    want, ost = osc.path_render(6, G.RNG_SAMPLE, threads=8)
    assert want.mean() > 0 and rel_l2(got, want) < 1e-7 and st.bounces == ost.bounces
    gb, gst = sc.render(4, G.RNG_SAMPLE)
    ob, gost = osc.render(4, G.RNG_SAMPLE, threads=8)
    for k in ("img", "cx0", "cy0", "cx1", "cy1"):
        assert rel_l2(gb[k], ob[k]) < 1e-7, k
    assert gst.bounces == gost.bounces


@pytest.mark.gpu
def test_gpu_matpreview_scene(G, O, scene_tmp):
    """scenes/matpreview/matpreview.xml as shipped by the reference: Mitsuba .serialized meshes (61 600 triangles), a
    RoughDielectric material ball, checkerboard floor, lit by the PIZ-compressed environment map only."""
    xml = scene_variant(scene_tmp, "matpreview/matpreview.xml", width=64, height=64)
    sd = G.parse_scene(xml)
    assert sd.desc.integrator == G.INTEGRATOR_PATH and sd.desc.has_envmap == 1
    sc = G.Scene(sd)
    got, st = sc.path_render(4, G.RNG_SAMPLE)
    want, ost = O.OracleScene(sd.ptr, use_bvh=True).path_render(4, G.RNG_SAMPLE, threads=8)
    assert want.mean() > 0.1 and rel_l2(got, want) < 1e-6 and st.bounces == ost.bounces


@pytest.mark.gpu
@pytest.mark.parametrize("scene,ref", [("disney_glass.xml", "extra_images/disney_glass_eta_1.5.exr"),
                                       ("disney_sheen.xml", "extra_images/disney_sheen_test_1.0.exr")])
def test_gpu_envmap_path_agrees_with_renders_shipped_by_the_reference(G, scene, ref):
    """End-to-end anchor of the environment-map code (importance table, lat-long lookup, MIS-weighted miss term, PIZ
    input): the Disney test scenes exactly as the reference ships them (683x512, Integrator::Path) against the images
    the reference repository holds for them. Measured at 64 spp: channel means within 0.04 %, 32x32 block means within
    0.6 % relative L2; asserted at 0.5 % / 2 %."""
    gold = _ref_stats(ref)
    sc = G.Scene(G.parse_scene(os.path.join(ROOT, "scenes", "disney_bsdf_test", scene)))
    img, st = sc.path_render(64, G.RNG_SAMPLE)
    h, w, _ = img.shape
    assert (w, h) == (gold["width"], gold["height"]) and st.nonfinite_samples == 0
    ratio = img.mean(axis=(0, 1)) / np.array(gold["mean"])
    assert np.all(np.abs(ratio - 1) < 0.005), ratio
    bs = 32
    thumb = img[:h // bs * bs, :w // bs * bs].reshape(h // bs, bs, w // bs, bs, 3).mean(axis=(1, 3))
    refb = np.array(gold["block_mean_32"])
    assert np.linalg.norm(thumb - refb) / np.linalg.norm(refb) < 0.02
