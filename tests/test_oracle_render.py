"""The oracle's GradPath restatement: statistics recorded by the survey from a shim-linked run of the reference
(SURVEY.md Appendix A.3), internal consistency (brute force == BVH, bands == whole image), RNG schemes."""
import os

import numpy as np
import pytest

from helpers import SCENES, rel_l2, scene_variant


def test_survey_statistics_cbox_512_4spp_tile_stream(G, O):
    """SURVEY.md A.3 (reference sources + a brute-force rtc shim, gcc 11 -O2): 1 048 576 samples, 71 546 primary
    misses, 3 167 356 bounce iterations, x0 valid for 968 693 samples. Primary rays depend only on PCG32, the camera,
    the filter and the geometry: reproduced exactly. Bounce totals depend on fp32 hit arithmetic at edges (the shim's
    is not available), so they are matched to 2e-4."""
    sd = G.parse_scene(os.path.join(SCENES, "cbox", "cbox_gdpt.xml"))
    sc = O.OracleScene(sd.ptr, use_bvh=True)
    bufs, st = sc.render(4, G.RNG_TILE, threads=os.cpu_count())
    assert st.samples == 1048576
    assert st.primary_misses == 71546
    assert abs(int(st.bounces) - 3167356) / 3167356 < 2e-4
    assert abs(int(st.x0_valid_initial) - 968693) / 968693 < 1e-4
    assert st.nonfinite_samples == 0
    # image means of the survey's probe run at the noise level of 4 spp
    means = bufs["img"].mean(axis=(0, 1))
    np.testing.assert_allclose(means, [0.2786, 0.1124, 0.0251], rtol=0.02)


def test_bruteforce_equals_oracle_bvh(G, O, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=40, height=24)
    sd = G.parse_scene(xml)
    a, sa = O.OracleScene(sd.ptr, use_bvh=False).render(3, G.RNG_SAMPLE, threads=4)
    b, sb = O.OracleScene(sd.ptr, use_bvh=True).render(3, G.RNG_SAMPLE, threads=4)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert sa.bounces == sb.bounces and sa.rays == sb.rays


def test_bands_equal_whole_image_and_threads_do_not_matter(G, O, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=32, height=48)
    sd = G.parse_scene(xml)
    sc = O.OracleScene(sd.ptr)
    for scheme in (G.RNG_TILE, G.RNG_SAMPLE):
        whole, _ = sc.render(2, scheme, threads=1)
        again, _ = sc.render(2, scheme, threads=5)
        top, _ = sc.render(2, scheme, rows=(0, 16), threads=3)
        rest, _ = sc.render(2, scheme, rows=(16, 48), threads=3)
        for k in whole:
            assert np.array_equal(whole[k], again[k])
            assert np.array_equal(whole[k], top[k] + rest[k])       # untouched rows stay zero
            assert not top[k][16:].any() and not rest[k][:16].any()


def test_sample_record_semantics(G, O):
    sd = G.parse_scene(os.path.join(SCENES, "cbox", "cbox_gdpt.xml"))
    sc = O.OracleScene(sd.ptr)
    state, inc = O.pcg_init(527)
    seen_miss = seen_hit = False
    for i in range(400):
        x, y = (i * 37) % 512, (i * 91) % 512
        rec, state2 = sc.grad_sample(x, y, state, inc)
        if rec.primary_miss:
            seen_miss = True     # base miss => default record: zeros, prob = 1, weights 1 (src/path_tracing.h:375-379)
            assert rec.prob == 1.0 and list(rec.radiance) == [0, 0, 0] and rec.wX0 == rec.wY1 == 1.0 and rec.rng_draws == 2
        else:
            seen_hit = True
            assert rec.rng_draws >= 5 and rec.bounces >= 1
            for wgt in (rec.wX0, rec.wX1, rec.wY0, rec.wY1):
                assert 0.0 <= wgt <= 1.0
        state = state2
    assert seen_hit


def test_max_depth_bounds_the_bounce_loop(G, O, scene_tmp):
    for md, max_bounces in ((1, 0), (2, 1), (3, 2)):
        xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=16, height=16, max_depth=md)
        sd = G.parse_scene(xml)
        sc = O.OracleScene(sd.ptr)
        _, st = sc.render(2, G.RNG_SAMPLE, threads=2)
        assert st.bounces <= max_bounces * st.samples
        if max_bounces == 0:
            assert st.bounces == 0


def test_sphere_scene_renders_finite(G, O, scene_tmp):
    xml = scene_variant(scene_tmp, "cbox/small_pt_compare.xml", width=24, height=16)
    sd = G.parse_scene(xml)
    sc = O.OracleScene(sd.ptr)
    bufs, st = sc.render(2, G.RNG_SAMPLE, threads=4)
    assert st.nonfinite_samples == 0 and np.isfinite(bufs["img"]).all()
    assert st.primary_misses < st.samples
