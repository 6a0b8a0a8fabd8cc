"""Manual sweep of the BVH leaf policy knobs (debug knobs bvh_leaf_max / bvh_leaf_factor, include/gdpt_debug.h) on one GPU (not collected by pytest)."""
import os, sys, tempfile, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
cases = [("cbox", "cbox/cbox_gdpt.xml", 512, 512, None, 64), ("sponza", "sponza/sponza.xml", 1280, 720, None, 16),
         ("disney_metal", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 16)]
for lm, lf in [(4, 0.8), (4, 0.6), (4, 1.0), (4, 1.3), (3, 0.8), (2, 0.8), (4, 0.8), (4, 0.5), (3, 0.6), (4, 0.7)]:
    G.debug_knobs.reset(); G.debug_knobs.set(bvh_leaf_max=lm, bvh_leaf_factor=lf)
    row = []
    for name, rel, w, h, integ, spp in cases:
        xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
        sc = G.Scene(G.parse_scene(xml))
        best = 1e9
        for _ in range(3):
            out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
            best = min(best, rs.render_ms)
        row.append(f"{name} {rs.samples / best / 1e3:7.1f} Ms/s")
        del sc
    print(f"leaf_max {lm} factor {lf}: " + " | ".join(row), flush=True)
