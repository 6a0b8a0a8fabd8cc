"""Manual sweep of the triangle pre-split budget (debug knob presplit, include/gdpt_debug.h) on one GPU (not collected by pytest)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
cases = [("sponza", "sponza/sponza.xml", 1280, 720, None, 16), ("disney_metal", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 16),
         ("veach_mi", "veach_mi/mi.xml", 768, 512, "gradpath", 16)]
ref = {}
for budget, floor in ((0.0, 1e-6), (0.0, 1e-6), (0.3, 1e-6), (1.0, 1e-6), (1.0, 1e-5), (1.0, 1e-4), (3.0, 1e-5), (3.0, 1e-6)):
    G.debug_knobs.reset(); G.debug_knobs.set(presplit_floor=floor, presplit=budget)
    row = []
    for name, rel, w, h, integ, spp in cases:
        xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
        sc = G.Scene(G.parse_scene(xml))
        best = 1e9
        for _ in range(3):
            out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
            best = min(best, rs.render_ms)
        img = np.asarray(bufs["img"]).copy()
        same = "" if name not in ref else (" same" if np.array_equal(ref[name], img, equal_nan=True) else " DIFFERENT")
        ref.setdefault(name, img)
        row.append(f"{name} {rs.samples / best / 1e3:7.1f} Ms/s{same}")
        del sc
    print(f"presplit {budget} floor {floor}: " + " | ".join(row), flush=True)
