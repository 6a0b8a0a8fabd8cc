"""Manual A/B (not collected by pytest), same process and box: Integrator::Path on cbox with one debug knob off/on.
   python tests/ab_knob_path.py <knob> [value_on]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
knob = sys.argv[1]; on = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = tempfile.mkdtemp()
for name, rel, w, h, spp in (("cbox path 512x512x64", "cbox/cbox_gdpt.xml", 512, 512, 64), ("cbox path 200x120x16", "cbox/cbox_gdpt.xml", 200, 120, 16)):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator="path")))
    res, img = {}, {}
    for rep in range(5):
        for mode in (0, on):
            with G.debug_knobs(**{knob: mode}):
                im, st = sc.path_render(spp, G.RNG_SAMPLE)
            res.setdefault(mode, []).append(st.render_ms); img[mode] = im
    print(f"{name}: {knob}=0 {min(res[0]):.3f} ms ({st.samples / min(res[0]) / 1e3:.1f} Msamples/s) | {knob}={on} {min(res[on]):.3f} ms "
          f"({st.samples / min(res[on]) / 1e3:.1f} Msamples/s) | image identical: {np.array_equal(img[0], img[on])}", flush=True)
