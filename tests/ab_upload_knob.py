"""Manual A/B (not collected by pytest), same process and box: a debug knob that acts at scene upload (tree build / layout), off/on.
   python tests/ab_upload_knob.py <knob> [value_on [case,case.. [other=value,..]]] — best of 5 render times per setting, and whether the five
   buffers are bit-identical; `other=value` knobs are set beside the ON setting only."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
knob = sys.argv[1]; on = float(sys.argv[2]) if len(sys.argv) > 2 else 1
on = int(on) if on == int(on) else on
tmp = tempfile.mkdtemp()
cases = [("sponza 1280x720x16", "sponza/sponza.xml", 1280, 720, None, 16), ("disney_diffuse 512x512x16", "disney_bsdf_test/disney_diffuse.xml", 512, 512, "gradpath", 16),
         ("disney_metal 512x512x16", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 16),
         ("disney_glass 512x512x16", "disney_bsdf_test/disney_glass.xml", 512, 512, "gradpath", 16), ("disney_bsdf 512x512x16", "disney_bsdf_test/disney_bsdf.xml", 512, 512, "gradpath", 16)]
if len(sys.argv) > 3 and sys.argv[3] != "all":
    cases = [c for c in cases if any(c[0].startswith(w) for w in sys.argv[3].split(","))]
extra = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in sys.argv[4].split(",")} if len(sys.argv) > 4 else {}
for name, rel, w, h, integ, spp in cases:
    sd = G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator=integ))
    scs = {}
    for mode in (0, on):
        with G.debug_knobs(**({knob: mode, **extra} if mode else {})):      # baseline: no knob set at all (the product path)
            scs[mode] = G.Scene(sd)
    res, bufs, sts = {}, {}, {}
    for rep in range(5):
        for mode in (0, on):
            b, st = scs[mode].render(spp, G.RNG_SAMPLE)
            res.setdefault(mode, []).append(st.render_ms); bufs[mode] = b; sts[mode] = st
    same = all(np.array_equal(bufs[0][k], bufs[on][k], equal_nan=True) for k in bufs[0])
    print(f"{name}: {extra if extra else ''} default {min(res[0]):.3f} ms ({st.samples / min(res[0]) / 1e3:.1f} Msamples/s) | {knob}={on} {min(res[on]):.3f} ms "
          f"({st.samples / min(res[on]) / 1e3:.1f} Msamples/s) | buffers identical: {same}", flush=True)
