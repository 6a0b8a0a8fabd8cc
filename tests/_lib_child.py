"""Child process of tests/test_gpu_bvh8_variant.py: renders a few small GradPath cases with ANOTHER build of libgdpt.so and prints
the SHA-1 of every buffer plus the counters as one JSON line.    python tests/_lib_child.py <lib.so|->"""
import os, sys, json, hashlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gdpt_amd as G
if sys.argv[1] != "-":
    G.LIB_PATH = sys.argv[1]
from helpers import scene_variant

CASES = (("sponza", "sponza/sponza.xml", None, (160, 96), 4, {}),
         ("disney_metal", "disney_bsdf_test/disney_metal.xml", "gradpath", (96, 80), 5, {}),
         ("disney_glass", "disney_bsdf_test/disney_glass.xml", "gradpath", (64, 48), 3, {}),
         ("cbox_from_hbm", "cbox/cbox_gdpt.xml", None, (96, 80), 7, {"no_lds_scene": 1}))


def run():
    tmp = tempfile.mkdtemp()
    out = {}
    for name, rel, integ, film, spp, knobs in CASES:
        sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=film[0], height=film[1], integrator=integ)))
        with G.debug_knobs(**knobs):
            bufs, st = sc.render(spp, G.RNG_SAMPLE)
        out[name] = {k: hashlib.sha1(np.ascontiguousarray(bufs[k]).tobytes()).hexdigest() for k in ("img", "cx0", "cy0", "cx1", "cy1")}
        out[name].update(rays=int(st.rays), bounces=int(st.bounces), samples=int(st.samples))
    return out


if __name__ == "__main__":
    print("RESULT " + json.dumps(run()))
