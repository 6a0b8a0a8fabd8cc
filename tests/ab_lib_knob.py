"""Manual sweep (not collected by pytest): one debug knob over several values, for one build of libgdpt.so ("-" = the tree's).
    python tests/ab_lib_knob.py <lib.so|-> <knob> v1,v2,...      (Msamples/s per scene, best of 3)"""
import os, sys, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
if sys.argv[1] != "-":
    G.LIB_PATH = sys.argv[1]
from helpers import scene_variant
knob = sys.argv[2]; values = [int(v) for v in sys.argv[3].split(",")]
tmp = tempfile.mkdtemp()
scenes = {}
for name, rel, w, h, integ, spp in (("sponza", "sponza/sponza.xml", 1280, 720, None, 16), ("metal", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 64),
                                    ("glass", "disney_bsdf_test/disney_glass.xml", 512, 512, "gradpath", 32), ("bsdf", "disney_bsdf_test/disney_bsdf.xml", 512, 512, "gradpath", 32)):
    scenes[name] = (G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator=integ))), spp)
for v in values:
    out = {}
    with G.debug_knobs(**{knob: v}):
        for name, (sc, spp) in scenes.items():
            best = 1e9
            for _ in range(3):
                _, st = sc.render(spp, G.RNG_SAMPLE); best = min(best, st.render_ms)
            out[name] = round(st.samples / best / 1e3, 1)
    print(os.path.basename(sys.argv[1]), knob, v, json.dumps(out), flush=True)
