import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for rel in ("disney_bsdf_test/disney_bsdf.xml", "disney_bsdf_test/disney_glass.xml"):
    xml = scene_variant(tmp, rel, width=512, height=512, integrator="gradpath")
    sc = G.Scene(G.parse_scene(xml))
    for i in range(2):
        b, st = sc.render(64, G.RNG_SAMPLE)
    print(rel, st.render_ms, st.samples / st.render_ms / 1e3)
