"""Manual profiling target (not collected by pytest): disney_glass (two-sided lane machine) 512x512 at SPP (default 32)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
name = os.environ.get("SCENE", "disney_glass")
xml = scene_variant(tempfile.mkdtemp(), f"disney_bsdf_test/{name}.xml", width=512, height=512, integrator="gradpath")
sc = G.Scene(G.parse_scene(xml))
spp = int(os.environ.get("SPP", "32"))
for i in range(2):
    bufs, st = sc.render(spp, G.RNG_SAMPLE)
print("render_ms", st.render_ms, "Msamples/s", st.samples / st.render_ms / 1e3, "rays/sample", st.rays / st.samples)
