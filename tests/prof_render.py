"""Manual profiling target (not collected by pytest): a few launches of the render kernel on cbox 512x512x16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G
sd = G.parse_scene(os.path.join(ROOT, "scenes", "cbox", "cbox_gdpt.xml"))
sc = G.Scene(sd)
for i in range(3):
    bufs, st = sc.render(int(os.environ.get("SPP", "16")), G.RNG_SAMPLE)
print("render_ms", st.render_ms, "rays", st.rays)
