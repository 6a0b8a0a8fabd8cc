"""Manual profiling driver (not collected by pytest): one scene through the wavefront pipeline (or the lane machine), twice.
usage: prof_wf_once.py [sponza|metal] [spp] [knob=value ...]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
knobs = dict(wavefront=1)
for kv in sys.argv[3:]:
    k, v = kv.split("="); knobs[k] = float(v)
if which == "sponza":
    sd = G.parse_scene(os.path.join(ROOT, "scenes/sponza/sponza.xml"), film=(1280, 720))
else:
    sd = G.parse_scene(scene_variant(tempfile.mkdtemp(), "disney_bsdf_test/disney_metal.xml", integrator="gradpath"), film=(512, 512))
sc = G.Scene(sd)
with G.debug_knobs(**knobs):
    for _ in range(2):
        bufs, st = sc.render(spp, G.RNG_SAMPLE)
print(f"{which} spp {spp} {knobs}: {st.render_ms:.2f} ms, {st.samples / st.render_ms / 1e3:.0f} Msamples/s, {st.rays / st.render_ms / 1e6:.2f} Grays/s", flush=True)
