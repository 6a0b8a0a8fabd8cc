"""Manual profiling target (not collected by pytest): sponza 1280x720 at SPP (default 8)."""
import os, sys, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
import numpy as np
from helpers import scene_variant
xml = scene_variant(tempfile.mkdtemp(), "sponza/sponza.xml", width=1280, height=720)
sc = G.Scene(G.parse_scene(xml))
print(sc.info())
spp = int(os.environ.get("SPP", "8"))
for i in range(2):
    bufs, st = sc.render(spp, G.RNG_SAMPLE)
print("render_ms", st.render_ms, "Msamples/s", st.samples / st.render_ms / 1e3, "rays/sample", st.rays / st.samples, "bounces/sample", st.bounces / st.samples)
cs = G.GdptRenderStats(); cs.nodes_visited = 2 ** 64 - 1
p = G._params(spp, G.RNG_SAMPLE, (0, 0))
b = {k: np.zeros((720, 1280, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[b[k].ctypes.data_as(C.POINTER(C.c_double)) for k in b], C.byref(cs)))
print("nodes/ray", cs.nodes_visited / cs.rays, "prims/ray", cs.tris_tested / cs.rays, "counting ms", cs.render_ms)
print("node-trip lane utilisation", cs.nodes_visited / max(1, 64 * cs.wave_node_trips), "node trips/ray-step", cs.wave_node_trips / max(1, cs.wave_steps),
      "leaf trips/step", cs.wave_leaf_trips / max(1, cs.wave_steps), "lanes/step", cs.lane_steps / max(1, cs.wave_steps), "wave steps", cs.wave_steps)
