"""Manual profiling target (not collected by pytest): cbox 512x512 at SPP (default 16), SAMPLE streams."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G
sc = G.Scene(G.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml")))
spp = int(os.environ.get("SPP", "16"))
for i in range(3):
    bufs, st = sc.render(spp, G.RNG_SAMPLE)
print("render_ms", st.render_ms, "Msamples/s", st.samples / st.render_ms / 1e3)
if os.environ.get("COUNT", "1") == "1":
    cs = G.GdptRenderStats(); cs.nodes_visited = 2 ** 64 - 1
    p = G._params(spp, G.RNG_SAMPLE, (0, 0))
    G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[bufs[k].ctypes.data_as(C.POINTER(C.c_double)) for k in ("img", "cx0", "cy0", "cx1", "cy1")], C.byref(cs)))
    print("nodes/ray", cs.nodes_visited / cs.rays, "prims/ray", cs.tris_tested / cs.rays,
          "node-trip util", cs.nodes_visited / max(1, 64 * cs.wave_node_trips), "node trips/step", cs.wave_node_trips / max(1, cs.wave_steps),
          "leaf trips/step", cs.wave_leaf_trips / max(1, cs.wave_steps), "lanes/step", cs.lane_steps / max(1, cs.wave_steps), "steps", cs.wave_steps)
