"""Manual experiment (not collected by pytest): the fixed cost of a render launch = T(16 spp) - T(64 spp)/4 scaled, with
unbounded paths (Russian roulette only) and with max_depth 8 (no long tails): how much of it is the drain of the longest paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import gdpt_amd as G
sd = G.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml"))
sc = G.Scene(sd)
shape = (sc.height, sc.width, 3)
bufs = [np.zeros(shape) for _ in range(5)]
def t(spp, md, reps=5):
    best = 1e9
    for _ in range(reps):
        st = G.GdptRenderStats()
        p = G._params(spp, G.RNG_SAMPLE, (0, 0), max_depth_override=md)
        G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs], C.byref(st)))
        best = min(best, st.render_ms)
    return best, st
for md in (0, 8, 4):
    a, sa = t(16, md); b, sb = t(64, md); c, scc = t(256, md, 2)
    slope = (c - b) / 192.0
    print(f"max_depth {md or 'inf'}: T16 {a:.3f} T64 {b:.3f} T256 {c:.3f} ms; per-spp {slope:.4f} ms; fixed {a - 16 * slope:.3f} ms; "
          f"bounces/sample {sa.bounces / sa.samples:.2f}", flush=True)
