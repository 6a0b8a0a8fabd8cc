"""Manual sweep of lanes per pixel (debug knob log2k, include/gdpt_debug.h) for the reconnect kernel on one GPU (not collected by pytest)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name, rel, w, h, integ, spp in (("cbox", "cbox/cbox_gdpt.xml", 512, 512, None, 16), ("sponza", "sponza/sponza.xml", 1280, 720, None, 16),
                                    ("veach_mi", "veach_mi/mi.xml", 768, 512, "gradpath", 16)):
    xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
    sc = G.Scene(G.parse_scene(xml))
    row = []
    for k in ("", "0", "1", "2", "3", "4"):
        G.debug_knobs.reset()
        if k: G.debug_knobs.set(log2k=int(k))
        best = 1e9
        for _ in range(3):
            b, st = sc.render(spp, G.RNG_SAMPLE, shift=G.SHIFT_RECONNECT)
            best = min(best, st.render_ms)
        row.append(f"log2k={k or 'auto'}: {best:.2f} ms")
    G.debug_knobs.reset()
    print(name, " | ".join(row), flush=True)
