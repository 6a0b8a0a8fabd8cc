"""Manual sweep (not collected by pytest): the work-item plan's share knob on Integrator::Path scenes."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name, rel, w, h, spp in (("matpreview path", "matpreview/matpreview.xml", 512, 512, 32), ("cbox path", "cbox/cbox_gdpt.xml", 512, 512, 64), ("sponza path", "sponza/sponza.xml", 1280, 720, 16)):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator="path")))
    line = []
    for shrink in (0, 35, 40, 45, 65):
        with G.debug_knobs(plan_shrink=shrink):
            best = min(sc.path_render(spp, G.RNG_SAMPLE)[1].render_ms for _ in range(3))
        line.append(f"shrink {shrink or 55}: {w * h * spp / best / 1e3:.1f}")
    print(f"{name} {w}x{h}x{spp}: " + " | ".join(line) + " Msamples/s", flush=True)
