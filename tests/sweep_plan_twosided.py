"""Manual sweep (not collected by pytest): work-item plan knobs on the two-sided scenes at their configuration's own size (512x512, 64 spp)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name in ("disney_glass", "disney_bsdf"):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, f"disney_bsdf_test/{name}.xml", width=512, height=512, integrator="gradpath")))
    for spp in (16, 64):
        line = []
        for shrink in (0, 35, 40, 45, 50):
            with G.debug_knobs(plan_shrink=shrink):
                best = min(sc.render(spp, G.RNG_SAMPLE)[1].render_ms for _ in range(3))
            line.append(f"shrink {shrink or 55}: {512 * 512 * spp / best / 1e3:.1f}")
        print(f"{name} {spp} spp: " + " | ".join(line) + " Msamples/s", flush=True)
