"""Manual sweep (not collected by pytest): replay iterations per wave step (knob replay_per_step) on the two-sided scenes, 512x512, 64 spp."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name in ("disney_glass", "disney_bsdf"):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, f"disney_bsdf_test/{name}.xml", width=512, height=512, integrator="gradpath")))
    line = []
    for budget in (0, 1, 2, 3, 6, 8):
        with G.debug_knobs(replay_per_step=budget):
            best = min(sc.render(64, G.RNG_SAMPLE)[1].render_ms for _ in range(3))
        line.append(f"budget {budget or '4 (default)'}: {512 * 512 * 64 / best / 1e3:.1f}")
    print(f"{name} 64 spp: " + " | ".join(line) + " Msamples/s", flush=True)
