"""Manual A/B (not collected by pytest), same process and box: two-sided lane machine built for the scene's material set vs
the kernel with the full material switch (debug knob full_material_switch)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name in ("disney_glass", "disney_bsdf"):
    xml = scene_variant(tmp, f"disney_bsdf_test/{name}.xml", width=512, height=512, integrator="gradpath")
    sc = G.Scene(G.parse_scene(xml))
    res = {}
    for rep in range(3):
        for mode in (0, 1):
            with G.debug_knobs(full_material_switch=mode):
                _, st = sc.render(64, G.RNG_SAMPLE)
            res.setdefault(mode, []).append(st.render_ms)
    print(f"{name}: material-set kernel {min(res[0]):.2f} ms ({st.samples / min(res[0]) / 1e3:.1f} Msamples/s) | full switch {min(res[1]):.2f} ms "
          f"({st.samples / min(res[1]) / 1e3:.1f} Msamples/s)", flush=True)
