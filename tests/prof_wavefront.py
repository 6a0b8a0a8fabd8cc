"""Manual A/B (not collected by pytest): lane machine vs wavefront pipeline on scenes walked from HBM; images must be
bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gdpt_amd as G
from helpers import scene_variant
import tempfile
tmp = tempfile.mkdtemp()
cases = [("sponza 320x180x4", os.path.join(ROOT, "scenes/sponza/sponza.xml"), (320, 180), None, 4),
         ("sponza 1280x720x8", os.path.join(ROOT, "scenes/sponza/sponza.xml"), (1280, 720), None, 8),
         ("sponza 1280x720x64", os.path.join(ROOT, "scenes/sponza/sponza.xml"), (1280, 720), None, 64),
         ("disney_metal 512x512x64", scene_variant(tmp, "disney_bsdf_test/disney_metal.xml", integrator="gradpath"), (512, 512), "gradpath", 64)]
only = sys.argv[1:] 
for name, xml, film, integ, spp in cases:
    if only and not any(o in name for o in only):
        continue
    sd = G.parse_scene(xml, film=film)
    sc = G.Scene(sd)
    res = {}
    for mode in (0, 1):
        with G.debug_knobs(wavefront=mode):
            best = 1e9
            for _ in range(3):
                bufs, st = sc.render(spp, G.RNG_SAMPLE)
                best = min(best, st.render_ms)
        res[mode] = (bufs, st, best)
    same = all(np.array_equal(res[0][0][k], res[1][0][k], equal_nan=True) for k in res[0][0])
    a, b = res[0], res[1]
    print(f"{name}: lane machine {a[2]:.2f} ms ({a[1].samples / a[2] / 1e3:.0f} Msamples/s) | wavefront {b[2]:.2f} ms ({b[1].samples / b[2] / 1e3:.0f} Msamples/s) | "
          f"identical {same} | rays {a[1].rays} {b[1].rays} bounces {a[1].bounces} {b[1].bounces}", flush=True)
    if not same:
        for k in res[0][0]:
            d = np.abs(res[0][0][k] - res[1][0][k]); print("   ", k, "max abs diff", np.nanmax(d), "differing px", int((d > 0).any(axis=2).sum()))
