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
    modes = [("lane machine", dict(wavefront=0)), ("wavefront unsorted", dict(wavefront=1, wf_sort=0)),
             ("wavefront octant-major", dict(wavefront=1, wf_sort=1)), ("wavefront cell-major", dict(wavefront=1, wf_sort=2))]
    if os.environ.get("WF_ONLY"):
        modes = [m for m in modes if m[0].endswith(os.environ["WF_ONLY"])]
    for label, knobs in modes:
        with G.debug_knobs(**knobs):
            best = 1e9
            for _ in range(3):
                bufs, st = sc.render(spp, G.RNG_SAMPLE)
                best = min(best, st.render_ms)
        res[label] = (bufs, st, best)
    ref = res[modes[0][0]]
    for label, (bufs, st, best) in res.items():
        same = all(np.array_equal(ref[0][k], bufs[k], equal_nan=True) for k in bufs)
        print(f"{name}: {label:24s} {best:8.2f} ms ({st.samples / best / 1e3:6.0f} Msamples/s, {st.rays / best / 1e6:5.2f} Grays/s) | identical {same} | rays {st.rays} bounces {st.bounces}", flush=True)
        if not same:
            for k in bufs:
                d = np.abs(ref[0][k] - bufs[k]); print("   ", k, "max abs diff", np.nanmax(d), "differing px", int((d > 0).any(axis=2).sum()))
