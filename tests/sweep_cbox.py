"""Manual sweep (not collected by pytest) of the lane machine's knobs on cbox 512x512x16 after a kernel change:
leaf size of the SAH build, trace-exit fractions, persistent blocks per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G
xml = os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml")
def run(**kn):
    G.debug_knobs.reset(); G.debug_knobs.set(**kn)
    sc = G.Scene(G.parse_scene(xml))
    best = 1e9
    for _ in range(5):
        _, st = sc.render(16, G.RNG_SAMPLE); best = min(best, st.render_ms)
    print(f"{kn}: {best:.3f} ms = {st.samples / best / 1e3:.1f} Msamples/s", flush=True)
run()
for lm in (2, 3, 4, 6, 8):
    for lf in (0.8, 1.0):
        run(bvh_leaf_max=lm, bvh_leaf_factor=lf)
for kf in (32, 48, 64, 80, 96, 128):
    run(keep_frac=kf)
for sf in (64, 96, 112, 128, 160, 200):
    run(search_frac=sf)
run()
