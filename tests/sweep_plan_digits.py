"""Manual sweep (not collected by pytest): explicit work-item plans (knob plan_digits) on the headline film, cbox 512x512 at 16 spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G
sc = G.Scene(G.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml")))
plans = [0, 8521, 552211, 642211, 543211, 633211, 4432111, 4422211, 5322211, 44221111, 33222211, 6322111, 7321111, 5521111, 55111111, 442211, 0]
res = {}
for rep in range(3):
    for d in plans:
        with G.debug_knobs(plan_digits=d):
            sc.render(16, G.RNG_SAMPLE)
            t = min(sc.render(16, G.RNG_SAMPLE)[1].render_ms for _ in range(4))
        res.setdefault(d, []).append(t)
for d in dict.fromkeys(plans):
    print(f"plan {d or 'default (552211)'}: best {min(res[d]):.3f} ms, runs " + " ".join(f"{t:.3f}" for t in res[d]), flush=True)
