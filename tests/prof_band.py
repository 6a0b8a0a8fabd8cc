"""Manual profile (not collected by pytest): a 64-row band of the cbox 512x512 film at 128 spp (what one of eight weak-scaling ranks
renders) against the whole film at 16 spp (the same number of samples): stamp shares, drain, throughput."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G

sc = G.Scene(G.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml")))
for name, spp, rows, plan in (("whole film, 16 spp", 16, (0, 0), 0), ("rows 192..256, 128 spp, work items cut for a 64-row band", 128, (192, 256), 64),
                              ("rows 192..256, 128 spp, 2^3 equal chunks", 128, (192, 256), -3), ("rows 192..256, 128 spp, 2^5 equal chunks", 128, (192, 256), -5),
                              ("rows 128..256, 64 spp, work items cut for a 128-row band", 64, (128, 256), 128)):
    knobs = {}
    if plan < 0:
        knobs["log2k"] = -plan; plan = 64
    with G.debug_knobs(**knobs):
        sc.render(spp, G.RNG_SAMPLE, rows=rows, plan_rows=plan)
        best = min(sc.render(spp, G.RNG_SAMPLE, rows=rows, plan_rows=plan)[1].render_ms for _ in range(5))
        _, st = sc.render(spp, G.RNG_SAMPLE, rows=rows, plan_rows=plan)
    with G.debug_knobs(stamps=1, **knobs):
        _, sst = sc.render(spp, G.RNG_SAMPLE, rows=rows, plan_rows=plan)
        stamps = G.debug_knobs.stamps()
    busy, drain = stamps.pop("busy_us"), stamps.pop("drain_us")
    ws = stamps.pop("wave_steps")
    tot = sum(stamps.values())
    print(f"== {name}: render {best:.3f} ms ({st.samples / best / 1e3:.0f} Msamples/s), stamped build {sst.render_ms:.3f} ms")
    print("   segment shares of wave cycles: " + ", ".join(f"{k} {100 * v / tot:.1f}%" for k, v in stamps.items()))
    print(f"   stamped build: queue handed out in {busy:.0f} us, drain of the items in flight {drain:.0f} us; wave steps {ws:.0f}; cycles per wave step {tot / ws:.0f}", flush=True)
