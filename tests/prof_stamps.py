"""Manual profile (not collected by pytest): where the lane machine's wave cycles go, from the diagnostic build with
in-kernel stamps (debug knob "stamps", include/gdpt_debug.h), plus the counting build's SIMT utilisation figures.
    python tests/prof_stamps.py            # cbox 512x512x16 and sponza 1280x720x8"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import gdpt_amd as G

def run(name, xml, film, spp):
    sd = G.parse_scene(xml, film=film)
    sc = G.Scene(sd)
    sc.render(spp, G.RNG_SAMPLE)
    _, st = sc.render(spp, G.RNG_SAMPLE)
    with G.debug_knobs(stamps=1):
        _, sst = sc.render(spp, G.RNG_SAMPLE)
        stamps = G.debug_knobs.stamps()
    # counting build
    shape = (sc.height, sc.width, 3)
    bufs = [np.zeros(shape) for _ in range(5)]
    cs = G.GdptRenderStats(); cs.nodes_visited = 2 ** 64 - 1
    p = G._params(spp, G.RNG_SAMPLE, (0, 0))
    G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs], C.byref(cs)))
    busy, drain = stamps.pop("busy_us"), stamps.pop("drain_us")
    tot = sum(v for k, v in stamps.items() if k != "wave_steps")
    print(f"== {name}: render {st.render_ms:.3f} ms ({st.samples / st.render_ms / 1e3:.0f} Msamples/s), stamped build {sst.render_ms:.3f} ms")
    print("   segment shares of wave cycles: " + ", ".join(f"{k} {100 * v / tot:.1f}%" for k, v in stamps.items() if k != "wave_steps"))
    print(f"   stamped build: queue handed out in {busy:.0f} us, drain of the items in flight {drain:.0f} us")
    ws = stamps["wave_steps"]
    print(f"   wave steps {ws:.0f}; cycles per wave step {tot / ws:.0f}; rays {cs.rays}; rays per wave step {cs.rays / ws:.1f}")
    print(f"   counting build: lane-step utilisation {cs.lane_steps / (64.0 * cs.wave_steps):.3f}; node-loop utilisation {cs.nodes_visited / (64.0 * max(1, cs.wave_node_trips)):.3f}; "
          f"leaf-loop {cs.tris_tested / (64.0 * 4 * max(1, cs.wave_leaf_trips)):.3f} (of 4 slots); nodes/ray {cs.nodes_visited / cs.rays:.2f}, prims/ray {cs.tris_tested / cs.rays:.2f}, "
          f"node trips per wave step {cs.wave_node_trips / cs.wave_steps:.1f}, leaf trips per wave step {cs.wave_leaf_trips / cs.wave_steps:.1f}", flush=True)

run("cbox 512x512x16", os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml"), (0, 0), 16)
run("cbox 512x512x64", os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml"), (0, 0), 64)
run("sponza 1280x720x8", os.path.join(ROOT, "scenes/sponza/sponza.xml"), (1280, 720), 8)
