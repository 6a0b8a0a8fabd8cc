"""Manual A/B (not collected by pytest): two builds of libgdpt.so on the same box, alternating processes.
    python tests/ab_lib.py <other.so> [<another.so> ...]     (each line: Msamples/s per scene; sponza also nodes and primitives per ray)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gdpt_amd as G
    if sys.argv[2] != "-":
        G.LIB_PATH = sys.argv[2]
    from helpers import scene_variant
    import tempfile
    tmp = tempfile.mkdtemp()
    out = {}
    for name, rel, w, h, integ, spp in (("cbox", "cbox/cbox_gdpt.xml", 512, 512, None, 16), ("sponza", "sponza/sponza.xml", 1280, 720, None, 16),
                                        ("metal", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 64), ("diffuse", "disney_bsdf_test/disney_diffuse.xml", 512, 512, "gradpath", 64),
                                        ("glass", "disney_bsdf_test/disney_glass.xml", 512, 512, "gradpath", 32), ("bsdf", "disney_bsdf_test/disney_bsdf.xml", 512, 512, "gradpath", 32)):
        sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator=integ)))
        best = 1e9
        for _ in range(3):
            _, st = sc.render(spp, G.RNG_SAMPLE); best = min(best, st.render_ms)
        out[name] = round(st.samples / best / 1e3, 1)
        if name in ("sponza", "metal"):
            import ctypes as C, numpy as np
            cs = G.GdptRenderStats(); cs.nodes_visited = 2 ** 64 - 1
            p = G._params(spp, G.RNG_SAMPLE, (0, 0))
            b = {k: np.zeros((h, w, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
            G._check(G.lib().gdpt_render(sc.handle, C.byref(p), *[b[k].ctypes.data_as(C.POINTER(C.c_double)) for k in b], C.byref(cs)))
            out[name + "_nodes/ray"] = round(cs.nodes_visited / cs.rays, 2); out[name + "_prims/ray"] = round(cs.tris_tested / cs.rays, 2)
            out[name + "_trips/step"] = round((cs.wave_node_trips + cs.wave_leaf_trips) / max(1, cs.wave_steps), 1)
    print("RESULT " + json.dumps(out))
else:
    others = sys.argv[1:]
    for rep in range(2):
        for tag, lib in [("tree ", "-")] + [(os.path.basename(o), o) for o in others]:
            r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(tag, line[0] if line else r.stderr[-300:], flush=True)
