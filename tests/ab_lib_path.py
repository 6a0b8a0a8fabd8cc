"""Manual A/B (not collected by pytest): Integrator::Path with two builds of libgdpt.so on the same box, alternating processes.
    python tests/ab_lib_path.py <other.so>"""
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
    for name, rel, w, h, spp in (("cbox", "cbox/cbox_gdpt.xml", 512, 512, 64), ("sponza", "sponza/sponza.xml", 1280, 720, 16),
                                 ("veach_mi", "veach_mi/mi.xml", 768, 512, 64), ("matpreview", "matpreview/matpreview.xml", 512, 512, 32)):
        sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator="path")))
        best = 1e9
        for _ in range(3):
            _, st = sc.path_render(spp, G.RNG_SAMPLE); best = min(best, st.render_ms)
        out[name] = round(st.samples / best / 1e3, 1)
    print("RESULT " + json.dumps(out))
else:
    for rep in range(2):
        for tag, lib in [("tree ", "-")] + [(os.path.basename(o), o) for o in sys.argv[1:]]:
            r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(tag, line[0] if line else r.stderr[-300:], flush=True)
