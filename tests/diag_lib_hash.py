"""Manual (not collected by pytest): buffer hashes of the lane machine and the wavefront pipeline from two builds of the library.
    python tests/diag_lib_hash.py <other.so>"""
import os, sys, subprocess, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gdpt_amd as G
    if sys.argv[2] != "-":
        G.LIB_PATH = sys.argv[2]
    from helpers import scene_variant
    import tempfile, numpy as np
    tmp = tempfile.mkdtemp()
    out = {}
    for name, rel, w, h, integ, spp in (("sponza", "sponza/sponza.xml", 200, 112, None, 6), ("cbox", "cbox/cbox_gdpt.xml", 64, 64, None, 6),
                                        ("metal", "disney_bsdf_test/disney_metal.xml", 96, 80, "gradpath", 5)):
        sc = G.Scene(G.parse_scene(scene_variant(tmp, rel, width=w, height=h, integrator=integ)))
        for mode in (0, 1):
            with G.debug_knobs(wavefront=mode):
                b, st = sc.render(spp, G.RNG_SAMPLE)
            out[f"{name}_wf{mode}"] = {k: hashlib.sha1(np.ascontiguousarray(b[k]).tobytes()).hexdigest()[:10] for k in b} | {"rays": st.rays}
    print("RESULT " + json.dumps(out))
else:
    for tag, lib in (("tree ", "-"), ("other", sys.argv[1])):
        r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if not line: print(tag, r.stderr[-400:]); continue
        d = json.loads(line[0][7:])
        for k, v in d.items(): print(tag, k, v, flush=True)
