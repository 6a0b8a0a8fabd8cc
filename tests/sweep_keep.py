"""Manual tuning target (not collected by pytest): render time vs. GDPT_KEEP_FRAC on cbox 512^2x16 and sponza 1280x720x8."""
import os, sys, tempfile, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gdpt_amd as G
    G.debug_knobs.from_env()      # the parent passes GDPT_KEEP_FRAC / GDPT_SEARCH_FRAC; the library itself reads no environment
    from helpers import scene_variant
    out = {}
    for name, xml, spp in (("cbox", os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml"), 16),
                           ("sponza", scene_variant(tempfile.mkdtemp(), "sponza/sponza.xml", width=1280, height=720), 8)):
        sc = G.Scene(G.parse_scene(xml))
        best = 1e9
        for i in range(4):
            _, st = sc.render(spp, G.RNG_SAMPLE)
            best = min(best, st.render_ms)
        out[name] = (best, st.samples / best / 1e3)
    print(json.dumps(out))
else:
    for kf in sys.argv[1:]:
        k, sf, ww = (kf.split(":") + ["0", "0"])[:3]
        env = dict(os.environ, GDPT_KEEP_FRAC=k, GDPT_SEARCH_FRAC=sf, GDPT_LDS_WW=ww)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("keep_frac", kf, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
