"""Manual experiment (not collected by pytest): sponza 1280x720 with every bitmap reflectance replaced by a constant, against the
textured scene on the same box — what the texture lookups (fp64 mip chains in HBM) and the uv / footprint work cost."""
import os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
xml = scene_variant(tmp, "sponza/sponza.xml", width=1280, height=720)
text = open(xml).read()
flat = re.sub(r'<ref name="reflectance" id="[^"]+"/>', '<rgb name="reflectance" value="0.6 0.5 0.4"/>', text)
xml2 = xml.replace("_variant.xml", "_notex.xml")
open(xml2, "w").write(flat)
spp = int(os.environ.get("SPP", "16"))
for tag, path in (("textured", xml), ("constant reflectances", xml2)):
    sc = G.Scene(G.parse_scene(path))
    best = 1e9
    for _ in range(3):
        _, st = sc.render(spp, G.RNG_SAMPLE); best = min(best, st.render_ms)
    print(f"sponza {tag}: {best:.2f} ms = {st.samples / best / 1e3:.1f} Msamples/s, rays/sample {st.rays / st.samples:.2f}, bounces/sample {st.bounces / st.samples:.2f}", flush=True)
