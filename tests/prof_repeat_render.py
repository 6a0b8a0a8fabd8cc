"""Manual: renders one scene a few times, as a target for rocprofv3 --kernel-trace / --pmc passes on a single kernel, e.g.
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d gpurun_out/p -o run --output-format csv -- python3 tests/prof_repeat_render.py cbox 512 512 16 20
(not collected by pytest). PC sampling is not used: the GPU pool refuses that profiler mode."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
rel = {"cbox": "cbox/cbox_gdpt.xml", "sponza": "sponza/sponza.xml", "bsdf": "disney_bsdf_test/disney_bsdf.xml",
       "glass": "disney_bsdf_test/disney_glass.xml"}[sys.argv[1]]
w, h, spp, reps = (int(a) for a in sys.argv[2:6])
xml = scene_variant(tempfile.mkdtemp(), rel, width=w, height=h, integrator="gradpath" if "disney" in rel else None)
sc = G.Scene(G.parse_scene(xml))
for _ in range(reps):
    out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
print(f"render {rs.render_ms:.3f} ms", flush=True)
