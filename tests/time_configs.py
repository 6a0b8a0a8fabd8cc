"""Manual timing of BASELINE.json's other configurations on one GPU (not collected by pytest)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
cases = [("C2 cbox 512x512 16spp", "cbox/cbox_gdpt.xml", 512, 512, None, 16),
         ("C2' cbox 512x512 256spp", "cbox/cbox_gdpt.xml", 512, 512, None, 256),
         ("C3 cbox 1024x1024 256spp (1 GPU)", "cbox/cbox_gdpt.xml", 1024, 1024, None, 256),
         ("C4 sponza 1280x720 64spp", "sponza/sponza.xml", 1280, 720, None, 64),
         ("C5 disney_diffuse 512x512 64spp", "disney_bsdf_test/disney_diffuse.xml", 512, 512, "gradpath", 64),
         ("C5 disney_metal 512x512 64spp", "disney_bsdf_test/disney_metal.xml", 512, 512, "gradpath", 64),
         ("C5 disney_clearcoat 512x512 64spp", "disney_bsdf_test/disney_clearcoat.xml", 512, 512, "gradpath", 64),
         ("C5 disney_sheen 512x512 64spp", "disney_bsdf_test/disney_sheen.xml", 512, 512, "gradpath", 64),
         ("C5 disney_glass 512x512 64spp (two-sided lane machine)", "disney_bsdf_test/disney_glass.xml", 512, 512, "gradpath", 64),
         ("C5 disney_bsdf 512x512 64spp (two-sided lane machine)", "disney_bsdf_test/disney_bsdf.xml", 512, 512, "gradpath", 64)]
if len(sys.argv) > 1 and sys.argv[1] == "gradpath":
    PATH_CASES = False
for name, rel, w, h, integ, spp in cases:
    xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
    sc = G.Scene(G.parse_scene(xml))
    out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
    out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
    print(f"{name}: render {rs.render_ms:.2f} ms = {rs.samples / rs.render_ms / 1e3:.1f} Msamples/s, rays/sample {rs.rays / rs.samples:.2f}, "
          f"bounces/sample {rs.bounces / rs.samples:.2f}, nonfinite {rs.nonfinite_samples}, poisson {ps.solve_ms:.3f} ms", flush=True)
# Integrator::Path
if len(sys.argv) > 1 and sys.argv[1] == "gradpath":
    sys.exit(0)
for name, rel, w, h, spp in (("P cbox path 512x512 64spp", "cbox/cbox_gdpt.xml", 512, 512, 64), ("P sponza path 1280x720 16spp", "sponza/sponza.xml", 1280, 720, 16),
                               ("P veach_mi direct 768x512 64spp", "veach_mi/mi.xml", 768, 512, 64),
                               ("P matpreview (envmap, roughdielectric) 512x512 32spp", "matpreview/matpreview.xml", 512, 512, 32)):
    xml = scene_variant(tmp, rel, width=w, height=h, integrator="path")
    sc = G.Scene(G.parse_scene(xml))
    img, st = sc.path_render(spp, G.RNG_SAMPLE)
    img, st = sc.path_render(spp, G.RNG_SAMPLE)
    print(f"{name}: render {st.render_ms:.2f} ms = {st.samples / st.render_ms / 1e3:.1f} Msamples/s, rays/sample {st.rays / st.samples:.2f}, "
          f"bounces/sample {st.bounces / st.samples:.2f}, nonfinite {st.nonfinite_samples}", flush=True)
# GDPT_SHIFT_RECONNECT: straight-loop kernel, HBM scene
for name, rel, w, h, integ, spp in (("R cbox reconnect 512x512 16spp", "cbox/cbox_gdpt.xml", 512, 512, None, 16), ("R cbox reconnect 512x512 256spp", "cbox/cbox_gdpt.xml", 512, 512, None, 256),
                                    ("R sponza reconnect 1280x720 16spp", "sponza/sponza.xml", 1280, 720, None, 16), ("R veach_mi reconnect 768x512 64spp", "veach_mi/mi.xml", 768, 512, "gradpath", 64)):
    xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
    sc = G.Scene(G.parse_scene(xml))
    for _ in range(2):
        out, bufs, rs, ps = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True, shift=G.SHIFT_RECONNECT)
    print(f"{name}: render {rs.render_ms:.2f} ms = {rs.samples / rs.render_ms / 1e3:.1f} Msamples/s, rays/sample {rs.rays / rs.samples:.2f}, "
          f"bounces/sample {rs.bounces / rs.samples:.2f}, nonfinite {rs.nonfinite_samples}", flush=True)
