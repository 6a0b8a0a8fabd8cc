"""Manual exploration (not collected by pytest): GPU renders against statistics of renders shipped by the reference."""
import sys, json, numpy as np, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
d = json.load(open(os.path.join(ROOT, 'tests/golden/ref_images.json')))['reference_renders']
tmp = tempfile.mkdtemp()
def cmp(name, img):
    v = d[name]; ref = np.array(v['block_mean_32']); h, w, _ = img.shape; bs = 32; hh, ww = h // bs * bs, w // bs * bs
    th = img[:hh, :ww].reshape(hh // bs, bs, ww // bs, bs, 3).mean(axis=(1, 3))
    print(f"   vs {name}: mean ratio {img.mean(axis=(0,1)) / np.array(v['mean'])}, thumb relL2 {np.linalg.norm(th - ref) / np.linalg.norm(ref):.4f}")
for rel, integ, spp, refs in (("disney_bsdf_test/disney_glass.xml", "path", 64, ["extra_images/disney_glass_eta_1.5.exr"]),
                              ("disney_bsdf_test/disney_sheen.xml", "path", 64, ["extra_images/disney_sheen_test_1.0.exr"]),
                              ("sponza/sponza.xml", "path", 256, ["gdpt_renders/sponza_regular_path_trace/sp_256.exr", "gdpt_renders/sponza_reg_path_non_nee/sp_256.exr"]),
                              ("sponza/sponza.xml", "gradpath", 256, ["gdpt_renders/sponza_grad_path_trace/s_gp_256.exr", "gdpt_renders/sponza.exr"])):
    xml = scene_variant(tmp, rel, integrator=integ)
    sc = G.Scene(G.parse_scene(xml))
    if integ == "path":
        img, st = sc.path_render(spp, G.RNG_SAMPLE)
    else:
        img = sc.gradient_path_render(spp, G.RNG_SAMPLE, alpha=0.04)
    print(rel, integ, spp, img.shape, img.mean(axis=(0, 1)))
    for r in refs:
        cmp(r, img)
