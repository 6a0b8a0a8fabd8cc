"""Manual check of GDPT_SHIFT_RECONNECT on one GPU (not collected by pytest)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
for name, rel, w, h, integ in (("cbox", "cbox/cbox_gdpt.xml", 512, 512, None), ("veach_mi", "veach_mi/mi.xml", 384, 256, "gradpath"), ("sponza", "sponza/sponza.xml", 320, 180, None)):
    xml = scene_variant(tmp, rel, width=w, height=h, integrator=integ)
    sc = G.Scene(G.parse_scene(xml))
    ref, st = sc.render(8192, G.RNG_SAMPLE)                      # converged primal (parity mode)
    I = np.asarray(ref["img"])
    for spp in (16, 64):
        out_r, b_r, rs_r, _ = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True, shift=G.SHIFT_RECONNECT)
        out_p, b_p, rs_p, _ = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
        c, cx, cy = G.assemble(b_r) if hasattr(G, "assemble") else (None, None, None)
        prim_same = np.abs(np.asarray(b_r["img"]) - np.asarray(b_p["img"])).max()
        fdx = np.zeros_like(I); fdx[:, 1:] = I[:, 1:] - I[:, :-1]
        fdy = np.zeros_like(I); fdy[1:] = I[1:] - I[:-1]
        gx = np.asarray(b_r["cx0"]).copy(); gx[:, 1:] += np.asarray(b_r["cx1"])[:, :-1]
        gy = np.asarray(b_r["cy0"]).copy(); gy[1:] += np.asarray(b_r["cy1"])[:-1]
        m = lambda a: float(np.sqrt(np.mean(a ** 2)))
        print(f"{name} {spp} spp: primal max|diff| vs parity mode {prim_same:.2e}; "
              f"rmse primal {m(np.asarray(b_r['img']) - I):.4f}, reconstruction {m(np.asarray(out_r) - I):.4f} (parity-mode reconstruction {m(np.asarray(out_p) - I):.4f}); "
              f"grad x: mean est {gx[:, 1:].mean():.5f} vs fd {fdx[:, 1:].mean():.5f}, rmse(gx - fd) {m(gx[:, 1:] - fdx[:, 1:]):.4f} vs rmse(fd of noisy primal) {m((np.asarray(b_r['img'])[:, 1:] - np.asarray(b_r['img'])[:, :-1]) - fdx[:, 1:]):.4f}; "
              f"grad y rmse {m(gy[1:] - fdy[1:]):.4f}; render {rs_r.render_ms:.2f} ms vs {rs_p.render_ms:.2f} ms, rays/sample {rs_r.rays / rs_r.samples:.2f}, nonfinite {rs_r.nonfinite_samples}", flush=True)
