import os, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from helpers import scene_variant
tmp = tempfile.mkdtemp()
xml = scene_variant(tmp, "cbox/cbox_gdpt.xml", width=128, height=128)
sc = G.Scene(G.parse_scene(xml))
for spp in (4, 16, 64):
    a, sa = sc.render(spp, G.RNG_SAMPLE)
    b, sb = sc.render(spp, G.RNG_SAMPLE, shift=G.SHIFT_RECONNECT)
    A, B = np.asarray(a["img"]), np.asarray(b["img"])
    d = np.abs(A - B).max(axis=2)
    print(spp, "max diff", d.max(), "num differing pixels", int((d > 1e-12).sum()), "of", d.size, "bounces", sa.bounces, sb.bounces, "means", A.mean(), B.mean())
    ys, xs = np.nonzero(d > 1e-12)
    for y, x in list(zip(ys, xs))[:5]:
        print("  ", y, x, A[y, x], B[y, x])

for spp in (16,):
    o1, b1, r1, _ = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True, shift=G.SHIFT_RECONNECT)
    o2, b2, r2, _ = sc.gradient_path_render(spp, G.RNG_SAMPLE, return_buffers=True)
    a, sa = sc.render(spp, G.RNG_SAMPLE)
    print("gpr reconnect vs render parity", np.abs(np.asarray(b1["img"]) - np.asarray(a["img"])).max(), "gpr parity vs render parity", np.abs(np.asarray(b2["img"]) - np.asarray(a["img"])).max(), r1.samples, r2.samples, sa.samples)
