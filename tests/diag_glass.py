import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, tempfile
import gdpt_amd as G
import oracle_py as O
from helpers import scene_variant
xml = scene_variant(tempfile.mkdtemp(), "disney_bsdf_test/disney_glass.xml", width=512, height=512, integrator="gradpath")
sd = G.parse_scene(xml); sc = G.Scene(sd)
bufs, st = sc.render(64, G.RNG_SAMPLE)
print("nonfinite samples", st.nonfinite_samples)
for k, v in bufs.items():
    bad = ~np.isfinite(v)
    print(k, "nonfinite px", int(bad.any(axis=2).sum()), np.argwhere(bad.any(axis=2))[:5].tolist())
rows = sorted(set(np.argwhere(~np.isfinite(bufs["cx0"]).all(axis=2) | ~np.isfinite(bufs["cx1"]).all(axis=2) | ~np.isfinite(bufs["cy0"]).all(axis=2) | ~np.isfinite(bufs["cy1"]).all(axis=2))[:, 0].tolist()))
print("rows with nonfinite", rows[:10])
if rows:
    r0 = rows[0] // 16 * 16
    ob, ost = O.OracleScene(sd.ptr, use_bvh=True).render(64, G.RNG_SAMPLE, rows=(r0, r0 + 16), threads=16)
    for k in bufs:
        a, b = bufs[k][r0:r0 + 16], ob[k][r0:r0 + 16]
        print(k, "oracle nonfinite px", int((~np.isfinite(b)).any(axis=2).sum()), "same mask", np.array_equal(np.isfinite(a), np.isfinite(b)))
    print("oracle nonfinite samples in band", ost.nonfinite_samples if hasattr(ost, "nonfinite_samples") else "n/a")
