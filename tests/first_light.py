"""Manual GPU smoke script used during bring-up (not collected by pytest)."""
import sys, time, tempfile
sys.path.insert(0, __file__.rsplit('/', 2)[0] + '/tests')
from helpers import *
import gdpt_amd as G
import oracle_py as O

tmp = tempfile.mkdtemp()
for (w, h, spp, scheme) in [(64, 64, 8, G.RNG_SAMPLE), (48, 32, 3, G.RNG_TILE), (128, 128, 16, G.RNG_SAMPLE)]:
    xml = scene_variant(tmp, "cbox/cbox_gdpt.xml", width=w, height=h)
    sd = G.parse_scene(xml)
    sc = G.Scene(sd)
    print(sc.info())
    t = time.time(); bufs, st = sc.render(spp, scheme); dt = time.time() - t
    print("gpu", w, h, spp, scheme, "ms", st.render_ms, "rays", st.rays, "bounces", st.bounces, "nonfinite", st.nonfinite_samples, "wall", dt)
    osc = O.OracleScene(sd.ptr)
    ob, ost = osc.render(spp, scheme, threads=8)
    print("oracle rays", ost.rays, "bounces", ost.bounces, "sec", ost.seconds)
    for k in bufs:
        print("  ", k, "rel_l2", rel_l2(bufs[k], ob[k]), "max", np.abs(bufs[k]).max())
    c, cx, cy = O.assemble(ob)
    ref = O.fourier_solve(c, cx, cy, 0.04)
    out, st2 = G.fourierSolve(w, h, c, cx, cy, 0.04, return_stats=True)
    print("   poisson iters", st2.iterations, "res", st2.rel_residual, "ms", st2.solve_ms, "rel_l2 vs dct", rel_l2(out, ref))
    full, b2, rs, ps = sc.gradient_path_render(spp, scheme, return_buffers=True)
    print("   pipeline rel_l2", rel_l2(full, ref), "iters", ps.iterations)

xml = scene_variant(tmp, "cbox/cbox_gdpt.xml")
sd = G.parse_scene(xml); sc = G.Scene(sd)
for i in range(3):
    bufs, st = sc.render(16, G.RNG_SAMPLE)
    print("512x512x16: ms", st.render_ms, "Msamples/s", st.samples / st.render_ms / 1e3, "rays", st.rays, "bounces", st.bounces)
out, b2, rs, ps = sc.gradient_path_render(16, G.RNG_SAMPLE, return_buffers=True)
print("poisson 512: iters", ps.iterations, "ms", ps.solve_ms, "res", ps.rel_residual)
c, cx, cy = O.assemble(b2)
ref = O.fourier_solve(c, cx, cy, 0.04)
print("full-size pipeline vs scipy DCT on the GPU buffers:", rel_l2(out, ref))
