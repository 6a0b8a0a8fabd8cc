"""Manual timing (not collected by pytest): the per-rank workload of the weak-scaling bench (16*N spp on a band of
512/N rows of the 512x512 cbox film) on one GPU, N = 1, 2, 4, 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G
sc = G.Scene(G.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox_gdpt.xml")))
for n in (1, 2, 4, 8):
    rows = (0, 512 // n)
    best = 1e9
    for i in range(4):
        _, st = sc.render(16 * n, G.RNG_SAMPLE, rows=rows)
        best = min(best, st.render_ms)
    print(f"N={n}: band {rows}, {16 * n} spp: render {best:.3f} ms = {st.samples / best / 1e3:.1f} Msamples/s per GPU", flush=True)
