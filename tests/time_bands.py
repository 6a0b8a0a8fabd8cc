"""Manual timing (not collected by pytest): the cost of EVERY row band of the sharded tile loop, each rendered alone on one
GPU, for N = 2, 4, 8 bands — what a strong-scaling run on N devices would see as its per-rank render times (the render
has no collective inside; exchange and solve are priced separately in DESIGN.md 5).

  C2'  cbox 512x512, 256 spp (the north-star target workload)      C3  cbox 1024x1024, 256 spp (BASELINE configs[2])

For each N: per-band ms, max / mean, and the strong-scaling efficiency the imbalance alone implies
(one-GPU time / (N * slowest band)). Output: profiles/r03_band_costs.txt (via gpurun_out/).
Work items are cut for the largest band of the sharding (GdptRenderParams.plan_rows), as the multi-device hosts do; bands
of equal tile-row count and bands of equal pilot cost (Scene.tile_row_costs -> sharding.bands_weighted) side by side."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from gdpt_amd import sharding
from helpers import scene_variant
import tempfile
tmp = tempfile.mkdtemp()
for name, size, spp in (("C2' cbox 512x512x256", 512, 256), ("C3 cbox 1024x1024x256", 1024, 256), ("weak-scaling unit: cbox 512x512, 16 spp per GPU", 512, 16)):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, "cbox/cbox_gdpt.xml", width=size, height=size)))
    weak = spp == 16
    def cost(rows, plan_rows, spp_=spp):
        best = 1e9
        for _ in range(3):
            _, st = sc.render(spp_, G.RNG_SAMPLE, rows=rows, plan_rows=plan_rows)
            best = min(best, st.render_ms)
        return best
    whole = cost((0, size), 0)
    print(f"{name}: one GPU, whole film: {whole:.2f} ms = {size * size * spp / whole / 1e3:.0f} Msamples/s", flush=True)
    pilot = sc.tile_row_costs()
    print("   pilot (rays per tile row at 1 spp, relative to the mean): " + " ".join(f"{c / (sum(pilot) / len(pilot)):.2f}" for c in pilot), flush=True)
    for n in (2, 4, 8):
        k = n if weak else 1                       # weak scaling: 16 spp per GPU = 16 N spp on the band

        def run(label, bands, with_film_plan=False):
            plan = max(b[1] - b[0] for b in bands)
            ms = [cost(b, plan, spp * k) for b in bands]
            film_plan = [cost(b, 0, spp * k) for b in bands[:1]][0] if with_film_plan else None
            mean = sum(ms) / n
            print(f"   N={n} {label:34s} rows " + " ".join(str(b[1] - b[0]) for b in bands) + ": " + " ".join(f"{m:.2f}" for m in ms) +
                  f" ms | max/mean {max(ms) / mean:.3f} | efficiency of the render (one-GPU time / ({'1' if weak else 'N'} x slowest band)) {whole / ((1 if weak else n) * max(ms)):.3f}" +
                  (f" | band 0 with the work items cut for the whole film (round 2): {film_plan:.2f} ms" if film_plan else ""), flush=True)
            return ms
        run("equal tile rows", sharding.all_bands(size, n), True)
        run("equal pilot cost, whole tile rows", sharding.bands_weighted(size, n, pilot, granularity=16))
        bands = sharding.bands_weighted(size, n, pilot, granularity=1)
        ms = run("equal pilot cost, any row", bands)
        rows = sharding.row_costs_from_tiles(size, pilot)
        for it in (1, 2):                              # what bench.py does during its warm-up steps: the frame's own clock corrects the pilot
            rows = sharding.refine_row_costs(rows, bands, ms)
            bands = sharding.bands_from_row_costs(size, n, rows)
            ms = run(f"after {it} round(s) of time feedback", bands)
