"""Manual timing (not collected by pytest): the cost of EVERY row band of the sharded tile loop, each rendered alone on one
GPU, for N = 2, 4, 8 bands — what a strong-scaling run on N devices would see as its per-rank render times (the render
has no collective inside; exchange and solve are priced separately in DESIGN.md 5).

  C2'  cbox 512x512, 256 spp (the north-star target workload)      C3  cbox 1024x1024, 256 spp (BASELINE configs[2])

For each N: per-band ms, max / mean, and the strong-scaling efficiency the imbalance alone implies
(one-GPU time / (N * slowest band)). Output: profiles/r03_band_costs.txt (via gpurun_out/).
usage: time_bands.py [--balanced]      --balanced: bands from sharding.band_rows_weighted (cost-balanced), if present"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gdpt_amd as G
from gdpt_amd import sharding
from helpers import scene_variant
import tempfile
tmp = tempfile.mkdtemp()
for name, size, spp in (("C2' cbox 512x512x256", 512, 256), ("C3 cbox 1024x1024x256", 1024, 256)):
    sc = G.Scene(G.parse_scene(scene_variant(tmp, "cbox/cbox_gdpt.xml", width=size, height=size)))
    def cost(rows):
        best = 1e9
        for _ in range(3):
            _, st = sc.render(spp, G.RNG_SAMPLE, rows=rows)
            best = min(best, st.render_ms)
        return best
    whole = cost((0, size))
    print(f"{name}: one GPU, whole film: {whole:.2f} ms = {size * size * spp / whole / 1e3:.0f} Msamples/s", flush=True)
    # per tile row (16 pixel rows): the finest unit a band boundary can move by
    tile_rows = size // 16
    per_tile_row = [cost((t * 16, (t + 1) * 16)) for t in range(tile_rows)] if size == 512 else None
    if per_tile_row:
        print("   tile-row costs (ms): " + " ".join(f"{c:.2f}" for c in per_tile_row), flush=True)
    for n in (2, 4, 8):
        bands = sharding.all_bands(size, n)
        ms = [cost(b) for b in bands]
        mean = sum(ms) / n
        print(f"   N={n}: bands of {bands[0][1] - bands[0][0]} rows: " + " ".join(f"{m:.2f}" for m in ms) +
              f" ms | max/mean {max(ms) / mean:.3f} | sum/whole {sum(ms) / whole:.3f} | strong-scaling efficiency of the render {whole / (n * max(ms)):.3f}", flush=True)
