"""Rank process of tests/test_bench_launch.py (test infrastructure, started by bench.launch_ranks): drives the SAME
sharded step bench.py times — gdpt_amd.sharding.ShardedGradPath — over gloo on CPU tensors, with the CPU oracle standing in
for the three GPU phases (band render, band assembly, Poisson solve), and writes what this rank ends up with."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    xml, spp, out_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gdpt_amd as G
    import oracle_py as O
    from gdpt_amd import sharding
    sd = G.parse_scene(xml)
    W, H = sd.width, sd.height
    osc = O.OracleScene(sd.ptr)

    def render_band(bufs, rows, want_stats):
        got, st = osc.render(spp, G.RNG_SAMPLE, rows=rows, threads=2)
        for k in bufs:
            bufs[k].fill_(-7.0)                                  # poison: rows of other bands must come from the exchange
            bufs[k][rows[0]:rows[1]] = torch.from_numpy(got[k][rows[0]:rows[1]])
        return st

    def assemble(bufs, dst, rows):
        c, cx, cy = O.assemble({k: v.numpy() for k, v in bufs.items()})       # whole-image formula; only the band is kept
        for t, a in zip(dst, (c, cx, cy)):
            t.fill_(-7.0)
            t[rows[0]:rows[1]] = torch.from_numpy(a[rows[0]:rows[1]])

    def solve(c, cx, cy, out, want_stats):
        out.copy_(torch.from_numpy(O.fourier_solve(c.numpy(), cx.numpy(), cy.numpy(), 0.04)))

    bands = None
    if len(sys.argv) > 4 and sys.argv[4] == "weighted":          # cost-balanced bands from a pilot (the oracle's ray counts per tile row)
        costs = [float(osc.render(1, G.RNG_SAMPLE, rows=(t * 16, min(H, t * 16 + 16)), threads=2)[1].rays) for t in range((H + 15) // 16)]
        bands = sharding.bands_weighted(H, world, costs)
    if len(sys.argv) > 4 and sys.argv[4] == "feedback":          # what bench.py does by default: cuts at any row, corrected by (here: made-up) band times
        costs = [float(osc.render(1, G.RNG_SAMPLE, rows=(t * 16, min(H, t * 16 + 16)), threads=2)[1].rays) for t in range((H + 15) // 16)]
        bands = sharding.bands_weighted(H, world, costs, granularity=1)
        mine = torch.tensor([1.0 + 0.37 * rank], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows = sharding.refine_row_costs(sharding.row_costs_from_tiles(H, costs), bands, [float(t.item()) for t in every])
        bands = sharding.bands_from_row_costs(H, world, rows)
    pipe = sharding.ShardedGradPath(dist, world, rank, H, lambda: torch.zeros((H, W, 3), dtype=torch.float64),
                                    render_band, assemble, solve, bands=bands)
    pipe.step()
    pipe.step()                                                  # a second step: scratch reuse, no stale state
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), out=pipe.out.numpy(), c=pipe.c.numpy(), cx=pipe.cx.numpy(), cy=pipe.cy.numpy())
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "rows": list(pipe.rows), "bands": [list(b) for b in pipe.bands], "ok": True}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
