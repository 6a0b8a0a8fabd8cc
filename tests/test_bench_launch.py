"""`python bench.py --gpus N` as typed: the launcher starts N fresh rank processes, relays rank 0's line and fails
loudly when a rank fails or the node has too few GPUs. The ranks here run bench.py's own sharded step
(gdpt_amd.sharding.ShardedGradPath: band render -> cy1 halo -> band assembly -> packed all-gather -> solve) over gloo, the
CPU oracle standing in for the GPU phases (tests/_rank_child.py); every rank must end with the single-process image."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, scene_variant

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("world,height,bands", [(2, 64, "equal"), (3, 80, "equal"), (3, 112, "weighted"), (3, 112, "feedback")])
def test_launcher_runs_the_sharded_step_on_every_rank(G, O, scene_tmp, tmp_path, world, height, bands, capfd):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=24, height=height)
    sd = G.parse_scene(xml)
    whole, _ = O.OracleScene(sd.ptr).render(2, G.RNG_SAMPLE, threads=4)
    c, cx, cy = O.assemble(whole)
    want = O.fourier_solve(c, cx, cy, 0.04)
    rc = bench.launch_ranks(world, [os.path.join(ROOT, "tests", "_rank_child.py"), xml, "2", str(tmp_path), bands], timeout=300)
    assert rc == 0
    line = json.loads([l for l in capfd.readouterr().out.splitlines() if l.startswith("{")][-1])      # rank 0's line came through the parent
    assert line["world"] == world
    if bands == "weighted":               # the cbox film is cheaper at the top (ceiling, light) than in the middle: the bands are not equal
        from gdpt_amd import sharding
        assert line["bands"] != [list(b) for b in sharding.all_bands(height, world)] and line["bands"][-1][1] == height
    if bands == "feedback":               # cuts at any row (halo row, band assembly and per-band broadcasts on bands that are no whole tile rows)
        assert any(b[0] % 16 for b in line["bands"][1:]) and line["bands"][-1][1] == height
        assert line["bands"][0][1] - line["bands"][0][0] > line["bands"][2][1] - line["bands"][2][0]      # the "slow" last rank got fewer rows
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for name, w in (("c", c), ("cx", cx), ("cy", cy), ("out", want)):
            assert np.array_equal(got[name], w), (r, name)


def test_launcher_reports_a_failing_rank(tmp_path):
    bad = tmp_path / "bad_rank.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(30)\n")
    rc = bench.launch_ranks(2, [str(bad)], timeout=60)
    assert rc == 7                                   # rank 0 (still sleeping) was ended, the failure is the exit status


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 2)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert f"needs {n + 2} GPUs" in r.stderr and "visible" in r.stderr
    assert not r.stdout.strip()                      # no JSON line pretending to be a measurement


def test_one_rank_takes_the_fused_assembly_and_solve_and_several_ranks_do_not():
    """ShardedGradPath(assemble_solve=...): with one rank the whole film is local, so assembly and solve go through the one fused
    call (gdpt_assemble_solve_device in bench.py) and the phase hooks still fire in order; with several ranks the bands have to
    be gathered between the two, so the hook is ignored. Host logic only: the phases are stand-ins."""
    import torch
    from gdpt_amd import sharding
    calls = []

    def render_band(bufs, rows, want_stats):
        calls.append(("render", rows)); bufs["img"][rows[0]:rows[1]] = 1.0

    def assemble(bufs, dst, rows):
        calls.append(("assemble", rows))

    def solve(c, cx, cy, out, want_stats):
        calls.append(("solve",)); return "separate"

    def assemble_solve(bufs, dst, out, want_stats):
        calls.append(("assemble_solve",)); out.copy_(bufs["img"]); return "fused"

    new = lambda: torch.zeros((32, 8, 3), dtype=torch.float64)
    pipe = sharding.ShardedGradPath(None, 1, 0, 32, new, render_band, assemble, solve, phase_hook=lambda n: calls.append(("hook", n)), assemble_solve=assemble_solve)
    _, pstats = pipe.step()
    assert pstats == "fused" and float(pipe.out.sum()) == 32 * 8 * 3
    assert calls == [("render", (0, 32)), ("hook", "render"), ("hook", "exchange"), ("assemble_solve",), ("hook", "solve")]
    calls.clear()
    pipe = sharding.ShardedGradPath(None, 1, 0, 32, new, render_band, assemble, solve)
    assert pipe.step()[1] == "separate" and [c[0] for c in calls] == ["render", "assemble", "solve"]
    two = sharding.ShardedGradPath(object(), 2, 1, 32, new, render_band, assemble, solve, assemble_solve=assemble_solve)
    assert two.assemble_solve is None and two.rows == (16, 32)


def test_band_cuts_at_any_row_and_the_feedback_step():
    """sharding.bands_weighted(granularity=...), bands_from_row_costs, refine_row_costs: host logic of bench.py's band balance."""
    from gdpt_amd import sharding
    costs = [0.5] + [1.0] * 6 + [0.25]                       # 8 tile rows of a 128-row film
    tiles = sharding.bands_weighted(128, 4, costs)           # default: whole tile rows (mirror of gdpt_band_rows_weighted)
    assert all(b[0] % 16 == 0 for b in tiles) and tiles[0][0] == 0 and tiles[-1][1] == 128
    rows = sharding.bands_weighted(128, 4, costs, granularity=1)
    model = sharding.row_costs_from_tiles(128, costs)
    assert len(model) == 128 and abs(sum(model) - sum(costs)) < 1e-12
    cost_of = lambda bands, m: [sum(m[a:b]) for a, b in bands]
    assert max(cost_of(rows, model)) <= max(cost_of(tiles, model)) + 1e-12            # finer cuts never balance worse
    assert max(cost_of(rows, model)) < 1.02 * sum(costs) / 4
    assert rows == sharding.bands_from_row_costs(128, 4, model) and sharding.bands_from_row_costs(128, 4, model, 16) == tiles
    # ragged film, more ranks than units, a short last tile row
    assert sharding.bands_weighted(40, 2, [1.0, 1.0, 1.0], granularity=4)[-1][1] == 40
    assert sharding.bands_from_row_costs(3, 5, [1.0, 1.0, 1.0]) == [(0, 1), (1, 2), (2, 3), (3, 3), (3, 3)]
    with pytest.raises(ValueError):
        sharding.bands_weighted(128, 4, costs, granularity=5)
    # feedback: band 2 took twice as long as the model says -> its rows get dearer, it shrinks; equal times leave the model's ratios
    slow = sharding.refine_row_costs(model, rows, [1.0, 1.0, 2.0, 1.0])
    again = sharding.bands_from_row_costs(128, 4, slow)
    assert again[2][1] - again[2][0] < rows[2][1] - rows[2][0] and again[-1][1] == 128
    mod = cost_of(rows, model)
    same = sharding.refine_row_costs(model, rows, mod)      # times proportional to the model: nothing to correct
    assert all(abs(a - b) < 1e-12 for a, b in zip(same, model))
    assert sharding.refine_row_costs(model, rows, [0.0, 0.0, 0.0, 0.0]) == model                  # no clock, no change
