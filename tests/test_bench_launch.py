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


@pytest.mark.parametrize("world,height,bands", [(2, 64, "equal"), (3, 80, "equal"), (3, 112, "weighted")])
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
