"""The tile loop sharded over several devices behind the C ABI (gdpt_multi_*, csrc/hip/multi_gpu.hip) and bench.py's
N>1 step, on the ONE GPU a test box has:

  * device set {0} over RCCL: communicator, streams, band bookkeeping, solve on devices[0];
  * device sets {0,0} and {0,0,0} with the peer-copy transport (the only one that takes a device twice): two equal
    bands / three ragged bands with a real halo row and a real gather — the indexing RCCL's in-place all-gather uses;
  * `lajolla --devices 0,0 --exchange peer`;
  * `bench.py --gpus 2 --dist-backend gloo`: the launcher, two rank processes sharing the GPU, the sharded step with
    the HIP kernels, exchange staged through host memory.
Every result must equal the single-device one bit for bit (same per-sample streams, same per-pixel summation order).
What stays unverified on a one-GPU box: RCCL send/recv/all-gather between distinct devices (the driver's 8-GPU run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, scene_variant

BUFS = ("img", "cx0", "cy0", "cx1", "cy1")


def test_band_rows_of_the_c_host_equal_the_python_mirror(G):
    from gdpt_amd import sharding
    for height in (1, 15, 16, 17, 64, 80, 512, 720, 1024, 1279):
        for n in (1, 2, 3, 4, 7, 8, 16):
            bands = [G.band_rows(height, n, b) for b in range(n)]
            assert bands == sharding.all_bands(height, n), (height, n)
            assert bands[0][0] == 0 and bands[-1][1] == height
            for a, b in zip(bands, bands[1:]):
                assert a[1] == b[0] and (a[0] % 16 == 0 or a[0] == a[1] == height)      # bands without rows sit at the end


def test_cost_balanced_bands_of_the_c_host_equal_the_python_mirror(G):
    """gdpt_band_rows_weighted and sharding.bands_weighted cut the same bands: contiguous, whole tile rows, rank-ordered, and
    no contiguous split has a cheaper most-expensive band (checked by brute force on small cases)."""
    import itertools
    from gdpt_amd import sharding
    rng = np.random.default_rng(7)
    for height in (16, 40, 64, 80, 200, 512, 720, 1024):
        T = (height + 15) // 16
        for costs in (np.ones(T), rng.uniform(0.5, 2.0, T), np.concatenate([np.full(T // 2, 1.0), np.full(T - T // 2, 3.0)]), np.zeros(T)):
            for n in (1, 2, 3, 4, 8, 16):
                bands = [G.band_rows_weighted(height, n, b, list(costs)) for b in range(n)]
                assert bands == sharding.bands_weighted(height, n, list(costs)), (height, n)
                assert bands[0][0] == 0 and bands[-1][1] == height
                for a, b in zip(bands, bands[1:]):
                    assert a[1] == b[0] and (a[0] % 16 == 0 or a[0] == a[1] == height)
                owned = [b for b in bands if b[1] > b[0]]
                assert len(owned) == min(n, T)
                worst = max(costs[b[0] // 16:(b[1] + 15) // 16].sum() for b in owned)
                if T <= 8:                                   # every split into min(n, T) non-empty runs
                    k = min(n, T)
                    for cuts in itertools.combinations(range(1, T), k - 1):
                        edges = (0,) + cuts + (T,)
                        assert max(costs[a:b].sum() for a, b in zip(edges, edges[1:])) >= worst - 1e-12
    # uniform costs and a tile-row count the bands divide: the equal split
    assert sharding.bands_weighted(512, 8, [1.0] * 32) == sharding.all_bands(512, 8)


def test_row_cost_bands_of_the_c_host_equal_the_python_mirror(G):
    """gdpt_band_rows_from_row_costs and sharding.bands_from_row_costs: the same cuts at every granularity, optimal by brute force
    on small cases."""
    import itertools
    from gdpt_amd import sharding
    rng = np.random.default_rng(11)
    for height in (1, 3, 16, 37, 64, 130, 512):
        for costs in (np.ones(height), rng.uniform(0.2, 3.0, height), np.concatenate([np.full(height // 2, 1.0), np.full(height - height // 2, 2.5)]), np.zeros(height)):
            for n in (1, 2, 3, 8):
                for g in (1, 4, 16):
                    bands = [G.band_rows_from_row_costs(height, n, b, list(costs), g) for b in range(n)]
                    assert bands == sharding.bands_from_row_costs(height, n, list(costs), g), (height, n, g)
                    assert bands[0][0] == 0 and bands[-1][1] == height and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
                    assert all(b[0] % g == 0 or b[0] == height for b in bands)
                    if height <= 12 and g == 1:
                        k = min(n, height)
                        worst = max(costs[a:b].sum() for a, b in bands if b > a)
                        for cuts in itertools.combinations(range(1, height), k - 1):
                            edges = (0,) + cuts + (height,)
                            assert max(costs[a:b].sum() for a, b in zip(edges, edges[1:])) >= worst - 1e-12
    with pytest.raises(G.GdptError):
        G.band_rows_from_row_costs(64, 2, 0, [1.0] * 64, 5)


@pytest.mark.gpu
def test_device_set_rebalanced_by_band_times(G, scene_tmp):
    """gdpt_multi_rebalance: measured band times correct the handle's cost model and the bands are cut again at any row; the
    images of the next call equal the single-device images with the new work-item plan, and the cuts are the Python mirror's
    (what bench.py's ranks compute from the all-gathered times)."""
    from gdpt_amd import sharding
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=64, height=128)
    sd = G.parse_scene(xml)
    one = G.Scene(sd)
    for balance in (False, True):
        ms = G.MultiScene(sd, (0, 0, 0), exchange=G.EXCHANGE_PEER_COPY, balance=balance)
        _, _, _, st = ms.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True)
        bands = [(st.row_begin[i], st.row_end[i]) for i in range(3)]
        model = sharding.row_costs_from_tiles(128, one.tile_row_costs()) if balance else [1.0] * 128
        times = [1.0, 2.0, 1.5]                                  # as if the middle band had taken twice as long as the first
        ms.rebalance(times, granularity=1)
        model = sharding.refine_row_costs(model, bands, times)
        want_bands = sharding.bands_from_row_costs(128, 3, model)
        out, bufs, rs, st2 = ms.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True)
        got = [(st2.row_begin[i], st2.row_end[i]) for i in range(3)]
        assert got == want_bands and got != bands and got[1][1] - got[1][0] < bands[1][1] - bands[1][0]
        assert any(b[0] % 16 for b in got[1:])                   # cuts at rows that are no tile boundaries
        want_out, want, wrs, _ = one.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True, plan_rows=max(b[1] - b[0] for b in got))
        assert np.array_equal(out, want_out)
        for k in BUFS:
            assert np.array_equal(bufs[k], want[k]), k
        assert rs.rays == wrs.rays
        # a second round with the real clock of the call just made: still a partition of the film
        ms.rebalance([st2.render_ms[i] for i in range(3)], granularity=16)
        _, _, _, st3 = ms.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True)
        again = [(st3.row_begin[i], st3.row_end[i]) for i in range(3)]
        assert again[0][0] == 0 and again[-1][1] == 128 and all(a[1] == b[0] and a[0] % 16 == 0 for a, b in zip(again, again[1:]))
        with pytest.raises(G.GdptError):
            ms.rebalance([1.0, -1.0, 1.0])
        ms.close()


@pytest.mark.gpu
def test_cost_balanced_device_set(G, scene_tmp):
    """GdptMultiConfig.balance: the bands come from a pilot render's ray counts (gdpt_tile_row_costs) and
    gdpt_band_rows_weighted; the image still equals the single-device image with the same work-item plan, and the same pilot
    through the Python mirror cuts the same bands (what bench.py's ranks do)."""
    from gdpt_amd import sharding
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=64, height=128)
    sd = G.parse_scene(xml)
    one = G.Scene(sd)
    costs = one.tile_row_costs()
    assert len(costs) == 8 and all(c > 0 for c in costs) and costs == one.tile_row_costs()      # exact counts: reproducible
    bands = sharding.bands_weighted(128, 3, costs)
    ms = G.MultiScene(sd, (0, 0, 0), exchange=G.EXCHANGE_PEER_COPY, balance=True)
    out, bufs, rs, st = ms.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True)
    assert [(st.row_begin[i], st.row_end[i]) for i in range(3)] == bands
    want_out, want, wrs, _ = one.gradient_path_render(5, G.RNG_SAMPLE, return_buffers=True, plan_rows=max(b[1] - b[0] for b in bands))
    assert np.array_equal(out, want_out)
    for k in BUFS:
        assert np.array_equal(bufs[k], want[k]), k
    assert rs.rays == wrs.rays


@pytest.mark.gpu
@pytest.mark.parametrize("devices,exchange,height", [((0,), "rccl", 64), ((0,), "peer", 64), ((0, 0), "peer", 64),
                                                     ((0, 0, 0), "peer", 80), ((0, 0, 0, 0), "peer", 48)])
def test_multi_device_set_equals_single_device(G, scene_tmp, devices, exchange, height):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=72, height=height)
    sd = G.parse_scene(xml)
    # the device set cuts every pixel's samples for its largest band (GdptRenderParams.plan_rows); a single device given the
    # same plan must produce the same bits
    plan = max(b[1] - b[0] for b in (G.band_rows(height, len(devices), i) for i in range(len(devices))))
    want_out, want, wrs, _ = G.Scene(sd).gradient_path_render(6, G.RNG_SAMPLE, return_buffers=True, plan_rows=plan)
    ms = G.MultiScene(sd, devices, exchange=G.EXCHANGE_RCCL if exchange == "rccl" else G.EXCHANGE_PEER_COPY)
    for _ in range(2):                                    # twice: no stale state between calls
        out, bufs, rs, st = ms.gradient_path_render(6, G.RNG_SAMPLE, return_buffers=True)
        assert np.array_equal(out, want_out)
        for k in BUFS:
            assert np.array_equal(bufs[k], want[k]), k
        assert rs.rays == wrs.rays and rs.bounces == wrs.bounces and rs.samples == wrs.samples
    assert st.num_devices == len(devices)
    assert [(st.row_begin[i], st.row_end[i]) for i in range(len(devices))] == [G.band_rows(height, len(devices), i) for i in range(len(devices))]
    assert st.solve_ms > 0 and st.render_ms_max > 0 and st.wall_ms > 0


@pytest.mark.gpu
@pytest.mark.parametrize("band", [0, 1, 2])
@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_a_band_that_fails_does_not_hang_the_others(G, scene_tmp, band, stage):
    """A device that fails in any stage of a call (render, halo + assembly, all-gather, solve) must not leave the other
    bands waiting for it: every rank arrives at the agreed checkpoint behind the stage, the verdict is formed inside the
    barrier, all stand down together, the call reports the band that failed — and the handle still works afterwards
    (peer-copy transport: nothing is torn down; with RCCL the communicators would be aborted and the handle refused).
    The failure is injected by the test knobs multi_fail_band / multi_fail_stage (csrc/hip/multi_gpu.hip: rank_body)."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=48, height=80)
    sd = G.parse_scene(xml)
    want_out, _, _, _ = G.Scene(sd).gradient_path_render(3, G.RNG_SAMPLE, return_buffers=True, plan_rows=32)   # bands of 32, 32, 16 rows
    ms = G.MultiScene(sd, (0, 0, 0), exchange=G.EXCHANGE_PEER_COPY)
    if stage == 4 and band != 0:
        pytest.skip("the solve runs on the first device only")
    with G.debug_knobs(multi_fail_band=band, multi_fail_stage=stage):
        with pytest.raises(G.GdptError, match=rf"band {band}\): injected failure \(stage {stage}\)"):
            ms.gradient_path_render(3, G.RNG_SAMPLE)
    out, _, _, _ = ms.gradient_path_render(3, G.RNG_SAMPLE, return_buffers=True)
    assert np.array_equal(out, want_out)


@pytest.mark.gpu
def test_multi_rejects_bad_device_sets(G, scene_tmp):
    sd = G.parse_scene(scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=32, height=32))
    with pytest.raises(G.GdptError, match="visible"):
        G.MultiScene(sd, (0, 99))
    with pytest.raises(G.GdptError, match="distinct"):
        G.MultiScene(sd, (0, 0), exchange=G.EXCHANGE_RCCL)
    with pytest.raises(G.GdptError):
        G.MultiScene(sd, ())


@pytest.mark.gpu
def test_cli_row_bands(G, scene_tmp, tmp_path):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=64, height=64)
    exe = os.path.join(ROOT, "gradient-based-path-tracing_amd", "lajolla")
    one, two = tmp_path / "one.pfm", tmp_path / "two.pfm"
    r1 = subprocess.run([exe, "-o", str(one), "--spp", "4", "--plan-rows", "32", xml], capture_output=True, text=True, timeout=300)   # work items cut as for the two 32-row bands
    r2 = subprocess.run([exe, "-o", str(two), "--spp", "4", "--devices", "0,0", "--exchange", "peer", xml], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert "2 row bands (peer copies)" in r2.stdout
    assert one.read_bytes() == two.read_bytes()
    r3 = subprocess.run([exe, "--gpus", "64", xml], capture_output=True, text=True)
    assert r3.returncode != 0


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "4", "--strong-spp", "8",
           "--no-pmc", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_equals_one_rank(G):
    """`python bench.py --gpus 2` as typed (self-launching), both ranks on this box's one GPU over gloo: same strong-scaling
    image as the single-rank run, hash for hash."""
    one = _bench("--plan-bands", "2", "--bands", "cost")          # one rank, work items cut as for two bands: the image two ranks produce
    two = _bench("--gpus", "2", "--dist-backend", "gloo", "--bands", "cost")    # (pilot bands: the same cuts run after run)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and "rehearsal" in two and "rehearsal" not in one
    assert two["band_feedback"] is None
    assert two["scaling"] == "weak" and two["config"]["workload"].count("8 spp total")
    assert one["scaling_strong"]["out_sha1"] == two["scaling_strong"]["out_sha1"]
    assert two["value"] > 0 and two["exchange_ms"] > 0
    # the default: the pilot's bands corrected twice by the ranks' own render times during the warm-up (the cuts then depend on the
    # clock, the image on the cuts only through the work-item plan): every round's bands cover the film, the figures are there
    fb = _bench("--gpus", "2", "--dist-backend", "gloo")
    assert fb["band_feedback"]["rounds"] == 2 and len(fb["band_feedback"]["rows"]) == 3 and len(fb["band_feedback"]["render_ms_per_rank"]) == 2
    assert all(sum(rows) == 512 and len(rows) == 2 for rows in fb["band_feedback"]["rows"]) and all(t > 0 for ts in fb["band_feedback"]["render_ms_per_rank"] for t in ts)
    assert fb["value"] > 0 and "time feedback" in fb["config"]["sharding"]
    # and N=2 over RCCL on a one-GPU node is refused with a message, not a line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and not r.stdout.strip()
