"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the row-band sharding + gather used by bench.py.
The band renderer here is the CPU oracle (a stand-in for the GPU render, which cannot run in this container);
what is under test is the host logic: band boundaries, gather order, equality with the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, scene_variant


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, xml, spp, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gdpt_amd as G
    import oracle_py as O
    from gdpt_amd import sharding
    sd = G.parse_scene(xml)
    H = sd.height
    r0, r1 = sharding.band_rows(H, world, rank)
    sc = O.OracleScene(sd.ptr)
    if r1 > r0:
        bufs, _ = sc.render(spp, G.RNG_SAMPLE, rows=(r0, r1), threads=2)
    else:
        bufs = {k: np.zeros((H, sd.width, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
    gathered = {}
    for k, v in bufs.items():
        t = torch.from_numpy(v.copy())
        t[:r0] = -7.0
        t[r1:] = -7.0     # poison rows this rank does not own: the gather must overwrite them
        gathered[k] = sharding.gather_bands(dist, t, H, world, rank).numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 64), (3, 80), (2, 16)])
def test_band_sharding_and_gather_match_single_process(G, O, scene_tmp, tmp_path, world, height):
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=24, height=height)
    spp = 2
    sd = G.parse_scene(xml)
    whole, _ = O.OracleScene(sd.ptr).render(spp, G.RNG_SAMPLE, threads=4)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, xml, spp, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for k in whole:
            assert np.array_equal(got[k], whole[k]), (r, k)


def test_band_rows_cover_the_image_in_whole_tile_rows(G):
    from gdpt_amd import sharding
    for h in (16, 64, 512, 575, 720, 1024):
        for world in (1, 2, 3, 4, 8):
            bands = sharding.all_bands(h, world)
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                assert a1 == b0
            for r0, r1 in bands:
                assert r0 % 16 == 0 and (r1 % 16 == 0 or r1 == h) and r0 <= r1
            sizes = [(b[1] - b[0] + 15) // 16 for b in bands]
            assert max(sizes) - min(sizes) <= 1


def _assemble_np(img, cx0, cy0, cx1, cy1):
    """src/render.cpp:340-350 in numpy (c = img, cx = cx0 + cx1 shifted right, cy = cy0 + cy1 shifted down)."""
    cx = cx0.copy(); cx[:, 1:] += cx1[:, :-1]
    cy = cy0.copy(); cy[1:] += cy1[:-1]
    return img.copy(), cx, cy


def _halo_worker(rank, world, port, height, width, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gdpt_amd import sharding
    rng = np.random.default_rng(5)
    full = {k: rng.standard_normal((height, width, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
    r0, r1 = sharding.band_rows(height, world, rank)
    mine = {}
    for k, v in full.items():          # this rank only knows its own band; everything else is poison
        t = torch.full((height, width, 3), -7.0, dtype=torch.float64)
        t[r0:r1] = torch.from_numpy(v[r0:r1])
        mine[k] = t
    sharding.halo_exchange_cy1(dist, mine["cy1"], height, world, rank)
    c, cx, cy = _assemble_np(*[mine[k].numpy() for k in ("img", "cx0", "cy0", "cx1", "cy1")])
    parts = [torch.from_numpy(a.copy()) for a in (c, cx, cy)]
    for t in parts:                     # rows outside the band were assembled from poison: poison them again
        t[:r0] = -9.0
        t[r1:] = -9.0
    scratch = {}
    for _ in range(2):                  # second call reuses the scratch buffers
        sharding.gather_packed(dist, parts, height, world, rank, scratch)
    np.savez(os.path.join(out_dir, f"halo{rank}.npz"), c=parts[0].numpy(), cx=parts[1].numpy(), cy=parts[2].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 64), (4, 64), (3, 80), (3, 32)])
def test_halo_exchange_and_packed_gather_match_global_assembly(tmp_path, world, height):
    """One cy1 row from the band above + local assembly + one packed all-gather == assembling the whole image
    (equal bands use the single collective; (3,80) is ragged and (3,32) leaves a rank without rows)."""
    width = 24
    rng = np.random.default_rng(5)
    full = {k: rng.standard_normal((height, width, 3)) for k in ("img", "cx0", "cy0", "cx1", "cy1")}
    want = _assemble_np(*[full[k] for k in ("img", "cx0", "cy0", "cx1", "cy1")])
    port = _free_port()
    mp.spawn(_halo_worker, args=(world, port, height, width, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"halo{r}.npz")
        for name, w in zip(("c", "cx", "cy"), want):
            assert np.array_equal(got[name], w), (r, name)


def _reconnect_pipeline_worker(rank, world, port, xml, spp, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gdpt_amd as G
    import oracle_py as O
    from gdpt_amd import sharding
    sd = G.parse_scene(xml)
    H, W = sd.height, sd.width
    r0, r1 = sharding.band_rows(H, world, rank)
    band, _ = O.OracleScene(sd.ptr, use_bvh=True).reconnect_render(spp, rows=(r0, r1), threads=2)
    mine = {}
    for k, v in band.items():
        t = torch.full((H, W, 3), -7.0, dtype=torch.float64)
        t[r0:r1] = torch.from_numpy(v[r0:r1])
        mine[k] = t
    sharding.halo_exchange_cy1(dist, mine["cy1"], H, world, rank)
    c, cx, cy = _assemble_np(*[mine[k].numpy() for k in ("img", "cx0", "cy0", "cx1", "cy1")])
    parts = [torch.from_numpy(a.copy()) for a in (c, cx, cy)]
    sharding.gather_packed(dist, parts, H, world, rank, {})
    out = O.fourier_solve(*[t.numpy() for t in parts], 0.04)
    np.save(os.path.join(out_dir, f"rec{rank}.npy"), out)
    dist.barrier()
    dist.destroy_process_group()


def test_reconnect_mode_sharded_pipeline_matches_single_process(G, O, scene_tmp, tmp_path):
    """bench.py's N>1 step (band render -> cy1 halo -> local assembly -> packed all-gather -> replicated Poisson solve) with
    the GDPT_SHIFT_RECONNECT buffers: every rank ends with the single-process image, bit for bit."""
    xml = scene_variant(scene_tmp, "cbox/cbox_gdpt.xml", width=24, height=32)
    sd = G.parse_scene(xml)
    whole, _ = O.OracleScene(sd.ptr, use_bvh=True).reconnect_render(3, threads=4)
    want = O.fourier_solve(*_assemble_np(*[whole[k] for k in ("img", "cx0", "cy0", "cx1", "cy1")]), 0.04)
    port = _free_port()
    mp.spawn(_reconnect_pipeline_worker, args=(2, port, xml, 3, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rec{r}.npy"), want), r
