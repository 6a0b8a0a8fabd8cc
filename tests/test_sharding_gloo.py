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
