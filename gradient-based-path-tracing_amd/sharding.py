"""Row-band sharding of the GradPath tile loop across ranks (one process per GPU) and the gather that
precedes the global Poisson solve.

The reference parallelises gradient_path_render over 16x16 tiles only (src/render.cpp:271-277); a tile's result
depends on nothing but the read-only scene and its RNG streams, so contiguous bands of whole tile rows go to
ranks. The only exchange step of the path is gathering the five accumulation buffers before
gradient assembly (cy needs cy1 of the row above, src/render.cpp:348-349) and the global solve.
Backend-agnostic: works with torch.distributed over RCCL ("nccl") on GPUs and over gloo on CPU tensors (tests).
"""

TILE = 16  # src/render.cpp:271


def band_rows(height, world, rank):
    """Rows [r0, r1) owned by `rank`: whole tile rows, balanced by tile-row count, in rank order."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    tile_rows = (height + TILE - 1) // TILE
    base, extra = divmod(tile_rows, world)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * TILE, height), min(t1 * TILE, height)


def all_bands(height, world):
    return [band_rows(height, world, r) for r in range(world)]


def gather_bands(dist, buf, height, world, rank):
    """In-place all-gather of one HxWx3 image whose rows [r0,r1) are valid on this rank.
    Bands are contiguous row ranges ordered by rank, so the gathered image is the concatenation."""
    if world == 1:
        return buf
    bands = all_bands(height, world)
    r0, r1 = bands[rank]
    sizes = {b[1] - b[0] for b in bands}
    mine = buf[r0:r1].reshape(-1)
    if len(sizes) == 1 and bands[-1][1] == height:
        dist.all_gather_into_tensor(buf.view(-1), mine.clone())
    else:  # ragged bands (tile rows do not divide evenly, or some ranks own nothing): pad to the largest band
        row_elems = buf.shape[1] * buf.shape[2]
        biggest = max(b[1] - b[0] for b in bands) * row_elems
        send = buf.new_zeros((biggest,))
        send[:mine.numel()] = mine
        parts = [buf.new_empty((biggest,)) for _ in bands]
        dist.all_gather(parts, send)
        for b, p in zip(bands, parts):
            if b[1] > b[0]:
                buf[b[0]:b[1]] = p[:(b[1] - b[0]) * row_elems].view(b[1] - b[0], buf.shape[1], buf.shape[2])
    return buf
