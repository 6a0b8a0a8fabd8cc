"""Row-band sharding of the GradPath tile loop across ranks (one process per GPU) and the gather that
precedes the global Poisson solve.

The reference parallelises gradient_path_render over 16x16 tiles only (src/render.cpp:271-277); a tile's result
depends on nothing but the read-only scene and its RNG streams, so contiguous bands of whole tile rows go to
ranks. The exchange step of the path sits between the render and the global solve. Two forms:

  * gather_bands: all-gather of an accumulation buffer (five of them, then assembly everywhere);
  * halo_then_gather (what bench.py uses): with row bands cx = cx0(x,y) + cx1(x-1,y) is band-local and
    cy = cy0(x,y) + cy1(x,y-1) needs only the last cy1 row of the band above (src/render.cpp:345-349), so each rank
    receives that one row from its predecessor (W*24 bytes, point to point), assembles c, cx, cy for its own band,
    and the three assembled images are all-gathered in place (3/5 of the bytes, no packing).
Backend-agnostic: works with torch.distributed over RCCL ("nccl") on GPUs and over gloo on CPU tensors (tests).
"""

TILE = 16  # src/render.cpp:271


def wire_device(dist, t):
    """Device the exchanged bytes travel from: the tensor's own under RCCL; host memory under gloo, which has no
    device-side all-gather / send-recv (rehearsals of the N>1 step on one GPU, `bench.py --dist-backend gloo`; the copies
    to and from the wire buffers are plain tensor copies, so the exchange code below is the same in both cases)."""
    if t.is_cuda and dist is not None and dist.get_backend() == "gloo":
        import torch
        return torch.device("cpu")
    return t.device


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


def bands_weighted(height, world, costs, granularity=TILE):
    """Contiguous bands whose largest COST is as small as a contiguous split allows; `costs[t]` >= 0 is the measured cost of tile
    row t (Scene.tile_row_costs: rays of a pilot render), spread evenly over the tile row's pixel rows. Cuts fall on multiples of
    `granularity` rows (16, the default: whole tile rows, as GDPT_RNG_TILE needs them — the mirror of gdpt_band_rows_weighted,
    csrc/hip/multi_gpu.hip, cut for cut; 1: any row — the persistent kernels anchor their 16x16 items at the band's first row, so
    a band of the SAMPLE streams need not consist of whole tile rows: what bench.py uses). Linear partition by dynamic programming
    over (bands, units), ties to the split found first. With fewer units than ranks the last ranks own nothing, as in band_rows."""
    return bands_from_row_costs(height, world, row_costs_from_tiles(height, costs), granularity)


def row_costs_from_tiles(height, costs):
    """Per-row costs from per-tile-row costs: a tile row's cost spread evenly over its rows (the last tile row may be short)."""
    T = (height + TILE - 1) // TILE
    if len(costs) != T:
        raise ValueError("one cost per 16-pixel tile row expected")
    if any(not (c >= 0.0) for c in costs):
        raise ValueError("negative or non-finite cost")
    out = []
    for t in range(T):
        rows = min(height, t * TILE + TILE) - t * TILE
        out.extend([float(costs[t]) / float(rows)] * rows)
    return out


def bands_from_row_costs(height, world, row_costs, granularity=1):
    """The linear partition behind bands_weighted on per-row costs (len(row_costs) == height)."""
    import numpy as np
    if len(row_costs) != height:
        raise ValueError("one cost per pixel row expected")
    if any(not (c >= 0.0) for c in row_costs):
        raise ValueError("negative or non-finite cost")
    g = int(granularity)
    if g < 1 or TILE % g != 0:
        raise ValueError("granularity must divide the tile height")
    U = (height + g - 1) // g                               # units of g rows (the last may be shorter)
    pre = [0.0]
    for u in range(U):
        c = 0.0
        for r in range(u * g, min(height, u * g + g)):
            c += float(row_costs[r])
        pre.append(pre[-1] + c)
    pre = np.asarray(pre, dtype=np.float64)
    used = min(world, U)
    inf = 1e300
    best = np.full((used + 1, U + 1), inf)
    cut = np.zeros((used + 1, U + 1), dtype=np.int64)
    best[0][0] = 0.0
    for k in range(1, used + 1):
        for t in range(k, U - (used - k) + 1):
            v = np.maximum(best[k - 1][k - 1:t], pre[t] - pre[k - 1:t])
            v = np.where(best[k - 1][k - 1:t] >= inf, inf, v)
            i = int(np.argmin(v))                           # first minimum: the split found first
            if v[i] < best[k][t]:
                best[k][t], cut[k][t] = v[i], k - 1 + i
    first = [U] * (world + 1)
    t = U
    for k in range(used, 0, -1):
        first[k] = t
        t = int(cut[k][t])
    first[0] = 0
    return [(min(first[r] * g, height), min(first[r + 1] * g, height)) for r in range(world)]


def refine_row_costs(row_costs, bands, times):
    """Feedback for the next frames: `times[b]` = measured render time of band b = rows bands[b] under the cost model `row_costs`.
    Returns the model with every band's rows rescaled so that the band's modelled cost is proportional to its measured time (the
    pilot's ray counts miss what a ray costs where; a frame's own clock does not). Bands without rows or without a time keep theirs.
    Every rank must pass the same `times` (all-gathered) to arrive at the same bands."""
    out = [float(c) for c in row_costs]
    tot_t = sum(float(t) for (r0, r1), t in zip(bands, times) if r1 > r0 and t > 0)
    tot_c = sum(sum(out[r0:r1]) for (r0, r1), t in zip(bands, times) if r1 > r0 and t > 0)
    if not (tot_t > 0 and tot_c > 0):
        return out
    for (r0, r1), t in zip(bands, times):
        c = sum(out[r0:r1])
        if r1 > r0 and t > 0 and c > 0:
            f = (float(t) / tot_t) / (c / tot_c)
            for r in range(r0, r1):
                out[r] *= f
    return out


def gather_bands(dist, buf, height, world, rank, bands=None):
    """In-place all-gather of one HxWx3 image whose rows [r0,r1) are valid on this rank.
    Bands are contiguous row ranges ordered by rank, so the gathered image is the concatenation. Equal bands: one in-place
    all-gather. Ragged bands (tile rows that do not divide evenly, cost-balanced bands, ranks that own nothing): one
    in-place broadcast per band, issued together — what the C host does with grouped ncclBroadcast."""
    if world == 1:
        return buf
    bands = all_bands(height, world) if bands is None else bands
    r0, r1 = bands[rank]
    wire = wire_device(dist, buf)
    if wire != buf.device:            # gloo over device tensors: gather a host copy, write it back
        host = buf.to(wire)
        gather_bands(dist, host, height, world, rank, bands)
        buf.copy_(host)
        return buf
    if equal_bands(height, world, bands):
        flat = buf.view(-1)
        n = (r1 - r0) * buf.shape[1] * buf.shape[2]
        dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n])
    else:
        works = [dist.broadcast(buf[b[0]:b[1]], src=r, async_op=True) for r, b in enumerate(bands) if b[1] > b[0]]
        for wk in works:
            wk.wait()
    return buf


def equal_bands(height, world, bands=None):
    """True when every rank owns the same, non-zero number of rows (the in-place all-gather applies)."""
    bands = all_bands(height, world) if bands is None else bands
    sizes = {b[1] - b[0] for b in bands}
    return len(sizes) == 1 and bands[-1][1] == height and bands[0][1] > bands[0][0]


def halo_exchange_cy1(dist, cy1, height, world, rank, bands=None):
    """Row r0-1 of `cy1` (HxWx3) is filled with the last row of the band above (owned by rank-1); this rank's last
    row goes to rank+1. Point-to-point, W*3 values each way."""
    if world == 1:
        return cy1
    bands = all_bands(height, world) if bands is None else bands
    r0, r1 = bands[rank]
    ops, recv_row = [], None
    # neighbours by band adjacency (skip ranks that own nothing)
    owners = [r for r in range(world) if bands[r][1] > bands[r][0]]
    if r1 > r0:
        i = owners.index(rank)
        wire = wire_device(dist, cy1)
        if i + 1 < len(owners):
            ops.append(dist.P2POp(dist.isend, cy1[r1 - 1].contiguous().to(wire), owners[i + 1]))
        if i > 0:
            recv_row = cy1.new_empty(cy1[r0 - 1].shape, device=wire)
            ops.append(dist.P2POp(dist.irecv, recv_row, owners[i - 1]))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if recv_row is not None:
        cy1[r0 - 1].copy_(recv_row)
    return cy1


def gather_packed(dist, images, height, world, rank, scratch=None, bands=None):
    """In-place all-gather of several HxWx3 images whose rows [r0,r1) are valid on this rank.

    Equal bands: bands are contiguous row ranges in rank order, so a rank's band already sits at its final offset
    rank * band_elems of the flat image and every image gathers IN PLACE, as the C host does (csrc/hip/multi_gpu.hip:
    grouped in-place ncclAllGather) — no packing and no copies; the collectives of the images are issued together and
    waited for together. Ragged bands fall back to gather_bands per image. When the wire is not the tensors' own device
    (gloo rehearsal over device tensors) the bands are packed into ONE host buffer and ONE collective instead; `scratch`
    is an optional dict reused across calls for those buffers."""
    if world == 1:
        return images
    if not equal_bands(height, world, bands):
        for im in images:
            gather_bands(dist, im, height, world, rank, bands)
        return images
    r0, r1 = band_rows(height, world, rank)
    n = len(images)
    band_elems = (r1 - r0) * images[0].shape[1] * images[0].shape[2]
    wire = wire_device(dist, images[0])
    if wire == images[0].device:
        works = []
        for im in images:
            flat = im.view(-1)
            works.append(dist.all_gather_into_tensor(flat, flat[rank * band_elems:(rank + 1) * band_elems], async_op=True))
        for wk in works:
            wk.wait()
        return images
    scratch = {} if scratch is None else scratch
    key = (n, band_elems, images[0].dtype, wire)
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["send"] = images[0].new_empty((n, band_elems), device=wire)
        scratch["recv"] = images[0].new_empty((world, n, band_elems), device=wire)
    send, recv = scratch["send"], scratch["recv"]
    for i, im in enumerate(images):
        send[i].copy_(im[r0:r1].reshape(-1))
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    for i, im in enumerate(images):      # [world][band] of image i is the image itself (bands are rank-ordered)
        im.view(world, band_elems).copy_(recv[:, i])
    return images


class ShardedGradPath:
    """One rank's view of the sharded hot path — the step bench.py times and the gloo tests drive:

        render own band -> last cy1 row to the band below -> assemble own band -> in-place all-gather of c, cx, cy
        -> global screened-Poisson solve (replicated on every rank)

    The class owns the order of those phases and the exchange between ranks (halo_exchange_cy1, gather_packed); what a
    phase computes is injected, so the same step runs over RCCL with the HIP kernels (bench.py: device pointers of torch
    CUDA tensors) and over gloo with CPU tensors (tests):

        render_band(bufs, rows, want_stats)   fills rows [r0, r1) of the five tensors img, cx0, cy0, cx1, cy1
        assemble(bufs, (c, cx, cy), rows)     src/render.cpp:340-350 on rows [r0, r1)
        solve(c, cx, cy, out, want_stats)     fourierSolve, src/render.cpp:172-254
        new_image()                           a zeroed HxWx3 float64 tensor on the rank's device
        phase_hook(name)                      optional; called after 'render', 'exchange', 'solve' (per-phase timing)
        assemble_solve(bufs, (c, cx, cy), out, want_stats)   optional; one rank only: assemble + solve as one call
                                              (gdpt_assemble_solve_device: one pass over the film instead of two, same bits)

    `dist` is torch.distributed (any backend) or None when world == 1.
    """

    NAMES = ("img", "cx0", "cy0", "cx1", "cy1")

    def __init__(self, dist, world, rank, height, new_image, render_band, assemble, solve, phase_hook=None, bands=None, assemble_solve=None):
        self.dist, self.world, self.rank, self.height = dist, int(world), int(rank), int(height)
        # `bands`: every rank's rows (the same list on all ranks), e.g. bands_weighted(...); default: equal tile-row counts
        self.bands = [tuple(b) for b in bands] if bands is not None else all_bands(self.height, self.world)
        if len(self.bands) != self.world or self.bands[0][0] != 0 or any(a[1] != b[0] for a, b in zip(self.bands, self.bands[1:])) or self.bands[-1][1] != self.height:
            raise ValueError("bands must be contiguous, rank-ordered and cover the film")
        self.rows = self.bands[self.rank]
        self.render_band, self.assemble, self.solve = render_band, assemble, solve
        self.assemble_solve = assemble_solve if self.world == 1 else None
        self.phase_hook = phase_hook or (lambda name: None)
        self.bufs = {k: new_image() for k in self.NAMES}
        self.c, self.cx, self.cy, self.out = (new_image() for _ in range(4))
        self._scratch = {}

    def step(self, want_stats=False):
        r0, r1 = self.rows
        rstats = self.render_band(self.bufs, self.rows, want_stats) if r1 > r0 else None
        self.phase_hook("render")
        if self.assemble_solve is not None:       # the whole film is here: no exchange, assembly rides on the solve's first pass
            self.phase_hook("exchange")
            pstats = self.assemble_solve(self.bufs, (self.c, self.cx, self.cy), self.out, want_stats)
            self.phase_hook("solve")
            return rstats, pstats
        if self.world > 1:            # exchange 1: the last cy1 row of the band above (W*24 bytes, point to point)
            halo_exchange_cy1(self.dist, self.bufs["cy1"], self.height, self.world, self.rank, self.bands)
        if r1 > r0:
            self.assemble(self.bufs, (self.c, self.cx, self.cy), self.rows)
        if self.world > 1:            # exchange 2: the assembled bands, gathered in place
            gather_packed(self.dist, [self.c, self.cx, self.cy], self.height, self.world, self.rank, self._scratch, self.bands)
        self.phase_hook("exchange")
        pstats = self.solve(self.c, self.cx, self.cy, self.out, want_stats)
        self.phase_hook("solve")
        return rstats, pstats
