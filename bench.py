#!/usr/bin/env python3
"""bench.py — GDPT hot-path benchmark (contract: one JSON line on rank 0).

A *step* is one pass of the hot path over one batch: five-buffer GradPath render (base + 4 offsets per sample) ->
gradient assembly -> screened-Poisson solve, with the scene (BVH, triangles, materials) and every image buffer resident
in HBM when the timed region starts.

N = 1: BASELINE.json configs[1] — cbox_gdpt geometry, 512x512, 16 spp, one MI355X.
N > 1: the film is sharded into N contiguous row bands; every rank renders its band,
       sends the last cy1 row to the rank below, assembles c, cx, cy for its band, an in-place all-gather (RCCL) puts
       the three images on every rank, then the solve runs replicated — `gdpt_amd.sharding.ShardedGradPath.step`, the
       same object the gloo tests drive on CPU tensors. No collective inside the render.
       `value` is the weak-scaling figure (spp = 16*N on the same film: per-GPU work fixed); `scaling_strong` holds the
       north-star target beside it (256 spp in total on the same film, split over the N bands).

`python bench.py --gpus N` works as typed: without WORLD_SIZE in the environment the process only counts devices (no
HIP call), starts N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of the ranks itself.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_SCENE = os.path.join(ROOT, "scenes", "cbox", "cbox_gdpt.xml")
SPONZA = os.path.join(ROOT, "scenes", "sponza", "sponza.xml")


def host_cores():
    """CPU share of this process: cgroup quota if there is one, else the affinity mask (capped at 16, the GPU box's
    per-GPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spp", type=int, default=16, help="samples per pixel PER GPU (total = spp * gpus)")
    ap.add_argument("--scene", default=DEFAULT_SCENE)
    ap.add_argument("--film", default="", help="WxH: replace the scene's film extent")
    ap.add_argument("--alpha", type=float, default=0.04)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline counters become null)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling block (256 spp in total)")
    ap.add_argument("--strong-spp", type=int, default=256, help="total spp of the strong-scaling block (north_star: 256)")
    ap.add_argument("--shift", choices=("reference", "reconnect"), default="reference",
                    help="offset-path shift: 'reference' = the reference's behaviour (the headline metric); 'reconnect' = the "
                         "extension mode of DESIGN.md 4.4 (profiling runs only: a different workload, named in config.workload)")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the bounded CPU-oracle sample")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL (the measured configuration). gloo = rehearsal of the N>1 step where fewer than N GPUs "
                         "exist: ranks share the visible devices and the exchange is staged through host memory; the line "
                         "is marked \"rehearsal\" and is not a measurement")
    ap.add_argument("--bands", choices=("feedback", "cost", "equal"), default="feedback",
                    help="N>1: 'cost' = row bands of equal measured cost (a 1-spp pilot counts the rays of every tile row; exact "
                         "counts, so every rank cuts the same bands, run after run); 'feedback' = the same, then corrected twice during "
                         "the warm-up by every rank's own render time of a frame (all-gathered, so the ranks still agree; the bands then "
                         "depend on the clock); 'equal' = equal tile-row counts")
    ap.add_argument("--plan-bands", type=int, default=0,
                    help="cut every pixel's samples into work items as for this many row bands (default: --gpus); a 1-GPU run with "
                         "--plan-bands N produces the N-GPU run's images bit for bit (GdptRenderParams.plan_rows)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # internal: workload under rocprofv3
    return ap.parse_args(argv)


def film_of(args):
    if not args.film:
        return (0, 0)
    w, h = args.film.lower().split("x")
    return (int(w), int(h))


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes (the parent never touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, child_argv, extra_env=None, timeout=None):
    """Starts `n` processes `python child_argv...` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one
    per rank, relays rank 0's stdout and returns the first non-zero exit code (0 if all succeeded). A failing rank ends
    the others. Used by `bench.py --gpus N` and by tests/test_bench_launch.py."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(extra_env or {})
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable] + list(child_argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)   # drains rank 0 while it runs
    reader.start()
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        pending = set(range(n))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write(f"bench.py: rank {r} exited with status {code}\n")
            if rc == 0 and t_end is not None and time.time() > t_end:
                rc = 124
                sys.stderr.write("bench.py: ranks timed out\n")
            time.sleep(0.05)
    finally:
        for p in procs:                       # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(captured).decode(errors="replace"))
    sys.stdout.flush()
    return rc


def visible_gpus():
    import torch
    return torch.cuda.device_count()          # counts devices without creating a HIP context on this image


# ---------------------------------------------------------------------------------------------------------------------
# the workload rocprofv3 is put around (child process of a bench run; no torch)
# ---------------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    import gdpt_amd as G
    sd = G.parse_scene(args.scene, film=film_of(args))
    sc = G.Scene(sd, device=0)
    shift = G.SHIFT_RECONNECT if args.shift == "reconnect" else G.SHIFT_REFERENCE
    for _ in range(10):                  # (pmc.kernel_times leaves every kernel's first launch out of its average)
        sc.gradient_path_render(spp=args.spp, rng_scheme=G.RNG_SAMPLE, alpha=args.alpha, shift=shift)


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch
    import gdpt_amd as G
    from gdpt_amd import pmc, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    rehearsal = args.dist_backend == "gloo" and world > 1
    if local_rank >= ndev:
        if not rehearsal:
            raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, {ndev} visible (use --dist-backend gloo to rehearse "
                             f"the N>1 step on fewer GPUs)")
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    sd = G.parse_scene(args.scene, film=film_of(args))
    scene = G.Scene(sd, device=local_rank)
    W, H = scene.width, scene.height
    shift = G.SHIFT_RECONNECT if args.shift == "reconnect" else G.SHIFT_REFERENCE
    stream = torch.cuda.current_stream().cuda_stream
    names = sharding.ShardedGradPath.NAMES
    ptr = lambda t: t.data_ptr()
    scene_rel = os.path.relpath(os.path.abspath(args.scene), ROOT)

    # work items are cut for the largest band of the sharding (the same on every rank): GdptRenderParams.plan_rows
    plan_bands = args.plan_bands if args.plan_bands > 0 else world
    tile_costs = scene.tile_row_costs() if args.bands in ("cost", "feedback") and max(world, plan_bands) > 1 else None

    def cut(n):        # cuts at any row: the SAMPLE streams do not need whole tile rows (items are anchored at the band's first row)
        return sharding.bands_weighted(H, n, tile_costs, granularity=1) if tile_costs is not None and n > 1 else sharding.all_bands(H, n)
    bands = cut(world)
    plan_rows = max(b[1] - b[0] for b in cut(plan_bands))

    def make_pipeline(spp_total, hook=None):
        def render_band(bufs, rows, want_stats):
            return scene.render_device([ptr(bufs[k]) for k in names], spp=spp_total, rng_scheme=G.RNG_SAMPLE, rows=rows,
                                       stream=stream, want_stats=want_stats, shift=shift, plan_rows=plan_rows)

        def assemble(bufs, dst, rows):
            G.assemble_device(W, H, [ptr(bufs[k]) for k in names], [ptr(t) for t in dst], stream=stream, rows=rows)

        def solve(c, cx, cy, out, want_stats):
            return G.poisson_solve_device(W, H, ptr(c), ptr(cx), ptr(cy), ptr(out), alpha=args.alpha, stream=stream, want_stats=want_stats)

        def assemble_solve(bufs, dst, out, want_stats):
            return G.assemble_solve_device(W, H, [ptr(bufs[k]) for k in names], [ptr(t) for t in dst], ptr(out), alpha=args.alpha, stream=stream, want_stats=want_stats)

        return sharding.ShardedGradPath(dist, world, rank, H, lambda: torch.zeros((H, W, 3), dtype=torch.float64, device=dev),
                                        render_band, assemble, solve, phase_hook=hook, bands=bands, assemble_solve=assemble_solve)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(pipe, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(warmup):
            pipe.step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def phases(pipe, reps):
        """Per-phase device time of a step (events on the launch stream, outside the timed region): max over ranks."""
        marks = []
        pipe.phase_hook = lambda name: marks.append((name, _mark()))

        def _mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        acc = {"render": [], "exchange": [], "solve": []}
        for _ in range(reps):
            marks.clear()
            start = _mark()
            pipe.step()
            torch.cuda.synchronize()
            prev = start
            for name, e in marks:
                acc[name].append(prev.elapsed_time(e))
                prev = e
        pipe.phase_hook = lambda name: None
        out = {k: float(np.mean(v)) for k, v in acc.items()}
        if world > 1:
            t = torch.tensor([out["render"], out["exchange"], out["solve"]], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out = dict(zip(("render", "exchange", "solve"), [float(x) for x in t.tolist()]))
        return out

    def local_render_ms(pipe, reps):
        """This rank's own render time of a step (events on the launch stream): best of `reps`."""
        best = 1e30
        for _ in range(reps):
            got = {}
            pipe.phase_hook = lambda name: got.setdefault(name, _rec())

            def _rec():
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
            e0 = _rec()
            pipe.step()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(got["render"]))
        pipe.phase_hook = lambda name: None
        return best

    # ---- the timed region: weak scaling, args.spp per GPU
    spp_total = args.spp * world
    pipe = make_pipeline(spp_total)
    band_feedback = None
    if world > 1 and args.bands == "feedback" and tile_costs is not None:
        # The pilot counts rays; what a ray costs where, it does not know (the lower half of the cbox film is 10 % dearer per ray:
        # profiles/r03_band_costs.txt). Two rounds of feedback from the frame's own clock, before anything is timed: every rank
        # measures its band, the times are all-gathered, every rank rescales the cost model band by band and cuts again.
        row_costs = sharding.row_costs_from_tiles(H, tile_costs)
        band_feedback = {"rounds": 2, "rows": [[b[1] - b[0] for b in bands]], "render_ms_per_rank": []}
        for _ in range(2):
            pipe.step()
            mine = torch.tensor([local_render_ms(pipe, 2)], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            times = [float(t.item()) for t in every]
            row_costs = sharding.refine_row_costs(row_costs, bands, times)
            bands = sharding.bands_from_row_costs(H, world, row_costs)
            if plan_bands == world:
                plan_rows = max(b[1] - b[0] for b in bands)
            band_feedback["render_ms_per_rank"].append([round(t, 4) for t in times])
            band_feedback["rows"].append([b[1] - b[0] for b in bands])
            pipe = make_pipeline(spp_total)
    r0, r1 = pipe.rows
    elapsed = timed(pipe, args.steps, args.warmup)
    ph = phases(pipe, max(3, min(args.steps, 10)))
    rs, ps = pipe.step(want_stats=True)       # counters of one launch (synchronises)
    samples_all = W * H * spp_total
    ms_per_step = elapsed / args.steps * 1e3
    value = samples_all / (elapsed / args.steps) / 1e6

    # traversal counters from the counting build of the same kernel (one extra launch of this rank's band)
    import ctypes as C
    cs = G.GdptRenderStats()
    cs.nodes_visited = 2 ** 64 - 1            # request flag understood by gdpt_render_device
    p = G._params(spp_total, G.RNG_SAMPLE, (r0, r1), shift=shift, plan_rows=plan_rows)
    G._check(G.lib().gdpt_render_device(scene.handle, C.byref(p), *[C.c_void_p(ptr(pipe.bufs[k])) for k in names],
                                        C.c_void_p(stream), C.byref(cs)))

    # ---- supplementary: two frames in flight (N = 1 only). A launch ends with the drain of its longest paths (0.4 ms of
    # 2.4 on this workload: DESIGN.md 4.1) during which most of the chip idles; a renderer that produces a stream of
    # frames starts the next frame's render on a second stream meanwhile. Same work per step, same kernels, every step still
    # render -> assemble -> solve; `value` above stays the one-frame-at-a-time figure.
    pipelined = None
    if world == 1 and args.shift == "reference" and not args.no_strong:
        scene2 = G.Scene(sd, device=local_rank)                  # a scene handle owns one set of launch scratch: one per frame in flight
        side = torch.cuda.Stream(device=dev)

        def make2(sc, st_):
            def render_band(bufs, rows, want_stats):
                return sc.render_device([ptr(bufs[k]) for k in names], spp=spp_total, rng_scheme=G.RNG_SAMPLE, rows=rows, stream=st_, want_stats=False, shift=shift, plan_rows=plan_rows)

            def assemble(bufs, dst, rows):
                G.assemble_device(W, H, [ptr(bufs[k]) for k in names], [ptr(t) for t in dst], stream=st_, rows=rows)

            def solve(c, cx, cy, out, want_stats):
                return G.poisson_solve_device(W, H, ptr(c), ptr(cx), ptr(cy), ptr(out), alpha=args.alpha, stream=st_, want_stats=False)
            return sharding.ShardedGradPath(None, 1, 0, H, lambda: torch.zeros((H, W, 3), dtype=torch.float64, device=dev), render_band, assemble, solve)
        pa, pb = make2(scene, stream), make2(scene2, side.cuda_stream)
        for _ in range(2):
            pa.step(); pb.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            (pa if k % 2 == 0 else pb).step()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        same = bool(torch.equal(pa.out, pipe.out) and torch.equal(pb.out, pipe.out))
        pipelined = {"frames_in_flight": 2, "steps": args.steps, "ms_per_step": el2 / args.steps * 1e3,
                     "value": samples_all / (el2 / args.steps) / 1e6, "unit": "Msamples/s", "outputs_equal_sequential": same,
                     "note": "consecutive frames alternate between two HIP streams and two scene handles; not the headline value"}
        G.poisson_forget_stream(side.cuda_stream)      # the solver keeps scratch per (device, stream): dropped with the stream
        del pa, pb, scene2

    # ---- strong scaling: the north-star target, 256 spp in total split over the bands
    strong = None
    if not args.no_strong and args.shift == "reference":
        spipe = make_pipeline(args.strong_spp)
        ssteps = max(3, min(args.steps, 5))
        sel = timed(spipe, ssteps, 1)
        sph = phases(spipe, 3)
        torch.cuda.synchronize()
        import hashlib
        strong = {"workload": f"{scene_rel} Integrator::GradPath {W}x{H}, {args.strong_spp} spp in total over {world} band(s)",
                  "scaling": "strong", "steps": ssteps, "ms_per_step": sel / ssteps * 1e3,
                  "value": W * H * args.strong_spp / (sel / ssteps) / 1e6, "unit": "Msamples/s",
                  "render_ms": sph["render"], "exchange_ms": sph["exchange"], "poisson_ms": sph["solve"],
                  "out_sha1": hashlib.sha1(spipe.out.cpu().numpy().tobytes()).hexdigest()[:16]}
        del spipe

    # ---- §8(d) byte model (kept as roofline.model): rays*64 + nodes*node_bytes + prims*48 + bounces*320 per launch
    render_ms = ph["render"]
    alg_bytes = cs.rays * 64 + cs.nodes_visited * cs.node_bytes + cs.tris_tested * 48 + cs.bounces * 320
    model_gbs = alg_bytes / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
    render_kernel = "gdpt_render_phases" if args.shift == "reference" else "gdpt_render_reconnect"

    result = {
        "metric": f"GDPT Msamples/s (base+4 offset) + Poisson ms, {os.path.basename(os.path.dirname(scene_rel))} {W}x{H}",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": f"scene file shipped with the reference ({scene_rel}); PCG32 stream per sample, seeded as src/pcg.h",
        "config": {"workload": f"{scene_rel} Integrator::GradPath {W}x{H}, {args.spp} spp per GPU "
                               f"({spp_total} spp total), render+assemble+Poisson(DCT-I as folded fp64 MFMA GEMMs, own kernels) per step"
                               + (" [shift=reconnect: extension mode, NOT the headline workload]" if args.shift == "reconnect" else ""),
                   "rng": "sample-stream PCG32",
                   "sharding": (f"{world} row bands ({('equal measured cost' + (' (pilot + two rounds of time feedback)' if band_feedback else ' (pilot)') + ': rows ' + ' '.join(str(b[1] - b[0]) for b in bands)) if tile_costs is not None else 'equal tile-row counts'}), "
                                f"1-row halo + in-place all-gather / per-band broadcast of c,cx,cy ({args.dist_backend})") if world > 1 else "single GPU",
                   "alpha": args.alpha},
        "render_ms": render_ms, "exchange_ms": ph["exchange"], "poisson_ms": ph["solve"],
        "render_msamples_per_s": W * (r1 - r0) * spp_total / render_ms / 1e3 if render_ms > 0 else 0.0,
        "poisson_iterations": ps.iterations if ps is not None else 0,
        "rays_per_sample": cs.rays / max(1, cs.samples), "bounces_per_sample": cs.bounces / max(1, cs.samples),
        "nonfinite_samples": int(rs.nonfinite_samples) if rs is not None else None,
        "scaling_strong": strong,
        "pipelined": pipelined,
        "band_feedback": band_feedback,
    }
    if rehearsal:
        result["rehearsal"] = (f"{world} ranks on {ndev} visible GPU(s), exchange staged through host memory over gloo: exercises the "
                               "sharded step, NOT a multi-GPU measurement")

    # ---- counters: what binds the dominant kernel (rank 0, N = 1 only: one process under rocprofv3 at a time)
    roof = {"kernel": render_kernel, "launch_ms": render_ms,
            "model": {"what": "SURVEY.md 8(d) algorithmic bytes / launch time: rays*64 + nodes*node_bytes + prims*48 + bounces*320; "
                              "for an LDS-resident scene none of these bytes leave the CU, so this is NOT an HBM utilisation",
                      "bound": "hbm", "achieved": model_gbs, "peak": pmc.HBM_PEAK_GBS, "unit": "GB/s",
                      "model_bytes_per_second_over_hbm_peak": model_gbs / pmc.HBM_PEAK_GBS,
                      "algorithmic_bytes_per_launch": alg_bytes, "nodes_per_ray": cs.nodes_visited / max(1, cs.rays),
                      "node_bytes": cs.node_bytes, "prims_per_ray": cs.tris_tested / max(1, cs.rays)}}
    kernels = []
    if rank == 0 and world == 1 and not args.no_pmc:
        child = [os.path.abspath(__file__), "--pmc-child", "--scene", args.scene, "--spp", str(spp_total), "--alpha", str(args.alpha),
                 "--shift", args.shift] + (["--film", args.film] if args.film else [])
        work = tempfile.mkdtemp(prefix="gdpt_pmc_")
        got = pmc.collect(child, work)
        ctr = _find(got["counters"], render_kernel)
        tms = _find(got["times"], render_kernel)
        launch_us = tms["avg_us"] if tms else render_ms * 1e3
        vs = pmc.valu_summary(ctr, launch_us) if ctr else {}
        traffic = pmc.hbm_bytes(ctr) if ctr else None
        roof.update({
            "bound": "valu", "unit": "TFLOP/s", "peak": pmc.VECTOR_FP64_PEAK_TFLOPS,
            "achieved": vs.get("fp64_tflops"), "frac": (vs["fp64_tflops"] / pmc.VECTOR_FP64_PEAK_TFLOPS) if "fp64_tflops" in vs else None,
            "what": "vector-fp64 issue (no MFMA on this path): achieved = (2 FMA + ADD + MUL + TRANS f64 wave-instructions) * 64 * lane_util / launch "
                    "time, peak = MI355X vector fp64; lane_util, valu_share and wait_share say where the rest goes",
            "lane_util": vs.get("lane_util"), "valu_busy_share_of_launch": vs.get("valu_busy"), "issue_slot_frac": vs.get("issue_slot_frac"),
            "valu_share_of_wave_time": vs.get("valu_share"), "wait_share_of_wave_time": vs.get("wait_share"),
            "issue_stall_share_of_wave_time": vs.get("issue_stall_share"), "fp32_tflops": vs.get("fp32_tflops"),
            "traffic": traffic, "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (2*FETCH+WRITE KB, gfx950 correction)",
            "hbm_frac_counters": (traffic / (launch_us * 1e-6) / 1e9 / pmc.HBM_PEAK_GBS) if traffic else None,
            "launch_us_rocprof": launch_us, "launch_us_first": tms["first_us"] if tms else None, "pmc_errors": got["errors"] or None})
        kernels = _kernel_lines(got, W, H, spp_total, pmc)
        if os.path.abspath(args.scene) == os.path.abspath(DEFAULT_SCENE) and os.path.exists(SPONZA):
            result["secondary_hbm_scene"] = _sponza_block(pmc, work, args)
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    else:
        roof.update({"bound": "valu", "unit": "TFLOP/s", "peak": pmc.VECTOR_FP64_PEAK_TFLOPS, "achieved": None, "frac": None, "traffic": None,
                     "what": "counters are collected at N=1 on rank 0 without --no-pmc"})
    result["roofline"] = roof
    result["kernels"] = kernels or None

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        # the checker is built -O2 for every box; the baseline leg times the same source compiled for THIS box's cores
        flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-w"]
        native = os.path.join(tempfile.mkdtemp(prefix="gdpt_oracle_"), "liboracle_native.so")
        try:
            subprocess.run(["g++"] + flags + ["-shared", "-o", native, os.path.join(ROOT, "oracle", "oracle.cpp")], check=True, timeout=300,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.environ["GDPT_ORACLE_SO"] = native
            oracle_build = "g++ " + " ".join(flags[:-1])
        except Exception:        # noqa: BLE001 (no compiler on the box: the -O2 checker build is timed instead, and the line says so)
            oracle_build = "g++ -O2 -ffp-contract=off (oracle/Makefile; the -O3 -march=native build failed on this box)"
        import oracle_py as O
        cores = host_cores()
        osc = O.OracleScene(sd.ptr, use_bvh=True)
        ob, ost = osc.render(args.cpu_spp, G.RNG_TILE, threads=cores)
        tp = time.perf_counter()
        cc, ccx, ccy = O.assemble(ob)
        O.fourier_solve(cc, ccx, ccy, args.alpha)
        cpu_poisson_ms = (time.perf_counter() - tp) * 1e3
        result["cpu_baseline"] = {"value": ost.samples / ost.seconds / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                                  "cpu": cpu_model(),
                                  "oracle_build": oracle_build,
                                  "reference_itself": "SURVEY.md §6: the reference's own binary (gcc 11 -O2, Embree replaced by a brute-force 38-triangle shim) "
                                                      "ran this scene at ~0.85 Msamples/s on 8 Xeon vCPUs @2.1 GHz in the survey container; it cannot run on "
                                                      "the GPU box (no Embree, only this repository travels)",
                                  "sample": f"oracle (CPU restatement, tile-stream RNG, {cores} threads) on the same {W}x{H} scene at "
                                            f"{args.cpu_spp} spp = {ost.samples} samples in {ost.seconds:.2f} s; "
                                            f"scipy DCT-I Poisson {cpu_poisson_ms:.1f} ms on 1 core",
                                  "poisson_ms": cpu_poisson_ms}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _find(table, name):
    for k, v in table.items():
        if name in k:
            return v
    return None


def _num_chunks(spp, pixels, lanes=256 * 2 * 256):
    """Work items per pixel of the persistent render kernel (make_chunk_plan, csrc/hip/render_kernels.hip)."""
    cap = max(1, spp * pixels // (lanes * 4))
    q = min(8, (4 * lanes + pixels - 1) // pixels) if 4 * lanes > pixels > 0 else 1
    if spp <= q:
        q = 1
    slots = 64 // q
    head = (slots - 4 if slots > 4 else 1) if q > 1 else 56
    v = (spp + q - 1) // q
    if v > cap * head:
        cap = (v + head - 1) // head
    rem, sizes = v, []
    while rem > 0:
        sz = 1 if rem <= 2 else min(max(1, (rem * 11 + 19) // 20), cap)
        if len(sizes) == slots - 1:
            sz = rem
        rem -= sz
        sizes.append(sz)
    return len(sizes) * q - (0 if any(s_ > 1 for s_ in sizes) else q * v - spp)


def _kernel_lines(got, W, H, spp, pmc):
    """One roofline line per kernel of the step besides the render kernel: HBM-bound streams priced by their algorithmic
    bytes over the un-instrumented launch time, the two DCT GEMMs by their fp64 MFMA rate."""
    n3 = W * H * 3
    lines = []

    def add(pattern, label, work, per_step=1):
        t = _find(got["times"], pattern)
        if not t:
            return
        ach = work / (t["avg_us"] * 1e-6) / 1e9
        lines.append({"kernel": label, "launches_per_step": per_step, "avg_us": t["avg_us"], "bound": "hbm", "achieved": ach,
                      "peak": pmc.HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / pmc.HBM_PEAK_GBS, "algorithmic_bytes": work,
                      "traffic": pmc.hbm_bytes(_find(got["counters"], pattern) or {})})
    items = ((W + 15) // 16) * ((H + 15) // 16) * 256 * _num_chunks(spp, W * H)
    add("gdpt_reduce_partials", "gd::gdpt_reduce_partials", items * 128.0 + 5 * n3 * 8.0)   # 128-B record per work item in, five images out
    add("assemble_rhs_kernel", "gp::assemble_rhs_kernel", 9 * 8.0 * n3)    # one rank: assembly + right-hand side in one pass, 5 reads + 4 writes per unknown
    add("assemble_kernel", "gp::assemble_kernel", 8 * 8.0 * n3)            # (several ranks: 5 reads + 3 writes per unknown ...
    add("dct_rhs_kernel", "gp::dct_rhs_kernel", 4 * 8.0 * n3)              #  ... then read c, cx, cy; write h)
    for k, t in got["times"].items():
        if "dct_fold_gemm_f64" in k:          # own kernels: two row passes (X*Cw) + two column passes (Ch^T*T) per step, all under one name
            nominal = 2.0 * 3 * (W * W * H + H * H * W) / 2          # flops of one UNFOLDED pass, mean of the two shapes
            ach = nominal / (t["avg_us"] * 1e-6) / 1e12
            lines.append({"kernel": "gp::dct_fold_gemm_f64 (4 launches per step: 2 row + 2 column passes; even/odd fold = half the flops of the plain product)",
                          "launches_per_step": 4, "avg_us": t["avg_us"], "bound": "mfma", "unit": "TFLOP/s", "peak": pmc.VECTOR_FP64_PEAK_TFLOPS,
                          "achieved": ach / 2, "frac": ach / 2 / pmc.VECTOR_FP64_PEAK_TFLOPS,
                          "achieved_counting_the_unfolded_product": ach, "frac_counting_the_unfolded_product": ach / pmc.VECTOR_FP64_PEAK_TFLOPS,
                          "traffic": pmc.hbm_bytes(_find(got["counters"], k) or {})})
        if k.startswith("Cijk_"):                                          # rows: X*Cw (2*W*W*H flop per channel); columns: Ch^T*T (2*H*H*W)
            fl = 2.0 * 3 * (W * W * H if "Ailk_Bljk" in k else H * H * W)
            ach = fl / (t["avg_us"] * 1e-6) / 1e12
            lines.append({"kernel": "rocBLAS dgemm_strided_batched " + k[:40], "launches_per_step": 2, "avg_us": t["avg_us"], "bound": "mfma",
                          "achieved": ach, "peak": pmc.VECTOR_FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / pmc.VECTOR_FP64_PEAK_TFLOPS,
                          "traffic": pmc.hbm_bytes(_find(got["counters"], k) or {})})
    add("dct_scale_kernel", "gp::dct_scale_kernel", 2 * 8.0 * n3)
    add("dct_finalize_kernel", "gp::dct_finalize_kernel", 2 * 8.0 * n3)
    return lines


def _sponza_block(pmc, work, args):
    """BASELINE configs[3] (sponza 1280x720: the scene that lives in HBM), 8 spp: where the render kernel's time goes
    there — L2 hit rate and fabric bytes next to the VALU figures. Reported beside the headline, not part of `value`."""
    child = [os.path.abspath(__file__), "--pmc-child", "--scene", SPONZA, "--film", "1280x720", "--spp", "8", "--alpha", str(args.alpha)]
    got = pmc.collect(child, os.path.join(work, "sponza"), groups=("FETCH_SIZE", "WRITE_SIZE", pmc.SQ_TIME, pmc.TCC_HITS))
    ctr = _find(got["counters"], "gdpt_render_phases") or {}
    tms = _find(got["times"], "gdpt_render_phases")
    if not tms:
        return {"errors": got["errors"]}
    vs = pmc.valu_summary(ctr, tms["avg_us"])
    traffic = pmc.hbm_bytes(ctr)
    samples = 1280 * 720 * 8
    return {"workload": "scenes/sponza/sponza.xml Integrator::GradPath 1280x720, 8 spp (BVH4 + triangles walked from HBM/L2)",
            "kernel": "gdpt_render_phases", "launch_us": tms["avg_us"], "msamples_per_s": samples / tms["avg_us"],
            "lane_util": vs.get("lane_util"), "valu_share_of_wave_time": vs.get("valu_share"), "wait_share_of_wave_time": vs.get("wait_share"),
            "l2_hit_rate": vs.get("l2_hit_rate"), "fabric_bytes_per_launch": traffic,
            "fabric_gbs": (traffic / (tms["avg_us"] * 1e-6) / 1e9) if traffic else None,
            "hbm_frac_counters": (traffic / (tms["avg_us"] * 1e-6) / 1e9 / pmc.HBM_PEAK_GBS) if traffic else None,
            "pmc_errors": got["errors"] or None}


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the parent: count devices (no HIP context), start the ranks, relay rank 0's line
        n = visible_gpus()
        if n < args.gpus and args.dist_backend == "nccl":
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {n} visible on this node "
                             f"(--dist-backend gloo rehearses the sharded step on fewer devices)\n")
            return 3
        if n < 1:
            sys.stderr.write("bench.py needs a GPU (the hot path has no CPU fallback)\n")
            return 3
        return launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
