// multi_gpu.hip — Integrator::GradPath over several devices of one node, behind the C ABI (gdpt_multi_*).
//
// Replaces the reference's only parallelism: parallel_for over 16x16 tiles on a std::thread pool
// (src/render.cpp:271-277, src/parallel.cpp:183-256). A tile's result depends on nothing but the read-only scene and
// its RNG streams, so the tile loop shards into contiguous bands of whole tile rows, one band per device
// (SURVEY.md §8(e)). One host thread per device (the pool's worker, now driving a GPU):
//
//   render own band  ->  one-row halo  ->  assemble own band  ->  all-gather of c, cx, cy  ->  solve
//
//   * halo: cy(x,y) = cy0(x,y) + cy1(x,y-1) (src/render.cpp:345-349) needs the last cy1 row of the band above: W*24
//     bytes, point to point, to the next device; cx is band-local.
//   * all-gather: bands are contiguous row ranges in device order, so each image gathers IN PLACE (a device's band
//     already sits at its final offset): no packing, no copies. Three images, one group.
//   * solve: global (the DCT couples every pixel); runs on device 0 of the set, whose result goes to the host.
//
// Two transports for the exchange (GdptMultiConfig::exchange):
//   GDPT_EXCHANGE_RCCL       ncclSend/ncclRecv for the halo, grouped in-place ncclAllGather (equal bands) or grouped
//                            in-place ncclBroadcast per band (ragged bands), one communicator per device
//                            (ncclCommInitAll), every device's calls issued by its own host thread.
//   GDPT_EXCHANGE_PEER_COPY  direct device-to-device copies over xGMI: every device pushes its band into the image
//                            buffers of every other device (hipMemcpyPeerAsync on its own stream, ordered by events).
//                            xGMI is a full mesh of point-to-point links, so the N-1 pushes of a device use N-1
//                            different links at once — the "direct" all-gather SURVEY.md §5 prices at 1/7 of a ring.
//                            Also the only transport that accepts the same device twice (two bands on one GPU), which
//                            is how the band / halo / gather indexing is verified on a one-GPU box.
#include "../../../include/gdpt.h"
#include "../capi_common.h"
#include "poisson_kernels.h"
#include "scene_internal.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace {

using gdpt::ck;

void nk(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

constexpr int kTile = 16;   // src/render.cpp:271

// Rows [r0, r1) of band `b` of `n`: whole tile rows, balanced by tile-row count, in band order (sharding.band_rows).
void band_rows(int height, int n, int b, int *r0, int *r1) {
    const int tile_rows = (height + kTile - 1) / kTile;
    const int base = tile_rows / n, extra = tile_rows % n;
    const int t0 = b * base + std::min(b, extra);
    const int t1 = t0 + base + (b < extra ? 1 : 0);
    *r0 = std::min(t0 * kTile, height);
    *r1 = std::min(t1 * kTile, height);
}

// Bands of whole tile rows whose largest COST is as small as a contiguous split allows (linear partition, dynamic
// programme over (bands, tile rows); ties go to the split found first, so every caller gets the same bands). cost[t] > 0 is
// the measured cost of tile row t (gdpt_tile_row_costs: rays of a pilot render). With fewer tile rows than bands the last
// bands stay empty, as in band_rows. Mirror: sharding.band_rows_weighted.
void band_rows_weighted(int height, int n, const double *cost, int tile_rows, std::vector<int> *first_tile) {
    const int T = (height + kTile - 1) / kTile;
    if (tile_rows != T) throw std::runtime_error("band_rows_weighted: one cost per 16-pixel tile row expected");
    first_tile->assign((size_t)n + 1, T);
    const int used = std::min(n, T);                  // bands that own rows
    std::vector<double> pre((size_t)T + 1, 0.0);
    for (int t = 0; t < T; t++) {
        if (!(cost[t] >= 0.0)) throw std::runtime_error("band_rows_weighted: negative or non-finite cost");
        pre[(size_t)t + 1] = pre[(size_t)t] + cost[t];
    }
    const double inf = 1e300;
    // best[k][t]: smallest possible largest-band cost when the first t tile rows form k bands (each with >= 1 tile row)
    std::vector<std::vector<double>> best((size_t)used + 1, std::vector<double>((size_t)T + 1, inf));
    std::vector<std::vector<int>> cut((size_t)used + 1, std::vector<int>((size_t)T + 1, 0));
    best[0][0] = 0.0;
    for (int k = 1; k <= used; k++)
        for (int t = k; t <= T - (used - k); t++)
            for (int s = k - 1; s < t; s++) {
                if (best[(size_t)k - 1][(size_t)s] >= inf) continue;
                const double v = std::max(best[(size_t)k - 1][(size_t)s], pre[(size_t)t] - pre[(size_t)s]);
                if (v < best[(size_t)k][(size_t)t]) { best[(size_t)k][(size_t)t] = v; cut[(size_t)k][(size_t)t] = s; }
            }
    int t = T;
    for (int k = used; k >= 1; k--) { (*first_tile)[(size_t)k] = t; t = cut[(size_t)k][(size_t)t]; }
    (*first_tile)[0] = 0;
    for (int k = used + 1; k <= n; k++) (*first_tile)[(size_t)k] = T;
}

// The same linear partition on per-row costs, cuts on multiples of `gran` rows (gran divides 16); first_row gets n + 1 entries.
// Mirror: sharding.bands_from_row_costs (units summed row by row, prefix sums unit by unit, first minimum wins).
void bands_from_row_costs(int height, int n, const double *row_cost, int gran, std::vector<int> *first_row) {
    if (gran < 1 || kTile % gran != 0) throw std::runtime_error("bands_from_row_costs: granularity must divide the tile height");
    const int U = (height + gran - 1) / gran;
    std::vector<double> pre((size_t)U + 1, 0.0);
    for (int u = 0; u < U; u++) {
        double c = 0.0;
        for (int r = u * gran; r < std::min(height, u * gran + gran); r++) {
            if (!(row_cost[r] >= 0.0)) throw std::runtime_error("bands_from_row_costs: negative or non-finite cost");
            c += row_cost[r];
        }
        pre[(size_t)u + 1] = pre[(size_t)u] + c;
    }
    const int used = std::min(n, U);
    const double inf = 1e300;
    std::vector<std::vector<double>> best((size_t)used + 1, std::vector<double>((size_t)U + 1, inf));
    std::vector<std::vector<int>> cut((size_t)used + 1, std::vector<int>((size_t)U + 1, 0));
    best[0][0] = 0.0;
    for (int k = 1; k <= used; k++)
        for (int t = k; t <= U - (used - k); t++)
            for (int s = k - 1; s < t; s++) {
                if (best[(size_t)k - 1][(size_t)s] >= inf) continue;
                const double v = std::max(best[(size_t)k - 1][(size_t)s], pre[(size_t)t] - pre[(size_t)s]);
                if (v < best[(size_t)k][(size_t)t]) { best[(size_t)k][(size_t)t] = v; cut[(size_t)k][(size_t)t] = s; }
            }
    std::vector<int> first((size_t)n + 1, U);
    int t = U;
    for (int k = used; k >= 1; k--) { first[(size_t)k] = t; t = cut[(size_t)k][(size_t)t]; }
    first[0] = 0;
    first_row->resize((size_t)n + 1);
    for (int k = 0; k <= n; k++) (*first_row)[(size_t)k] = std::min(first[(size_t)k] * gran, height);
}

// Feedback from measured band times: every band's rows rescaled so that its modelled share equals its measured share
// (sharding.refine_row_costs, operation for operation). Bands without rows, time or modelled cost keep their rows' costs.
void refine_row_costs(std::vector<double> &row_cost, const std::vector<std::pair<int, int>> &bands, const double *ms) {
    double tot_t = 0.0, tot_c = 0.0;
    for (size_t b = 0; b < bands.size(); b++) {
        if (!(bands[b].second > bands[b].first) || !(ms[b] > 0)) continue;
        tot_t += ms[b];
        double c = 0.0;
        for (int r = bands[b].first; r < bands[b].second; r++) c += row_cost[(size_t)r];
        tot_c += c;
    }
    if (!(tot_t > 0 && tot_c > 0)) return;
    for (size_t b = 0; b < bands.size(); b++) {
        double c = 0.0;
        for (int r = bands[b].first; r < bands[b].second; r++) c += row_cost[(size_t)r];
        if (bands[b].second > bands[b].first && ms[b] > 0 && c > 0) {
            const double f = (ms[b] / tot_t) / (c / tot_c);
            for (int r = bands[b].first; r < bands[b].second; r++) row_cost[(size_t)r] *= f;
        }
    }
}

// Reusable barrier for the per-device host threads of one call (C++17: no std::barrier). Every arrival carries the
// rank's own verdict ("I have failed"); the verdict of the ROUND — has anyone? — is formed under the barrier's lock by the
// last arriver and handed to every participant, so all ranks take the same branch behind it. (Reading the other ranks'
// error strings after the barrier, as an earlier version did, raced with ranks that were already writing theirs in the
// next stage and could give two ranks different answers: one would enter a collective the other skips.)
class HostBarrier {
    std::mutex mu; std::condition_variable cv; int n, waiting = 0; unsigned long long generation = 0;
    bool any = false, verdict = false;
public:
    explicit HostBarrier(int n_) : n(n_) {}
    bool arrive_and_wait(bool failed) {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long gen = generation;
        any = any || failed;
        if (++waiting == n) { verdict = any; any = false; waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
        return verdict;       // (stable until every waiter of this round has left: the next round needs all n arrivals)
    }
};

struct Rank {
    int device = 0, row_begin = 0, row_end = 0;
    std::unique_ptr<GdptScene> scene;
    hipStream_t stream = nullptr;
    double *buf[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // img cx0 cy0 cx1 cy1 | c cx cy | out
    hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};  // timing: start, rendered, exchanged, solved
    hipEvent_t ev_rendered = nullptr, ev_pushed = nullptr;       // peer-copy ordering
    ncclComm_t comm = nullptr;
    GdptRenderStats rstats{};
    std::string error;
};

} // namespace

namespace { struct Job { const GdptRenderParams *params = nullptr; double alpha = 0; HostBarrier *bar = nullptr; bool want_stats = false; }; }

struct GdptMulti {
    int n = 0, w = 0, h = 0, exchange = GDPT_EXCHANGE_RCCL, scene_spp = 0;
    bool equal_bands = false;
    int plan_rows = 0;             // rows of the largest band: GdptRenderParams::plan_rows of every band's render
    std::vector<double> row_cost;  // cost model of the film's rows (uniform, the pilot's, or corrected by gdpt_multi_rebalance)
    bool comms_aborted = false;    // a failure behind the first collective tore the communicators down: the handle is spent
    std::vector<Rank> ranks;
    // one host thread per device 1..n-1, started once and parked between calls (device 0 is driven by the caller, as
    // parallel_for's caller takes part in the tile loop, src/parallel.cpp:204-236)
    std::vector<std::thread> workers;
    std::mutex job_mu; std::condition_variable job_cv, done_cv;
    unsigned long long job_gen = 0; int done = 0; bool quit = false;
    Job job;
    void start_workers();
    ~GdptMulti() {
        { std::lock_guard<std::mutex> lk(job_mu); quit = true; }
        job_cv.notify_all();
        for (auto &t : workers) t.join();
        for (Rank &r : ranks) {
            hipSetDevice(r.device);
            if (r.comm) ncclCommDestroy(r.comm);
            for (double *&b : r.buf) if (b) hipFree(b);
            for (hipEvent_t &e : r.ev_t) if (e) hipEventDestroy(e);
            if (r.ev_rendered) hipEventDestroy(r.ev_rendered);
            if (r.ev_pushed) hipEventDestroy(r.ev_pushed);
            if (r.stream) { gdpt::poisson_forget_stream(r.device, r.stream); hipStreamDestroy(r.stream); }   // the solver's per-stream scratch goes with the stream
            r.scene.reset();
        }
    }
};

namespace {

// Everything one device does for one call; runs on that device's host thread.
// Four stages, three agreed checkpoints. A rank that fails in a stage still arrives at the checkpoint behind it, says so,
// and every rank leaves together:
//   A (after the render)            nothing that can block has been enqueued: everyone just stands down.
//   B (after halo + band assembly)  a halo send / receive may be waiting for a peer that never posted its half;
//   C (after the all-gather)        likewise a collective some ranks entered and one did not. RCCL: every rank aborts its
//                                   communicator (ncclCommAbort ends the pending kernels), the handle is marked spent.
//                                   Peer copies wait on events only, and every event is recorded whatever happened.
// Test instrument (include/gdpt_debug.h: multi_fail_band / multi_fail_stage): throws in the named band and stage.
void rank_body(GdptMulti *m, int i, const GdptRenderParams *params, double alpha, HostBarrier *bar, bool want_stats) {
    Rank &me = m->ranks[(size_t)i];
    const int W = m->w, H = m->h, n = m->n;
    const size_t row = (size_t)W * 3;
    auto fail_safe = [&](auto &&fn) {            // a failing rank must still meet the others at every checkpoint
        if (!me.error.empty()) return;
        try { fn(); } catch (const std::exception &e) { me.error = e.what(); }
    };
    const int fail_band = gdpt::debug_knob_int("multi_fail_band", -1), fail_stage = gdpt::debug_knob_int("multi_fail_stage", 0);
    auto inject = [&](int stage) { if (fail_band == i && fail_stage == stage) throw std::runtime_error("injected failure (stage " + std::to_string(stage) + ")"); };
    auto stand_down = [&](bool collectives_pending) {
        if (me.error.empty()) me.error = "skipped: another device of the set failed";
        if (collectives_pending && me.comm) { ncclCommAbort(me.comm); me.comm = nullptr; m->comms_aborted = true; }   // (every rank writes the same value)
        hipSetDevice(me.device);
        hipStreamSynchronize(me.stream);
    };
    const bool owns = me.row_end > me.row_begin;
    // ---- stage 1: render own band
    fail_safe([&] {
        ck(hipSetDevice(me.device), "hipSetDevice");
        ck(hipEventRecord(me.ev_t[0], me.stream), "hipEventRecord");
        inject(1);
        if (owns) {
            GdptRenderParams p = params ? *params : GdptRenderParams{};
            p.row_begin = me.row_begin; p.row_end = me.row_end;
            if (p.plan_rows == 0) p.plan_rows = m->plan_rows;          // work items cut for the largest band, the same on every device
            if (me.row_begin == 0 && me.row_end == 0) throw std::runtime_error("empty band");   // (0,0) would mean "whole image"
            gdpt::render_device_impl(me.scene.get(), &p, m->scene_spp, me.buf[0], me.buf[1], me.buf[2], me.buf[3], me.buf[4], me.stream, nullptr);
        }
    });
    // (recorded whatever happened: the peer-copy transport of the rank below waits on it)
    hipSetDevice(me.device); hipEventRecord(me.ev_rendered, me.stream); hipEventRecord(me.ev_t[1], me.stream);
    // neighbours in band order that own rows (ranks without rows are skipped, as sharding.halo_exchange_cy1 does)
    int prev = -1, next = -1;
    for (int j = i - 1; j >= 0; j--) if (m->ranks[(size_t)j].row_end > m->ranks[(size_t)j].row_begin) { prev = j; break; }
    for (int j = i + 1; j < n; j++) if (m->ranks[(size_t)j].row_end > m->ranks[(size_t)j].row_begin) { next = j; break; }
    if (bar->arrive_and_wait(!me.error.empty())) { stand_down(false); return; }               // checkpoint A (every ev_rendered is recorded)
    // ---- stage 2: halo — the last cy1 row of my band goes to the next band's device (row r1-1 of ITS cy1 image) — and assembly
    fail_safe([&] {
        inject(2);
        if (!owns) return;
        if (m->exchange == GDPT_EXCHANGE_RCCL) {
            if (n > 1) {
                nk(ncclGroupStart(), "ncclGroupStart");
                if (next >= 0) nk(ncclSend(me.buf[4] + (size_t)(me.row_end - 1) * row, row, ncclDouble, next, me.comm, me.stream), "ncclSend(halo)");
                if (prev >= 0) nk(ncclRecv(me.buf[4] + (size_t)(me.row_begin - 1) * row, row, ncclDouble, prev, me.comm, me.stream), "ncclRecv(halo)");
                nk(ncclGroupEnd(), "ncclGroupEnd");
            }
        } else if (prev >= 0) {                   // pull model: wait for the band above to be rendered, copy its last row
            Rank &up = m->ranks[(size_t)prev];
            ck(hipStreamWaitEvent(me.stream, up.ev_rendered, 0), "hipStreamWaitEvent");
            ck(hipMemcpyPeerAsync(me.buf[4] + (size_t)(me.row_begin - 1) * row, me.device,
                                  up.buf[4] + (size_t)(up.row_end - 1) * row, up.device, row * sizeof(double), me.stream), "hipMemcpyPeerAsync(halo)");
        }
        // assemble own band (src/render.cpp:340-350)
        gdpt::launch_assemble(W, H, me.row_begin, me.row_end, me.buf[0], me.buf[1], me.buf[2], me.buf[3], me.buf[4], me.buf[5], me.buf[6], me.buf[7], me.stream);
    });
    if (bar->arrive_and_wait(!me.error.empty())) { stand_down(true); return; }                // checkpoint B
    // ---- stage 3: all-gather of c, cx, cy
    fail_safe([&] {
        inject(3);
        if (n == 1) return;
        if (m->exchange == GDPT_EXCHANGE_RCCL) {
            nk(ncclGroupStart(), "ncclGroupStart");
            if (m->equal_bands) {
                const size_t band_elems = (size_t)(me.row_end - me.row_begin) * row;
                for (int k = 5; k < 8; k++)       // in place: my band already sits at offset rank * band_elems
                    nk(ncclAllGather(me.buf[k] + (size_t)me.row_begin * row, me.buf[k], band_elems, ncclDouble, me.comm, me.stream), "ncclAllGather");
            } else {
                for (int j = 0; j < n; j++) {
                    const Rank &src = m->ranks[(size_t)j];
                    if (src.row_end <= src.row_begin) continue;
                    const size_t off = (size_t)src.row_begin * row, cnt = (size_t)(src.row_end - src.row_begin) * row;
                    for (int k = 5; k < 8; k++) nk(ncclBroadcast(me.buf[k] + off, me.buf[k] + off, cnt, ncclDouble, j, me.comm, me.stream), "ncclBroadcast");
                }
            }
            nk(ncclGroupEnd(), "ncclGroupEnd");
        } else if (owns) {
            // push my band into every other device's images. Nobody else writes those rows there (assembly is
            // band-local), and their previous readers (the last call's solve) were waited for before it returned.
            for (int j = 0; j < n; j++) {
                if (j == i) continue;
                Rank &dst = m->ranks[(size_t)j];
                const size_t off = (size_t)me.row_begin * row, bytes = (size_t)(me.row_end - me.row_begin) * row * sizeof(double);
                for (int k = 5; k < 8; k++)
                    if (dst.buf[k] != me.buf[k])   // (the same device twice shares nothing: each rank has its own buffers)
                        ck(hipMemcpyPeerAsync(dst.buf[k] + off, dst.device, me.buf[k] + off, me.device, bytes, me.stream), "hipMemcpyPeerAsync(band)");
            }
        }
    });
    hipSetDevice(me.device); hipEventRecord(me.ev_pushed, me.stream);
    if (bar->arrive_and_wait(!me.error.empty())) { stand_down(true); return; }                // checkpoint C (every ev_pushed is recorded)
    // ---- stage 4: solve on the first device
    fail_safe([&] {
        if (m->exchange == GDPT_EXCHANGE_PEER_COPY && i == 0)
            for (int j = 1; j < n; j++) ck(hipStreamWaitEvent(me.stream, m->ranks[(size_t)j].ev_pushed, 0), "hipStreamWaitEvent");
        ck(hipEventRecord(me.ev_t[2], me.stream), "hipEventRecord");
        inject(4);
        if (i == 0) {
            gdpt::poisson_solve_device(W, H, me.buf[5], me.buf[6], me.buf[7], alpha, me.buf[8], GDPT_SOLVER_DEFAULT, 0.0, 0, me.stream, false);
        }
        ck(hipEventRecord(me.ev_t[3], me.stream), "hipEventRecord");
        ck(hipStreamSynchronize(me.stream), "hipStreamSynchronize");
        if (want_stats && owns) {                 // counters of my band's render (the kernel has finished)
            GdptScene *sc = me.scene.get();
            ck(hipMemcpy(sc->h_counters, sc->d_counters, sizeof(gdpt::RenderCounters), hipMemcpyDeviceToHost), "hipMemcpy(counters)");
            me.rstats = GdptRenderStats{};
            me.rstats.rays = sc->h_counters->rays; me.rstats.bounces = sc->h_counters->bounces;
            me.rstats.nonfinite_samples = sc->h_counters->nonfinite;
            float ms = 0;
            ck(hipEventElapsedTime(&ms, me.ev_t[0], me.ev_t[1]), "hipEventElapsedTime");
            me.rstats.render_ms = ms;
        }
    });
    if (!me.error.empty()) { hipSetDevice(me.device); hipStreamSynchronize(me.stream); }     // (a failed solve: nothing of this call stays in flight)
}

void worker_loop(GdptMulti *m, int i) {
    unsigned long long seen = 0;
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> lk(m->job_mu);
            m->job_cv.wait(lk, [&] { return m->quit || m->job_gen != seen; });
            if (m->quit) return;
            seen = m->job_gen; job = m->job;
        }
        rank_body(m, i, job.params, job.alpha, job.bar, job.want_stats);
        { std::lock_guard<std::mutex> lk(m->job_mu); m->done++; }
        m->done_cv.notify_one();
    }
}

} // namespace

void GdptMulti::start_workers() { for (int i = 1; i < n; i++) workers.emplace_back(worker_loop, this, i); }

extern "C" {

int gdpt_band_rows(int height, int num_bands, int band, int32_t *row_begin, int32_t *row_end) {
    return gdpt::guarded([&]() {
        if (height <= 0 || num_bands <= 0 || band < 0 || band >= num_bands || !row_begin || !row_end) throw std::runtime_error("gdpt_band_rows: bad argument");
        int r0, r1;
        band_rows(height, num_bands, band, &r0, &r1);
        *row_begin = r0; *row_end = r1;
    });
}

int gdpt_band_rows_weighted(int height, int num_bands, int band, const double *tile_row_cost, int num_tile_rows, int32_t *row_begin, int32_t *row_end) {
    return gdpt::guarded([&]() {
        if (height <= 0 || num_bands <= 0 || band < 0 || band >= num_bands || !row_begin || !row_end || !tile_row_cost) throw std::runtime_error("gdpt_band_rows_weighted: bad argument");
        std::vector<int> first;
        band_rows_weighted(height, num_bands, tile_row_cost, num_tile_rows, &first);
        *row_begin = std::min(first[(size_t)band] * kTile, height); *row_end = std::min(first[(size_t)band + 1] * kTile, height);
    });
}

int gdpt_multi_create(const GdptSceneDesc *desc, const GdptMultiConfig *cfg, GdptMulti **out) {
    return gdpt::guarded([&]() {
        if (!desc || !cfg || !out) throw std::runtime_error("gdpt_multi_create: null argument");
        if (cfg->num_devices <= 0 || cfg->num_devices > GDPT_MULTI_MAX_DEVICES) throw std::runtime_error("gdpt_multi_create: num_devices out of range");
        if (cfg->exchange != GDPT_EXCHANGE_RCCL && cfg->exchange != GDPT_EXCHANGE_PEER_COPY) throw std::runtime_error("gdpt_multi_create: unknown exchange");
        int ndev = 0;
        ck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (ndev <= 0) throw std::runtime_error("gdpt_multi_create: no HIP device visible (this library has no CPU fallback)");
        const int n = cfg->num_devices;
        for (int i = 0; i < n; i++) {
            if (cfg->devices[i] < 0 || cfg->devices[i] >= ndev)
                throw std::runtime_error("gdpt_multi_create: device " + std::to_string(cfg->devices[i]) + " asked for, " + std::to_string(ndev) + " visible");
            if (cfg->exchange == GDPT_EXCHANGE_RCCL)
                for (int j = 0; j < i; j++) if (cfg->devices[j] == cfg->devices[i]) throw std::runtime_error("gdpt_multi_create: RCCL needs distinct devices (GDPT_EXCHANGE_PEER_COPY accepts a device twice)");
        }
        std::unique_ptr<GdptMulti> m(new GdptMulti());
        m->n = n; m->w = desc->camera.width; m->h = desc->camera.height; m->exchange = cfg->exchange;
        m->scene_spp = desc->samples_per_pixel;
        if (m->w <= 0 || m->h <= 0) throw std::runtime_error("gdpt_multi_create: empty film");
        m->ranks.resize((size_t)n);
        const size_t elems = (size_t)m->w * m->h * 3;
        bool equal = true;
        std::vector<int> first_tile;                 // cfg->balance: bands by the measured cost of their tile rows
        {   // every device gets its own copy of the scene; the host side of an upload (tree build with spatial splits, mip chains) is
            // seconds for a large mesh, so the N uploads run side by side instead of one after the other
            std::vector<std::exception_ptr> errs((size_t)n);
            std::vector<std::thread> th;
            auto one = [&](int i) {
                try {
                    Rank &r = m->ranks[(size_t)i];
                    r.device = cfg->devices[i];
                    r.scene.reset(new GdptScene());
                    gdpt::build_scene(desc, r.device, r.scene.get());       // sets the calling thread's device
                    r.scene->scene_spp = desc->samples_per_pixel;
                } catch (...) { errs[(size_t)i] = std::current_exception(); }
            };
            for (int i = 1; i < n; i++) th.emplace_back(one, i);
            one(0);
            for (std::thread &t : th) t.join();
            for (const std::exception_ptr &e : errs) if (e) std::rethrow_exception(e);
        }
        for (int i = 0; i < n; i++) {
            Rank &r = m->ranks[(size_t)i];
            ck(hipSetDevice(r.device), "hipSetDevice");
            if (cfg->balance && n > 1 && i > 0) {
                r.row_begin = std::min(first_tile[(size_t)i] * kTile, m->h); r.row_end = std::min(first_tile[(size_t)i + 1] * kTile, m->h);
            } else band_rows(m->h, n, i, &r.row_begin, &r.row_end);
            if (r.row_end - r.row_begin != m->ranks[0].row_end - m->ranks[0].row_begin || r.row_end <= r.row_begin) equal = false;
            if (cfg->balance && n > 1 && i == 0) {    // pilot on the first device: rays per tile row at 1 spp (exact counts: every caller sees the same bands)
                const int T = (m->h + kTile - 1) / kTile;
                std::vector<double> cost((size_t)T);
                if (gdpt_tile_row_costs(r.scene.get(), 1, cost.data(), T) != 0) throw std::runtime_error(std::string("gdpt_multi_create: pilot render: ") + gdpt_last_error());
                band_rows_weighted(m->h, n, cost.data(), T, &first_tile);
                m->row_cost.resize((size_t)m->h);                  // a tile row's cost spread over its rows (sharding.row_costs_from_tiles)
                for (int t = 0; t < T; t++) {
                    const int r0 = t * kTile, r1 = std::min(m->h, r0 + kTile);
                    for (int rr = r0; rr < r1; rr++) m->row_cost[(size_t)rr] = cost[(size_t)t] / (double)(r1 - r0);
                }
                r.row_begin = 0; r.row_end = std::min(first_tile[1] * kTile, m->h);
            }
            ck(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking), "hipStreamCreate");
            for (double *&b : r.buf) { ck(hipMalloc((void **)&b, elems * sizeof(double)), "hipMalloc(band images)"); ck(hipMemset(b, 0, elems * sizeof(double)), "hipMemset"); }
            for (hipEvent_t &e : r.ev_t) ck(hipEventCreate(&e), "hipEventCreate");
            ck(hipEventCreateWithFlags(&r.ev_rendered, hipEventDisableTiming), "hipEventCreate");
            ck(hipEventCreateWithFlags(&r.ev_pushed, hipEventDisableTiming), "hipEventCreate");
        }
        m->equal_bands = equal;
        for (const Rank &r : m->ranks) m->plan_rows = std::max(m->plan_rows, r.row_end - r.row_begin);
        if (m->row_cost.empty()) m->row_cost.assign((size_t)m->h, 1.0);
        if (cfg->exchange == GDPT_EXCHANGE_PEER_COPY) {
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    const int a = m->ranks[(size_t)i].device, b = m->ranks[(size_t)j].device;
                    if (a == b) continue;
                    int can = 0;
                    ck(hipDeviceCanAccessPeer(&can, a, b), "hipDeviceCanAccessPeer");
                    if (!can) throw std::runtime_error("gdpt_multi_create: devices " + std::to_string(a) + " and " + std::to_string(b) + " have no peer access");
                    ck(hipSetDevice(a), "hipSetDevice");
                    hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ck(e, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
        } else if (n > 1) {
            std::vector<ncclComm_t> comms((size_t)n);
            nk(ncclCommInitAll(comms.data(), n, cfg->devices), "ncclCommInitAll");
            for (int i = 0; i < n; i++) m->ranks[(size_t)i].comm = comms[(size_t)i];
        }
        m->start_workers();
        *out = m.release();
    });
}

int gdpt_band_rows_from_row_costs(int height, int num_bands, int band, const double *row_cost, int granularity_rows, int32_t *row_begin, int32_t *row_end) {
    return gdpt::guarded([&]() {
        if (height <= 0 || num_bands <= 0 || band < 0 || band >= num_bands || !row_begin || !row_end || !row_cost) throw std::runtime_error("gdpt_band_rows_from_row_costs: bad argument");
        std::vector<int> first;
        bands_from_row_costs(height, num_bands, row_cost, granularity_rows, &first);
        *row_begin = first[(size_t)band]; *row_end = first[(size_t)band + 1];
    });
}

int gdpt_multi_rebalance(GdptMulti *m, const double *band_ms, int granularity_rows) {
    return gdpt::guarded([&]() {
        if (!m || !band_ms) throw std::runtime_error("gdpt_multi_rebalance: null argument");
        for (int i = 0; i < m->n; i++) if (!(band_ms[i] >= 0.0)) throw std::runtime_error("gdpt_multi_rebalance: negative or non-finite time");
        std::vector<std::pair<int, int>> bands;
        for (const Rank &r : m->ranks) bands.push_back({r.row_begin, r.row_end});
        std::vector<double> model = m->row_cost;
        refine_row_costs(model, bands, band_ms);
        std::vector<int> first;
        bands_from_row_costs(m->h, m->n, model.data(), granularity_rows, &first);      // (throws before anything is changed)
        m->row_cost.swap(model);
        bool equal = true;
        m->plan_rows = 0;
        for (int i = 0; i < m->n; i++) {
            Rank &r = m->ranks[(size_t)i];
            r.row_begin = first[(size_t)i]; r.row_end = first[(size_t)i + 1];
            if (r.row_end - r.row_begin != first[1] - first[0] || r.row_end <= r.row_begin) equal = false;
            m->plan_rows = std::max(m->plan_rows, r.row_end - r.row_begin);
        }
        m->equal_bands = equal;
    });
}

void gdpt_multi_free(GdptMulti *m) { delete m; }

int gdpt_multi_gradient_path_render(GdptMulti *m, const GdptRenderParams *params, double dataCost, double *out_image,
                                    double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                                    GdptRenderStats *rstats, GdptMultiStats *mstats) {
    return gdpt::guarded([&]() {
        if (!m || !out_image) throw std::runtime_error("gdpt_multi_gradient_path_render: null argument");
        if (params && (params->row_begin != 0 || params->row_end != 0)) throw std::runtime_error("gdpt_multi_gradient_path_render: the bands are chosen by the library (row_begin/row_end must be 0)");
        const auto t0 = std::chrono::steady_clock::now();
        HostBarrier bar(m->n);
        const bool want = rstats != nullptr || mstats != nullptr;
        if (m->comms_aborted) throw std::runtime_error("gdpt_multi_gradient_path_render: an earlier call failed inside the exchange and its RCCL communicators were aborted; create a new handle");
        for (Rank &r : m->ranks) r.error.clear();
        {   // hand the call to the parked device threads; the calling thread drives device 0
            std::lock_guard<std::mutex> lk(m->job_mu);
            m->job = Job{params, dataCost, &bar, want}; m->done = 0; m->job_gen++;
        }
        m->job_cv.notify_all();
        rank_body(m, 0, params, dataCost, &bar, want);
        { std::unique_lock<std::mutex> lk(m->job_mu); m->done_cv.wait(lk, [&] { return m->done == m->n - 1; }); }
        for (int pass = 0; pass < 2; pass++)                  // report the device that failed, not the ones that stood down for it
            for (int i = 0; i < m->n; i++) {
                const std::string &e = m->ranks[(size_t)i].error;
                if (!e.empty() && (pass == 1 || e.compare(0, 8, "skipped:") != 0))
                    throw std::runtime_error("device " + std::to_string(m->ranks[(size_t)i].device) + " (band " + std::to_string(i) + "): " + e);
            }
        Rank &r0 = m->ranks[0];
        const size_t elems = (size_t)m->w * m->h * 3, row = (size_t)m->w * 3;
        ck(hipSetDevice(r0.device), "hipSetDevice");
        ck(hipMemcpy(out_image, r0.buf[8], elems * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
        double *host[5] = {img, cx0, cy0, cx1, cy1};
        for (int k = 0; k < 5; k++) {
            if (!host[k]) continue;
            for (Rank &r : m->ranks) {                        // the raw accumulation buffers stay sharded: collect band by band
                if (r.row_end <= r.row_begin) continue;
                ck(hipSetDevice(r.device), "hipSetDevice");
                ck(hipMemcpy(host[k] + (size_t)r.row_begin * row, r.buf[k] + (size_t)r.row_begin * row,
                             (size_t)(r.row_end - r.row_begin) * row * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H band)");
            }
        }
        if (rstats) {
            *rstats = GdptRenderStats{};
            int spp = (params && params->spp > 0) ? params->spp : m->scene_spp;
            rstats->samples = (uint64_t)m->w * (uint64_t)m->h * (uint64_t)spp;
            for (Rank &r : m->ranks) {
                rstats->rays += r.rstats.rays; rstats->bounces += r.rstats.bounces; rstats->nonfinite_samples += r.rstats.nonfinite_samples;
                rstats->render_ms = std::max(rstats->render_ms, r.rstats.render_ms);
            }
        }
        if (mstats) {
            *mstats = GdptMultiStats{};
            mstats->num_devices = m->n; mstats->exchange = m->exchange;
            for (int i = 0; i < m->n; i++) {
                Rank &r = m->ranks[(size_t)i];
                float a = 0, b = 0, c = 0;
                ck(hipSetDevice(r.device), "hipSetDevice");
                ck(hipEventElapsedTime(&a, r.ev_t[0], r.ev_t[1]), "hipEventElapsedTime");
                ck(hipEventElapsedTime(&b, r.ev_t[1], r.ev_t[2]), "hipEventElapsedTime");
                ck(hipEventElapsedTime(&c, r.ev_t[2], r.ev_t[3]), "hipEventElapsedTime");
                mstats->render_ms[i] = a; mstats->row_begin[i] = r.row_begin; mstats->row_end[i] = r.row_end;
                mstats->render_ms_max = std::max(mstats->render_ms_max, (double)a);
                mstats->exchange_ms = std::max(mstats->exchange_ms, (double)b);     // includes waiting for the slowest band
                if (i == 0) mstats->solve_ms = c;
            }
            mstats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    });
}

} // extern "C"
