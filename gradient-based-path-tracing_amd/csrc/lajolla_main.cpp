// lajolla_main.cpp — drop-in `./lajolla [-t num_threads] [-o output_file_name] filename.xml` for the
// GradPath path (reference CLI: src/main.cpp:11-51). Host C++ over the C ABI in include/gdpt.h.
//
// Same flags, same stdout lines ("Parsing and constructing scene ...", "Done. Took X seconds.",
// "Rendering...", "Image written to ..."), same output-name quirk (with several scenes the first
// scene's name sticks, src/main.cpp:42). Extra flags, because the reference hard-codes them:
//   --spp N       samples per pixel (default: the scene's <sampler sampleCount>)
//   --ref-spp     the reference's hard-coded 1000 spp (src/render.cpp:293)
//   --rng tile|sample   PCG stream assignment (default sample; tile = the reference's order, slow)
//   --alpha A     Poisson data weight (default 0.04, src/render.cpp:353)
//   --device D    GPU index
//   --film WxH    replace the scene's <film> extent (benchmark configurations quote their own)
//   --gpus N      shard the tile loop into N row bands over devices 0..N-1 (gdpt_multi_*: one host thread per GPU, as
//                 the reference's -t threads share the tile grid, src/parallel.cpp:183-256); --devices a,b,.. names them
//   --exchange rccl|peer   transport of the halo row + all-gather between the bands (default rccl)
//   --plan-rows R   cut every pixel's samples into work items as for a band of R rows (GdptRenderParams::plan_rows): a
//                 single-device run with the R of an N-band run (its largest band) writes that run's image bit for bit
// `-t` is accepted for compatibility; rendering runs on the GPU, so it has no effect.
#include "../../include/gdpt.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, char *argv[]) {
    if (argc <= 1) {
        std::cout << "[Usage] ./lajolla [-t num_threads] [-o output_file_name] filename.xml" << std::endl;
        return 0;
    }
    int num_threads = 0, spp = 0, device = 0, rng = GDPT_RNG_SAMPLE, shift = GDPT_SHIFT_REFERENCE;
    int film_w = 0, film_h = 0, plan_rows = 0;
    GdptMultiConfig multi{};          // num_devices == 0: single-device entry points
    double alpha = 0.04;
    std::string outputfile = "";
    std::vector<std::string> filenames;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; std::exit(2); } return argv[++i]; };
        if (a == "-t") num_threads = std::stoi(next());
        else if (a == "-o") outputfile = next();
        else if (a == "--spp") spp = std::stoi(next());
        else if (a == "--ref-spp") spp = 1000;
        else if (a == "--alpha") alpha = std::stod(next());
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--film") {
            std::string v = next();
            size_t xpos = v.find('x');
            if (xpos == std::string::npos) { std::cerr << "--film expects WxH" << std::endl; return 2; }
            film_w = std::stoi(v.substr(0, xpos)); film_h = std::stoi(v.substr(xpos + 1));
        }
        else if (a == "--gpus") {
            multi.num_devices = std::stoi(next());
            if (multi.num_devices < 1 || multi.num_devices > GDPT_MULTI_MAX_DEVICES) { std::cerr << "--gpus out of range" << std::endl; return 2; }
            for (int k = 0; k < multi.num_devices; k++) multi.devices[k] = k;
        }
        else if (a == "--devices") {      // comma-separated HIP ordinals, band order
            std::string v = next();
            multi.num_devices = 0;
            size_t pos = 0;
            while (pos <= v.size()) {
                size_t c = v.find(',', pos);
                if (c == std::string::npos) c = v.size();
                if (multi.num_devices >= GDPT_MULTI_MAX_DEVICES) { std::cerr << "--devices: too many" << std::endl; return 2; }
                multi.devices[multi.num_devices++] = std::stoi(v.substr(pos, c - pos));
                pos = c + 1;
            }
        }
        else if (a == "--exchange") {
            std::string v = next();
            if (v == "rccl") multi.exchange = GDPT_EXCHANGE_RCCL;
            else if (v == "peer") multi.exchange = GDPT_EXCHANGE_PEER_COPY;
            else { std::cerr << "unknown --exchange " << v << " (rccl | peer)" << std::endl; return 2; }
        }
        else if (a == "--plan-rows") plan_rows = std::stoi(next());
        else if (a == "--rng") { std::string v = next(); rng = (v == "tile") ? GDPT_RNG_TILE : GDPT_RNG_SAMPLE; }
        else if (a == "--shift") {        // extension: "reconnect" = GDPT_SHIFT_RECONNECT (include/gdpt.h); default = the reference's offsets
            std::string v = next();
            if (v == "reconnect") shift = GDPT_SHIFT_RECONNECT;
            else if (v == "reference") shift = GDPT_SHIFT_REFERENCE;
            else { std::cerr << "unknown --shift " << v << " (reference | reconnect)" << std::endl; return 2; }
        }
        else filenames.push_back(a);
    }
    (void)num_threads;

    using clock = std::chrono::system_clock;
    for (const std::string &filename : filenames) {
        auto t0 = clock::now();
        std::cout << "Parsing and constructing scene " << filename << "." << std::endl;
        GdptSceneDesc *desc = nullptr;
        if (gdpt_parse_scene_film(filename.c_str(), film_w, film_h, &desc) != 0) {
            std::cerr << "terminate: " << gdpt_last_error() << std::endl;   // the reference dies on an uncaught fl_exception
            return 134;
        }
        if (desc->integrator != GDPT_INTEGRATOR_GRADPATH && desc->integrator != GDPT_INTEGRATOR_PATH) {
            std::cerr << "terminate: this build implements Integrator::GradPath and Integrator::Path (scene asks for another integrator)" << std::endl;
            return 134;
        }
        const bool sharded = multi.num_devices > 0 && desc->integrator == GDPT_INTEGRATOR_GRADPATH;
        GdptScene *scene = nullptr;
        GdptMulti *mscene = nullptr;
        if ((sharded ? gdpt_multi_create(desc, &multi, &mscene) : gdpt_scene_upload(desc, device, &scene)) != 0) {
            std::cerr << "terminate: " << gdpt_last_error() << std::endl;
            return 134;
        }
        auto t1 = clock::now();
        std::cout << "Done. Took " << std::chrono::duration<double>(t1 - t0).count() << " seconds." << std::endl;
        std::cout << "Rendering..." << std::endl;
        const int w = desc->camera.width, h = desc->camera.height;
        std::vector<double> image((size_t)w * h * 3);
        GdptRenderParams p{};
        p.spp = spp; p.rng_scheme = rng; p.shift_mode = shift; p.plan_rows = plan_rows;
        GdptRenderStats rs{};
        GdptPoissonStats ps{};
        GdptMultiStats ms{};
        // render() dispatches on the integrator (src/render.cpp:374-392)
        int rc;
        if (sharded) {
            rc = gdpt_multi_gradient_path_render(mscene, &p, alpha, image.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &rs, &ms);
            ps.solve_ms = ms.solve_ms; ps.solver = GDPT_SOLVER_DEFAULT;
        } else if (desc->integrator == GDPT_INTEGRATOR_PATH) rc = gdpt_path_render(scene, &p, image.data(), &rs);
        else rc = gdpt_gradient_path_render(scene, &p, alpha, image.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &rs, &ps);
        if (rc != 0) {
            std::cerr << "terminate: " << gdpt_last_error() << std::endl;
            return 134;
        }
        // the reference prints one progress line per finished tile and a final 100% line (src/progress_reporter.h:22-29)
        unsigned long long tiles = (unsigned long long)((w + 15) / 16) * ((h + 15) / 16);
        std::fprintf(stdout, "\r %.2f Percent Done (%llu / %llu)\n", 100.0, tiles, tiles);
        if (outputfile.compare("") == 0) outputfile = desc->output_filename;
        auto t2 = clock::now();
        std::cout << "Done. Took " << std::chrono::duration<double>(t2 - t1).count() << " seconds." << std::endl;
        if (gdpt_imwrite(outputfile.c_str(), w, h, image.data()) != 0) {
            std::cerr << "terminate: " << gdpt_last_error() << std::endl;
            return 134;
        }
        std::cout << "Image written to " << outputfile << std::endl;
        std::cout << "[gdpt] " << rs.samples << " samples, " << rs.rays << " rays, render " << rs.render_ms << " ms ("
                  << (rs.render_ms > 0 ? rs.samples / rs.render_ms / 1e3 : 0.0) << " Msamples/s), Poisson " << ps.iterations
                  << " CG iterations " << ps.solve_ms << " ms, non-finite samples " << rs.nonfinite_samples << std::endl;
        if (sharded) {
            std::cout << "[gdpt] " << ms.num_devices << " row bands (" << (ms.exchange == GDPT_EXCHANGE_RCCL ? "RCCL" : "peer copies") << "): render";
            for (int k = 0; k < ms.num_devices; k++) std::cout << " " << ms.render_ms[k];
            std::cout << " ms, halo+assemble+gather " << ms.exchange_ms << " ms, solve " << ms.solve_ms << " ms, wall " << ms.wall_ms << " ms" << std::endl;
        }
        gdpt_multi_free(mscene);
        gdpt_scene_free(scene);
        gdpt_free_scene_desc(desc);
    }
    return 0;
}
