// capi_host.cpp — host-only entry points of include/gdpt.h (scene ingest, image output, errors).
#include "../../include/gdpt.h"
#include "capi_common.h"
#include "host/image_io.h"
#include "host/scene_loader.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace gdpt {
static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
} // namespace gdpt

namespace {
std::mutex g_mu;
std::map<GdptSceneDesc *, std::unique_ptr<gdpt::HostScene>> g_descs; // desc pointer -> owner
} // namespace

extern "C" {

const char *gdpt_last_error(void) { return gdpt::g_last_error.c_str(); }

int gdpt_parse_scene(const char *xml_path, GdptSceneDesc **out_desc) {
    return gdpt::guarded([&]() {
        if (!xml_path || !out_desc) throw std::runtime_error("gdpt_parse_scene: null argument");
        std::unique_ptr<gdpt::HostScene> hs = gdpt::load_scene_xml(xml_path);
        GdptSceneDesc *d = &hs->desc;
        std::lock_guard<std::mutex> lk(g_mu);
        g_descs[d] = std::move(hs);
        *out_desc = d;
    });
}

void gdpt_free_scene_desc(GdptSceneDesc *desc) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_descs.erase(desc);
}

int gdpt_imwrite(const char *filename, int width, int height, const double *rgb) {
    return gdpt::guarded([&]() {
        if (!filename || !rgb || width <= 0 || height <= 0) throw std::runtime_error("gdpt_imwrite: bad argument");
        gdpt::write_image(filename, width, height, rgb);
    });
}

} // extern "C"
