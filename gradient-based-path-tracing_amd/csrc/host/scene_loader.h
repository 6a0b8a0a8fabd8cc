// scene_loader.h — Mitsuba-0.x XML subset -> flattened GdptSceneDesc (host memory).
// Host-side mirror of parse_scene() (reference: src/parsers/parse_scene.cpp:1405-1630).
#pragma once
#include "../../../include/gdpt.h"
#include "host_math.h"
#include <deque>
#include <memory>
#include <string>
#include <vector>

namespace gdpt {

struct HostMesh {
    std::vector<double> positions; // 3*nv
    std::vector<int32_t> indices;  // 3*nt
    std::vector<double> normals;   // 3*nv or empty
    std::vector<double> uvs;       // 2*nv or empty
};

// Owns every array a GdptSceneDesc points into.
struct HostScene {
    GdptSceneDesc desc{};
    std::vector<GdptMaterial> materials;
    std::vector<GdptShape> shapes;
    std::vector<GdptLight> lights;
    std::vector<GdptImage> images;
    std::deque<HostMesh> meshes;                 // stable addresses
    std::deque<std::vector<double>> image_data;  // stable addresses
    void finalize();                             // (re)build desc pointers/counts
};

// Throws std::runtime_error with the reference's wording where it has one (src/flexception.h).
// film_width / film_height > 0 replace the extent of the scene's <film> (the camera is built for the new aspect).
std::unique_ptr<HostScene> load_scene_xml(const std::string &path, int film_width = 0, int film_height = 0);

// Pieces exposed for unit tests / golden checks against the reference's own functions.
HostMesh load_obj(const std::string &filename, const M4 &to_world);          // src/parsers/parse_obj.cpp:94-185
HostMesh load_serialized(const std::string &filename, int shape_index, const M4 &to_world); // src/parsers/load_serialized.cpp:179-256
std::vector<double> compute_vertex_normals(const std::vector<double> &positions,
                                           const std::vector<int32_t> &indices);           // src/parsers/shape_utils.h:15-49
V3 spectrum_to_rgb(const std::vector<std::pair<double, double>> &spec);                     // src/spectrum.h:72-118
void make_camera(const M4 &cam_to_world, double fov_deg, int width, int height,
                 int filter_type, double filter_param, GdptCamera *out);                    // src/camera.cpp:7-21

} // namespace gdpt
