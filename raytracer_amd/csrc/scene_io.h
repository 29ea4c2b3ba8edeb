// scene_io.h -- host-side scene input (OBJ/MTL) and image output (P3 PPM); see scene_io.cpp.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace rt {

struct SceneData {
    std::vector<float> positions;      // n x 3 vertices x 3
    std::vector<float> normals;        // n x 3 vertices x 3 (zeros where the file has none)
    std::vector<int32_t> material_ids; // n
    std::vector<float> materials;      // m x 18: ambient, diffuse, specular, transmittance, emission, shininess, ior, dissolve
    bool has_normals = false;
};

bool load_obj(const std::string& obj_path, const std::string& basepath, bool triangulate, SceneData& out, std::string& err);
bool write_ppm(const std::string& path, int width, int height, const float* rgb, std::string& err);

}  // namespace rt
