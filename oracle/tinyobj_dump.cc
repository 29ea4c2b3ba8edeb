// tinyobj_dump.cc -- TEST INFRASTRUCTURE.  Compiled (by oracle/Makefile, only where the reference checkout exists)
// against the reference's vendored OBJ parser, unmodified: loads an OBJ file the way the reference's LoadScene does
// (src/ObjLoader/obj_loader.cc:144, triangulate on) and prints every vertex coordinate as the 8 hex digits of its
// float.  Used to pin this build's own decimal parser (raytracer_amd/csrc/scene_io.cpp) to the reference's on
// literals where a correctly rounded strtod would give another float.
// Usage: tinyobj_dump <file.obj> <basepath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define TINYOBJLOADER_IMPLEMENTATION
#include "include/tiny_obj_loader.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, argv[1], argv[2], true)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    for (float v : attrib.vertices) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        std::printf("%08x\n", u);
    }
    return 0;
}
