#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "raytracer_amd.h"
#define CHECK(x) do { int rc_ = (x); if (rc_) { std::printf("FAIL %s -> %d: %s\n", #x, rc_, rt_last_error()); return 1; } } while (0)
int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-5.f, 5.f), S(0.01f, 2.f);
    for (int n : {0, 1, 5, 17, 300, 5000}) {
        std::vector<float> pos((size_t)n * 9);
        for (int i = 0; i < n; ++i) {
            float c[3] = {U(rng), U(rng), U(rng)}, s = S(rng);
            for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) pos[(size_t)i * 9 + v * 3 + k] = c[k] + s * U(rng) * 0.2f;
        }
        for (auto par : {std::pair<int,int>{16, 10}, {1, 6}, {4, 0}, {1000, 3}}) {
            rt_host_tree* t = nullptr;
            CHECK(rt_host_tree_build(pos.data(), n, par.first, par.second, &t));
            uint64_t nn, np, nr, ns; uint32_t ms;
            CHECK(rt_host_tree_sizes(t, &nn, &np, &nr, &ns, &ms));
            std::vector<uint8_t> leaf(nn); std::vector<uint32_t> depth(nn), cnt(nn), pool(np); std::vector<float> bmin(nn * 3), bmax(nn * 3);
            std::vector<int32_t> child(nn * 8); std::vector<uint64_t> beg(nn);
            CHECK(rt_host_tree_export(t, leaf.data(), depth.data(), bmin.data(), bmax.data(), child.data(), beg.data(), cnt.data(), pool.data()));
            std::vector<char> recs(nr * 128), slots(ns * 48); uint32_t root[2];
            CHECK(rt_host_tree_export_flat(t, recs.data(), slots.data(), root));
            rt_tree_stats st; CHECK(rt_host_tree_stats(t, &st));
            rt_host_tree_destroy(t);
        }
    }
    // scene IO
    const char* obj = "/tmp/rt_host_sanitize/t.obj"; const char* mtl = "/tmp/rt_host_sanitize/t.mtl";
    { FILE* f = fopen(mtl, "w"); fputs("newmtl a\nKd 1 0 0\nKe 1 1 1\nnewmtl b\nNs 5\n", f); fclose(f); }
    { FILE* f = fopen(obj, "w"); fputs("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nusemtl a\nf 1 2 3 4\nusemtl b\nf -4//1 -3//1 -2//1\nf 1/1/1 2/2/1 3\n#c\n\n", f); fclose(f); }
    rt_scene_data* sd = nullptr;
    CHECK(rt_scene_load_obj(obj, "/tmp/rt_host_sanitize/", 1, &sd));
    uint64_t nt; uint32_t nm; int hn; CHECK(rt_scene_data_sizes(sd, &nt, &nm, &hn));
    std::vector<float> p(nt * 9), nr(nt * 9), mats((size_t)nm * 18); std::vector<int32_t> mid(nt);
    CHECK(rt_scene_data_export(sd, p.data(), nr.data(), mid.data(), mats.data()));
    rt_scene_data_free(sd);
    if (rt_scene_load_obj("/tmp/rt_host_sanitize/missing.obj", "/tmp/rt_host_sanitize/", 1, &sd) == 0) { std::puts("missing file accepted"); return 1; }
    { FILE* f = fopen("/tmp/rt_host_sanitize/bad.obj", "w"); fputs("v 0 0 0\nf 1 2 9\nf 1 x\n", f); fclose(f); }
    if (rt_scene_load_obj("/tmp/rt_host_sanitize/bad.obj", "/tmp/rt_host_sanitize/", 1, &sd) == 0) { std::puts("bad file accepted"); return 1; }
    std::vector<float> img(13 * 7 * 3, 0.5f);
    CHECK(rt_write_ppm("/tmp/rt_host_sanitize/o.ppm", 13, 7, img.data()));
    int kind; CHECK(rt_ads_kind_from_name("octree_gpu", &kind));
    if (rt_ads_kind_from_name("nope", &kind) == 0) return 1;
    std::printf("host paths clean (%llu triangles loaded, %u materials)\n", (unsigned long long)nt, nm);
    return 0;
}
