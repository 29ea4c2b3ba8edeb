// dropin_test.cc -- TEST INFRASTRUCTURE.  Compiled (by oracle/Makefile, only where the reference checkout
// exists) together with the reference's own Octree sources and linked against libraytracer_amd.so:
// it instantiates the reference's `Octree` and this build's `GpuOctree` (raytracer_amd/host/gpu_octree_ads.h)
// behind the reference's `Ads` interface, feeds both the same triangles and rays, and requires
//   * Search() to return the same shared_ptr<const Triangle> sequence, element for element,
//   * PrintStats() of the two, after those identical Search calls, to print the same text line for line
//     (octree.cc:211-223; the "Search time" line, a wall-clock figure, excepted),
//   * the reference's closest-hit loop over its candidates (renderer.cc:123-134) to agree with the
//     batched TraceClosest on the winning triangle and on t bit for bit, and
//   * the reference's any-hit loop (Renderer::is_shadowed, renderer.cc:288-313, restated over the reference's
//     own Search and collision_ray_triangle) to agree with the batched TraceAny on every flag, with and
//     without back-face culling; the segments start ON the surfaces the closest hits found, like the renderer's.
// Usage: ads_dropin_test <scene.raw> [n_rays]      scene.raw = uint64 n, then n*9 float32 positions.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "include/json.hpp"

#include "src/ADS/Octree/octree.h"
#include "src/ADS/ads.h"
#include "src/collision_detection.h"
#include "src/ray.h"
#include "src/triangle.h"

#include "gpu_octree_ads.h"

using namespace raytracer;

static uint32_t lcg(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}
static float unit(uint32_t& s) { return (float)lcg(s) / (float)(1u << 24); }

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " <scene.raw> [n_rays]\n";
        return 2;
    }
    const int n_rays = argc > 2 ? atoi(argv[2]) : 2000;
    std::ifstream f(argv[1], std::ios::binary);
    uint64_t n = 0;
    f.read(reinterpret_cast<char*>(&n), sizeof n);
    std::vector<float> pos(n * 9);
    f.read(reinterpret_cast<char*>(pos.data()), (std::streamsize)(pos.size() * sizeof(float)));
    if (!f) {
        std::cerr << "cannot read " << argv[1] << "\n";
        return 2;
    }
    std::vector<std::shared_ptr<const Triangle>> tris;
    Vec3 lo(1e30f), hi(-1e30f);
    for (uint64_t i = 0; i < n; ++i) {
        Triangle t;
        for (int v = 0; v < 3; ++v) {
            t.vertices[v].pos = Vec3(pos[i * 9 + v * 3], pos[i * 9 + v * 3 + 1], pos[i * 9 + v * 3 + 2]);
            const Vec3& p = t.vertices[v].pos;
            lo = Vec3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
            hi = Vec3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
        }
        t.material_id = 0;
        t.normal = calculate_triangle_normal(t);
        tris.emplace_back(std::make_shared<const Triangle>(t));
    }
    nlohmann::json cfg = {{"name", "octree_gpu"}, {"max_triangles_per_BB", 16}, {"max_depth", 10}};
    try {
        std::unique_ptr<Ads> ref = std::make_unique<Octree>(cfg);
        std::unique_ptr<Ads> gpu = std::make_unique<GpuOctree>(cfg);
        ref->Build(tris);
        gpu->Build(tris);
        uint32_t seed = 12345;
        std::vector<float> origins, dirs;
        std::vector<float> ref_t;
        std::vector<int32_t> ref_tri;
        uint64_t total = 0;
        for (int r = 0; r < n_rays; ++r) {
            Vec3 o(lo.x + (hi.x - lo.x) * (unit(seed) * 1.4f - 0.2f), lo.y + (hi.y - lo.y) * (unit(seed) * 1.4f - 0.2f),
                   lo.z + (hi.z - lo.z) * (unit(seed) * 1.4f - 0.2f));
            Vec3 d(unit(seed) * 2 - 1, unit(seed) * 2 - 1, unit(seed) * 2 - 1);
            if (r % 9 == 0) d.x = 0.0f;
            if (r % 13 == 0) d.y = 0.0f;
            d = d.normalize();
            Ray ray(o, d);
            auto a = ref->Search(ray);
            auto b = gpu->Search(ray);
            if (a.size() != b.size()) {
                std::cerr << "ray " << r << ": candidate count " << a.size() << " vs " << b.size() << "\n";
                return 1;
            }
            for (size_t k = 0; k < a.size(); ++k)
                if (a[k].get() != b[k].get()) {
                    std::cerr << "ray " << r << ": candidate " << k << " differs\n";
                    return 1;
                }
            total += a.size();
            // the reference's closest-hit loop (renderer.cc:116-134) over the reference's candidates
            float t_pixel = infinity;
            int hit = -1;
            for (size_t k = 0; k < a.size(); ++k) {
                float t = collision_ray_triangle(ray, *a[k], true);
                if (t == infinity || t < epsilon) continue;
                if (t < t_pixel) {
                    t_pixel = t;
                    hit = (int)k;
                }
            }
            int32_t tri = -1;
            if (hit >= 0)
                for (size_t i = 0; i < tris.size(); ++i)
                    if (tris[i].get() == a[hit].get()) tri = (int32_t)i;
            ref_t.push_back(t_pixel);
            ref_tri.push_back(tri);
            origins.insert(origins.end(), {o.x, o.y, o.z});
            dirs.insert(dirs.end(), {d.x, d.y, d.z});
        }
        // PrintStats after identical Search calls (octree.cc:209-224): the same text, the wall-clock line aside
        {
            std::ostringstream sa, sb;
            ref->PrintStats(sa);
            gpu->PrintStats(sb);
            std::istringstream ia(sa.str()), ib(sb.str());
            std::string la, lb;
            int lines = 0;
            for (;;) {
                const bool ga = (bool)std::getline(ia, la), gb = (bool)std::getline(ib, lb);
                if (ga != gb) {
                    std::cerr << "PrintStats: different number of lines\n";
                    return 1;
                }
                if (!ga) break;
                ++lines;
                if (la.rfind("  Search time:", 0) == 0 && lb.rfind("  Search time:", 0) == 0) continue;
                if (la != lb) {
                    std::cerr << "PrintStats line " << lines << ": reference '" << la << "' vs '" << lb << "'\n";
                    return 1;
                }
            }
            if (lines != 13) {
                std::cerr << "PrintStats: " << lines << " lines, expected 13\n";
                return 1;
            }
        }
        std::vector<float> t(n_rays);
        std::vector<int32_t> tri(n_rays);
        static_cast<GpuOctree*>(gpu.get())->TraceClosest(origins.data(), dirs.data(), (uint64_t)n_rays, true, t.data(), tri.data());
        int hits = 0;
        for (int r = 0; r < n_rays; ++r) {
            if (tri[r] != ref_tri[r] || std::memcmp(&t[r], &ref_t[r], 4) != 0) {
                std::cerr << "ray " << r << ": closest hit (" << tri[r] << ", " << t[r] << ") vs reference (" << ref_tri[r] << ", " << ref_t[r] << ")\n";
                return 1;
            }
            hits += tri[r] >= 0;
        }
        // Renderer::is_shadowed (renderer.cc:288-313) over the reference's Search, against the batched TraceAny.
        // Surface points: where the closest hits landed (renderer.cc:146: origin + direction * t, no offset).
        uint64_t any_pairs = 0, any_shadowed = 0;
        for (int cull = 1; cull >= 0; --cull) {
            std::vector<float> points, lights;
            std::vector<uint8_t> want;
            for (int r = 0; r < n_rays; ++r) {
                if (ref_tri[r] < 0) continue;
                const Vec3 o(origins[r * 3], origins[r * 3 + 1], origins[r * 3 + 2]), d(dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2]);
                const Vec3 p = o + d * ref_t[r];
                const Vec3 l(lo.x + (hi.x - lo.x) * unit(seed), lo.y + (hi.y - lo.y) * unit(seed), lo.z + (hi.z - lo.z) * unit(seed));
                const float dist_light = (l - p).length();
                Ray shadow_ray(p, (l - p).normalize());
                auto cand = ref->Search(shadow_ray);
                bool shadowed = false;
                for (size_t k = 0; k < cand.size(); ++k) {
                    const float st = collision_ray_triangle(shadow_ray, *cand[k], cull != 0);
                    if (st == infinity || st < epsilon) continue;
                    if (st < dist_light) {
                        shadowed = true;
                        break;
                    }
                }
                points.insert(points.end(), {p.x, p.y, p.z});
                lights.insert(lights.end(), {l.x, l.y, l.z});
                want.push_back(shadowed ? 1 : 0);
            }
            std::vector<uint8_t> got(want.size());
            static_cast<GpuOctree*>(gpu.get())->TraceAny(points.data(), lights.data(), (uint64_t)want.size(), cull != 0, got.data());
            for (size_t k = 0; k < want.size(); ++k) {
                if (got[k] != want[k]) {
                    std::cerr << "segment " << k << " (cull " << cull << "): shadowed " << (int)got[k] << " vs reference " << (int)want[k] << "\n";
                    return 1;
                }
                any_shadowed += want[k];
            }
            any_pairs += want.size();
        }
        if (any_pairs == 0 || any_shadowed == 0 || any_shadowed == any_pairs) {
            std::cerr << "any-hit check is vacuous: " << any_shadowed << " of " << any_pairs << " segments shadowed\n";
            return 1;
        }
        gpu->PrintStats(std::clog);
        std::cout << "DROPIN PASS rays=" << n_rays << " segments=" << any_pairs << " shadowed=" << any_shadowed << " candidates=" << total << " hits=" << hits << std::endl;
    } catch (std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
