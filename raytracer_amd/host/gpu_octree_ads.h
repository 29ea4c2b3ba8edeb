// gpu_octree_ads.h -- the MI355X backend as a subclass of the reference renderer's `Ads`.
//
// This header is meant to be compiled INSIDE the reference tree (it includes the reference's own
// src/ADS/ads.h, src/ray.h and src/triangle.h) and linked against libraytracer_amd.so.  It follows the
// reference's extension recipe (README.md:136-144 of the reference): subclass Ads, override Build /
// Search / PrintStats, add an enum value to Renderer::ACCELERATION_DATA_STRUCTURE, a name to
// Renderer::config_setup and a case to Renderer::setup_ads.
//
// The single-ray Search keeps the reference's contract exactly (same candidates, same order), so the
// unmodified renderer loops work on top of it; the batched TraceClosest / TraceAny calls are what a
// renderer should use to get the GPU's throughput.
#pragma once

#include <cstdint>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "include/json.hpp"

#include "src/ADS/ads.h"
#include "src/ray.h"
#include "src/triangle.h"

#include "raytracer_amd.h"

namespace raytracer {

class GpuOctree : public Ads {
   public:
    // config: the "acceleratied_data_structure" object (renderer.cc:72-76).  "max_triangles_per_BB" and
    // "max_depth" are required like for Octree (octree.cc:271-272); "device" is optional.
    explicit GpuOctree(const nlohmann::json& config) : Ads(config) {
        rt_ads_config c{};
        c.struct_size = sizeof c;
        c.max_triangles_per_bb = config.at("max_triangles_per_BB");
        c.max_depth = config.at("max_depth");
        c.device = config.contains("device") ? static_cast<int>(config.at("device")) : 0;
        if (config.contains("builder")) {  // optional: where Build runs; both builders give the same tree
            const std::string b = config.at("builder");
            if (b == "host") c.builder = RT_BUILDER_HOST;
            else if (b == "device") c.builder = RT_BUILDER_DEVICE;
            else if (b != "auto") throw std::runtime_error("Invalid builder");
        }
        check(rt_ads_create(&c, &handle_));
    }
    ~GpuOctree() override { rt_ads_destroy(handle_); }
    GpuOctree(const GpuOctree&) = delete;
    GpuOctree& operator=(const GpuOctree&) = delete;

    void Build(const std::vector<std::shared_ptr<const Triangle>>& triangles) override {
        Ads::Build(triangles);  // keeps the shared_ptr copies that Search hands back (ads.h:21,27)
        std::vector<float> pos(triangles_.size() * 9);
        for (size_t i = 0; i < triangles_.size(); ++i)
            for (int v = 0; v < 3; ++v) {
                pos[i * 9 + v * 3 + 0] = triangles_[i]->vertices[v].pos.x;
                pos[i * 9 + v * 3 + 1] = triangles_[i]->vertices[v].pos.y;
                pos[i * 9 + v * 3 + 2] = triangles_[i]->vertices[v].pos.z;
            }
        check(rt_ads_build(handle_, pos.data(), triangles_.size()));
    }

    std::vector<std::shared_ptr<const Triangle>> Search(const Ray& r, bool /*first_hit*/ = false) const override {
        const float ray[6] = {r.origin().x, r.origin().y, r.origin().z, r.direction().x, r.direction().y, r.direction().z};
        uint64_t offsets[2] = {0, 0}, total = 0;
        scratch_.resize(scratch_.empty() ? 256 : scratch_.size());
        int rc = rt_ads_search(handle_, ray, 1, offsets, scratch_.data(), scratch_.size(), &total);
        if (rc == RT_ERR_CAPACITY) {
            scratch_.resize(total);
            rc = rt_ads_search(handle_, ray, 1, offsets, scratch_.data(), scratch_.size(), &total);
        }
        check(rc);
        std::vector<std::shared_ptr<const Triangle>> out;
        out.reserve(total);
        for (uint64_t i = 0; i < total; ++i) out.emplace_back(triangles_[scratch_[i]]);
        return out;
    }

    void PrintStats(std::ostream& os) const override {  // octree.cc:209-224
        rt_tree_stats t{};
        rt_query_stats q{};
        check(rt_ads_tree_stats(handle_, &t));
        check(rt_ads_query_stats(handle_, &q));
        os << "Octree stats: " << "\n";
        os << "  Max depth: " << t.max_depth << "\n";
        os << "  Nodes count: " << t.nodes << "\n";
        os << "  Leaf nodes count: " << t.leaves << "\n";
        os << "  Average depth of leaf nodes: " << t.avg_leaf_depth << "\n";
        os << "  Max triangles in leaf nodes: " << t.max_triangles_in_leaf << "\n";
        os << "  Average triangles in leaf nodes: " << t.avg_triangles_in_leaf << "\n";
        os << "  Search method call count: " << q.search_calls << "\n";
        os << "  Search node count: " << q.search_node_count << "\n";
        os << "  Search time: " << (float)q.search_seconds << " s" << "\n";
        os << "  Search return count: " << q.search_return_count << "\n";
        os << "  Average search return count: " << (float)q.search_return_count / q.search_calls << "\n";
        os << "  Search leaves visited: " << q.search_leaves_visited << "\n";
    }
    // Renderer::RenderScene prints this one itself (renderer.cc:60); a renderer on top of the batched calls reads it here.
    uint64_t RayTriangleCollisionCount() const {
        rt_query_stats q{};
        check(rt_ads_query_stats(handle_, &q));
        return q.ray_triangle_tests;
    }
    // The batched calls count nodes / leaves / candidates / triangle tests in the reference's definitions only while this is on.
    void CollectSearchStats(bool on) { check(rt_ads_collect_search_stats(handle_, on ? 1 : 0)); }

    // Batched forms of the renderer's two hot loops (renderer.cc:116-142 and :288-313).  n rays, arrays of
    // 3 floats per ray; results as documented in raytracer_amd.h.
    void TraceClosest(const float* origins, const float* dirs, uint64_t n, bool cull_backfaces, float* out_t, int32_t* out_tri) const {
        check(rt_ads_trace_closest(handle_, origins, dirs, n, cull_backfaces, RT_MEM_HOST, out_t, out_tri, nullptr));
    }
    void TraceAny(const float* points, const float* lights, uint64_t n, bool cull_backfaces, uint8_t* out_shadowed) const {
        check(rt_ads_trace_any(handle_, points, lights, n, cull_backfaces, RT_MEM_HOST, out_shadowed, nullptr));
    }
    rt_ads* handle() const { return handle_; }

   private:
    static void check(int rc) {
        if (rc != RT_OK) throw std::runtime_error(std::string("GpuOctree: ") + rt_last_error());  // main.cc:31-34 prints it
    }
    rt_ads* handle_ = nullptr;
    mutable std::vector<uint32_t> scratch_;
};

}  // namespace raytracer
