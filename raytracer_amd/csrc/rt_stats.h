// rt_stats.h -- the reference's Search / ray-triangle counters, computed on the device (device only).
//
// The fast kernels (rt_traverse*.h) visit nodes depth-first, share work between the rays of a wave, drop duplicate
// triangles and leave an any-hit ray at its first occluder, so what they execute is not what the reference's counters
// count.  This file walks every ray on its own, completely, and counts in the reference's definitions:
//
//   nodes    : nodes taken from Search's queue, the root included (search_node_count_, octree.cc:185)
//   leaves   : leaves among them (search_leaves_visited_, octree.cc:187)
//   returned : length of the candidate list (search_return_count_, octree.cc:204)
//   slab     : collision_ray_aabb calls = non-null, non-empty octants of the internal nodes taken (octree.cc:195)
//   tri      : collision_ray_triangle calls of the loop over the candidates: all of them for a closest-hit ray
//              (renderer.cc:137); for an any-hit ray up to and including the first occluder IN LIST ORDER
//              (renderer.cc:305), all if there is none (renderer.cc:311)
//
// The first four do not depend on the order of the walk.  For the last one the list order matters: the candidate
// list is the concatenation of the visited leaves in breadth-first order, and leaf slots are laid out in exactly that
// order, so the list is ascending in global slot index and the position of a candidate in it is the number of
// candidates with a smaller slot.  An any-hit ray therefore takes two walks: the first finds the smallest slot among
// its occluders, the second counts the candidates up to that slot.
//
// One ray per lane, per-lane stack of (record, pending octants) per tree level in private memory.  This is a
// measuring pass, not a fast path; results of the queries never come from here.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_traverse.h"
#include "rt_traverse_lane.h"

namespace rt {

// Totals of the instrumented pass; any-hit rays add to the first block AND to the second one (offset QS_ANY), which
// mirrors the reference-work split that a per-kernel roofline needs.
enum { QS_NODES = 0, QS_LEAVES = 1, QS_RETURNED = 2, QS_SLAB = 3, QS_TRI = 4, QS_RAYS = 5, QS_ANY = 8, QS_WORDS = 16 };

struct RayCounters {
    uint32_t nodes = 0, leaves = 0, returned = 0, slab = 0, tri = 0, rays = 0;
};

// collision_detection.h:12-68 for one lane: true iff the triangle is hit at a usable distance (the callers' filter
// `t == infinity || t < epsilon`, renderer.cc:126 / :298, included); same operations and order as ray_triangle<>.
template <bool CULL>
__device__ __forceinline__ bool ray_triangle_lane(f3 o, f3 s, f3 a, f3 e1, f3 e2, float& t) {
    const f3 p = cross(s, e2);
    const float d = dot(e1, p);
    if (CULL ? (d < kEps) : (fabsf(d) < kEps)) return false;
    const float d_inv = 1.0f / d;
    const f3 q = o - a;
    const float u = d_inv * dot(q, p);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 w = cross(q, e1);
    const float v = d_inv * dot(w, s);
    if (v < 0.0f || (u + v) > 1.0f) return false;
    t = d_inv * dot(e2, w);
    return !(t == RT_INF || t < kEps);
}

// One complete walk.  PASS 0: counts nodes / leaves / returned / slab and (ANY) finds the smallest occluding slot.
// PASS 1 (ANY only): counts the candidates whose slot is <= first_occluder.
template <bool ANY, bool CULL, int PASS>
__device__ __forceinline__ void stats_walk(const DevOctree& T, f3 o, f3 d, f3 inv, float t_limit, RayCounters& c, uint32_t& first_occluder) {
    uint32_t stack[kLaneMaxLevels];
    int sp = 0;
    auto leaf = [&](uint32_t begin, uint32_t count) {
        if (PASS == 0) {
            ++c.leaves;
            c.returned += count;
            if (ANY) {
                for (uint32_t k = 0; k < count; ++k) {
                    if (begin + k >= first_occluder) break;  // a later slot cannot lower the minimum
                    const LeafTri g = T.slots[begin + k];
                    float t = RT_INF;
                    if (ray_triangle_lane<CULL>(o, d, mk(g.ax, g.ay, g.az), mk(g.e1x, g.e1y, g.e1z), mk(g.e2x, g.e2y, g.e2z), t) && t < t_limit) {
                        first_occluder = begin + k;  // renderer.cc:302
                        break;
                    }
                }
            }
        } else {
            if (begin <= first_occluder) {
                const uint32_t upto = first_occluder - begin + 1u;
                c.tri += upto < count ? upto : count;
            }
        }
    };
    if (T.root_info1 == 0) return;  // empty tree: no queue, no candidates
    if (PASS == 0) ++c.nodes;       // the root
    if (!(T.root_info1 & kChildInternal)) {  // the root is a leaf (octree.cc:56-63,186-191)
        leaf(T.root_info0, T.root_info1);
        return;
    }
    uint32_t rec = T.root_info0;
    const uint32_t flip = ray_flip(inv);  // (lane_node_pass numbers the octants in the ray's own order)
    uint32_t pending = lane_node_pass<false>(T.nodes, rec, o, inv, flip, -RT_INF, RT_INF, nullptr);
    if (PASS == 0) c.slab += (uint32_t)__builtin_popcount(T.nodes[rec].nonempty);
    for (;;) {
        if (pending == 0) {
            if (sp == 0) break;
            --sp;
            rec = stack[sp] >> 8;
            pending = stack[sp] & 0xffu;
            continue;
        }
        const uint32_t oct = (uint32_t)__builtin_ctz(pending);
        pending &= pending - 1u;
        const uint32_t i0 = T.nodes[rec].info[oct ^ flip][0], i1 = T.nodes[rec].info[oct ^ flip][1];
        if (PASS == 0) ++c.nodes;
        if (i1 & kChildInternal) {
            if (pending != 0 && sp < kLaneMaxLevels) stack[sp++] = (rec << 8) | pending;
            rec = i0;
            pending = lane_node_pass<false>(T.nodes, rec, o, inv, flip, -RT_INF, RT_INF, nullptr);
            if (PASS == 0) c.slab += (uint32_t)__builtin_popcount(T.nodes[rec].nonempty);
        } else {
            leaf(i0, i1);
        }
    }
}

template <bool ANY, bool CULL>
__device__ __forceinline__ void stats_ray(const DevOctree& T, bool valid, f3 o, f3 d, float t_limit, RayCounters& c) {
    if (!valid) return;
    ++c.rays;
    const f3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // collision_detection.h:76
    uint32_t first = 0xffffffffu;
    const uint32_t before = c.returned;
    stats_walk<ANY, CULL, 0>(T, o, d, inv, t_limit, c, first);
    if (!ANY || first == 0xffffffffu) c.tri += c.returned - before;
    else stats_walk<ANY, CULL, 1>(T, o, d, inv, t_limit, c, first);
}

// Adds the lanes' counters to the 64-bit totals at `out` (QS_* order): one atomic per counter and wave.
template <bool ANY = false>
__device__ __forceinline__ void stats_flush(const RayCounters& c, unsigned long long* __restrict__ out) {
    uint32_t v[6] = {c.nodes, c.leaves, c.returned, c.slab, c.tri, c.rays};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned long long x = v[k];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
        if (__lane_id() == 0 && x) {
            atomicAdd(&out[k], x);
            if (ANY) atomicAdd(&out[QS_ANY + k], x);
        }
    }
}

}  // namespace rt
