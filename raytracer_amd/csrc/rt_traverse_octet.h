// rt_traverse_octet.h -- octree traversal with eight lanes per ray, for gfx950 (device only).
//
// The per-lane form (rt_traverse_lane.h) gives a ray one lane: a descent step slab-tests the 8 octants of a node one
// after the other and a wave needs as many steps as its slowest ray.  That is the right trade when a launch has many
// more rays than the chip has lanes.  A launch of a few ten thousand incoherent rays (deep recursion levels) is
// different: it leaves most SIMDs with one wave or none, and its duration is the latency of one wave's chain of steps.
// Here a ray owns 8 neighbouring lanes (a wave carries 8 rays): lane c of the group slab-tests octant c of the current
// node, or triangle c, c + 8, ... of the current leaf.  A step is a fraction of the instructions, a wave waits for
// the slowest of 8 rays instead of 64, and the launch has 8 times as many waves to hide memory latency with.
//
// Candidate set, arithmetic and tie rule are the same as in rt_traverse.h / rt_traverse_lane.h: every leaf reached
// through passing slab tests contributes all its triangle references; of equal t the smaller global leaf slot wins
// (each lane keeps the best of the triangles it tested; the group's result is the lexicographic minimum of (t, slot)).
//
// All lanes of a group carry the same ray and the same traversal state; the stack (one entry per tree level) is in
// LDS, one column per group, written by all 8 lanes with the same value.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_traverse.h"
#include "rt_traverse_lane.h"

namespace rt {

#ifdef RT_LANE_PROFILE
// [ANY][k]: 0 cycles, 1 iterations, 2 iterations with a descent step, 3 groups descending (sum), 4 iterations with a leaf step,
// 5 groups in a leaf step (sum), 6 waves, 7 triangles tested
__device__ unsigned long long g_octet_prof[2][8];
#define RT_OPROF_ADD(k, v) do { if (__lane_id() == 0) atomicAdd(&g_octet_prof[ANY ? 1 : 0][k], (unsigned long long)(v)); } while (0)
#endif

struct OctetStackLds {  // of one wave
    uint32_t e[kLaneMaxLevels][8];
};

// Slab test of octant (lane & 7) of node `rec` (collision_detection.h:71-86); returns the 8-bit set of the group's
// octants that are non-empty and pass.  All lanes of a group must call it together.
__device__ __forceinline__ uint32_t octet_node_pass(const NodeRec* __restrict__ nodes, uint32_t rec, f3 o, f3 inv) {
    const float4* p = reinterpret_cast<const float4*>(nodes + rec);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
    const uint32_t nonempty = reinterpret_cast<const uint32_t*>(p)[12];
    const uint32_t lane = __lane_id();
    const bool bx = (lane & 4u) != 0, by = (lane & 2u) != 0, bz = (lane & 1u) != 0;
    // NodeRec: lo[axis][bit] = q0.x q0.y | q0.z q0.w | q1.x q1.y, hi[axis][bit] = q1.z q1.w | q2.x q2.y | q2.z q2.w
    const float lo[3] = {bx ? q0.y : q0.x, by ? q0.w : q0.z, bz ? q1.y : q1.x};
    const float hi[3] = {bx ? q1.w : q1.z, by ? q2.y : q2.x, bz ? q2.w : q2.z};
    float t_min = -RT_INF, t_max = RT_INF;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float oo = comp(o, ax), ii = comp(inv, ax);
        float t0 = (lo[ax] - oo) * ii;
        float t1 = (hi[ax] - oo) * ii;
        if (ii < 0.0f) { const float s = t0; t0 = t1; t1 = s; }
        t_min = std_max(t_min, t0);
        t_max = std_min(t_max, t1);
    }
    const lanemask pass = ballot64(t_max >= t_min);
    return (uint32_t)(pass >> (lane & 56u)) & 0xffu & nonempty;
}

// valid, o, d, t_limit: the same in the 8 lanes of a group.  On return h holds the group's result in all its lanes;
// for ANY, h.found has the bits of all lanes of the groups whose ray is occluded.
template <bool ANY, bool CULL>
__device__ __forceinline__ void traverse_octet(const DevOctree& T, OctetStackLds& stk, bool valid, f3 o, f3 d, float t_limit, HitState& h) {
    h.t = RT_INF;
    h.slot = 0xffffffffu;
    h.tri = 0xffffffffu;
    h.found = 0;
    if (T.root_info1 == 0 || ballot64(valid) == 0) return;  // empty tree / no rays
    const uint32_t lane = __lane_id(), sub = lane & 7u, grp = lane >> 3, shift = lane & 56u;
    const f3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // collision_detection.h:76
    bool active = valid;
    bool found = false;
    uint32_t leaf_slot = 0, leaf_left = 0;
    uint32_t cur_rec = 0, cur_pending = 0;
    int sp = 0;
    if (!(T.root_info1 & kChildInternal)) {  // the root is a leaf: all triangles, no box test (octree.cc:56-63,186-191)
        leaf_slot = T.root_info0;
        leaf_left = active ? T.root_info1 : 0u;
    } else {
        cur_rec = T.root_info0;
        if (active) cur_pending = octet_node_pass(T.nodes, cur_rec, o, inv);
    }
#ifdef RT_LANE_PROFILE
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
    unsigned long long pit = 0, pd = 0, pdg = 0, pl = 0, plg = 0, ptri = 0;
#endif
    while (ballot64(active) != 0) {
#ifdef RT_LANE_PROFILE
        {
            const lanemask md = ballot64(active && leaf_left == 0);
            ++pit; pd += md != 0; pdg += __builtin_popcountll(md) / 8;
        }
#endif
        if (active && leaf_left == 0) {  // one descent step
            if (cur_pending == 0) {
                if (sp == 0) {
                    active = false;
                } else {
                    --sp;
                    const uint32_t e = stk.e[sp][grp];
                    cur_rec = e >> 8;
                    cur_pending = e & 0xffu;
                }
            } else {
                const uint32_t c = (uint32_t)__builtin_ctz(cur_pending);
                cur_pending &= cur_pending - 1;
                const uint2 info = *reinterpret_cast<const uint2*>(&T.nodes[cur_rec].info[c][0]);
                if (info.y & kChildInternal) {
                    if (cur_pending != 0) {
                        stk.e[sp][grp] = (cur_rec << 8) | cur_pending;
                        ++sp;
                    }
                    cur_rec = info.x;
                    cur_pending = octet_node_pass(T.nodes, cur_rec, o, inv);
                } else {
                    leaf_slot = info.x;
                    leaf_left = info.y;
                }
            }
        }
#ifdef RT_LANE_PROFILE
        {
            const lanemask ml = ballot64(active && leaf_left != 0);
            pl += ml != 0; plg += __builtin_popcountll(ml) / 8;
            ptri += __builtin_popcountll(ballot64(active && leaf_left != 0 && sub < leaf_left));
        }
#endif
        if (active && leaf_left != 0) {  // up to 8 triangles of the leaf
            const bool on = sub < leaf_left;
            const uint32_t slot = leaf_slot + sub;
            const float4* r = reinterpret_cast<const float4*>(T.slots + (on ? slot : leaf_slot));
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            const f3 a = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
            float t = RT_INF;
            const lanemask hit = ray_triangle<CULL>(ballot64(on), o, d, a, e1, e2, t);
            const uint32_t step = leaf_left < 8u ? leaf_left : 8u;
            leaf_slot += step;
            leaf_left -= step;
            if (ANY) {
                const lanemask good = hit & ballot64(t < t_limit);  // renderer.cc:302
                if (((good >> shift) & 0xffull) != 0) {  // this ray is decided
                    found = true;
                    active = false;
                    leaf_left = 0;
                }
            } else if (lane_of(hit)) {
                // renderer.cc:130: strictly smaller t wins; equal t keeps the smaller global slot
                const bool b = t < h.t || (t == h.t && slot < h.slot);
                h.t = b ? t : h.t;
                h.slot = b ? slot : h.slot;
                h.tri = b ? __float_as_uint(r2.y) : h.tri;
            }
        }
    }
#ifdef RT_LANE_PROFILE
    RT_OPROF_ADD(0, __builtin_amdgcn_s_memtime() - pt0); RT_OPROF_ADD(1, pit); RT_OPROF_ADD(2, pd); RT_OPROF_ADD(3, pdg);
    RT_OPROF_ADD(4, pl); RT_OPROF_ADD(5, plg); RT_OPROF_ADD(6, 1); RT_OPROF_ADD(7, ptri);
#endif
    if (ANY) {
        h.found = ballot64(found);
    } else {
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            const float t2 = __shfl_xor(h.t, m, 64);
            const uint32_t s2 = __shfl_xor(h.slot, m, 64), tri2 = __shfl_xor(h.tri, m, 64);
            const bool b = t2 < h.t || (t2 == h.t && s2 < h.slot);
            h.t = b ? t2 : h.t;
            h.slot = b ? s2 : h.slot;
            h.tri = b ? tri2 : h.tri;
        }
    }
}

}  // namespace rt
