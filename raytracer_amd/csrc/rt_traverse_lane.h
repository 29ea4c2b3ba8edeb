// rt_traverse_lane.h -- per-lane octree traversal for gfx950 (device only).
//
// rt_traverse.h walks the tree once per WAVE: ideal when the 64 rays of a wave reach the same nodes, but its
// work is the UNION of what the 64 rays reach.  Rays that leave a finely subdivided surface towards 64 different
// random points of an area light share almost nothing, and the union is then tens of times one ray's work.
// Here every lane walks its own ray: its own descent, its own leaves, vector loads instead of scalar loads.
// The wave alternates between two phases.  Descent: lanes that need a leaf take one octant per step (either an
// internal node, whose 8 octants are slab-tested at once, or a leaf); the phase ends when enough lanes hold a leaf.
// Leaf work: the triangles of all held leaves form one list of (ray, triangle) pairs that the 64 lanes test 64 at a
// time, whichever lane owns the ray -- full lanes whatever the leaf sizes, and neighbouring lanes read neighbouring
// triangle records.  The descent work of a wave is the MAXIMUM over its lanes, its leaf work the SUM / 64; neither
// is the union.
//
// Candidate set, arithmetic and tie rule are the same as in rt_traverse.h (see there): every leaf reached through
// passing slab tests contributes all its triangle references; equal t keeps the smaller global leaf slot.
//
// The per-lane stack holds one entry per tree level (record index of an internal node + the 8-bit set of its
// octants that passed the slab test and are still to be visited), in LDS, laid out [level][lane] so that a lane's
// bank depends only on its lane index (no conflicts whatever levels the lanes are at).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_traverse.h"

namespace rt {

constexpr int kLaneMaxLevels = 17;   // internal nodes at depths 0..16
// The descent phase is left once the lanes holding a leaf outnumber RT_LANE_A_NUM / RT_LANE_A_DEN of the lanes still
// searching (measured best on a 1.3 M-triangle scene: 1 / 2).
#ifndef RT_LANE_A_NUM
#define RT_LANE_A_NUM 1
#define RT_LANE_A_DEN 2
#endif

// A lane may hold a second leaf before the leaf phase starts (it keeps descending meanwhile).  Closest-hit rays gain
// from it (fewer, fuller rounds); any-hit rays do not: their first leaf often already decides them.
#ifndef RT_LANE_TWO_LEAVES_CLOSEST
#define RT_LANE_TWO_LEAVES_CLOSEST 1
#endif
#ifndef RT_LANE_TWO_LEAVES_ANY
#define RT_LANE_TWO_LEAVES_ANY 1
#endif

#ifdef RT_LANE_PROFILE
// [ANY][k]: 0 cycles in descent, 1 cycles in leaf work, 2 descent steps, 3 lanes stepping (sum), 4 leaf chunks, 5 pairs, 6 waves, 7 rounds
__device__ unsigned long long g_lane_prof[2][8];
#define RT_PROF_ADD(k, v) do { if (__lane_id() == 0) atomicAdd(&g_lane_prof[ANY ? 1 : 0][k], (unsigned long long)(v)); } while (0)
#endif

// Cross-lane traffic through LDS goes through volatile pointers in the LDS address space (ds_read / ds_write); a
// volatile generic pointer would compile to flat accesses with a wait after each.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;

struct LaneStackLds {  // of one wave
    uint32_t e[kLaneMaxLevels][64];
    uint32_t hd[64];  // which lane's leaf starts at each position of the current 64-pair chunk
};

// Wave-wide inclusive scans in six DPP steps (no LDS traffic): shifts by 1, 2, 4, 8 inside each row of 16 lanes (lanes
// shifted in from outside the row read 0), then lane 15 of row 0 / 2 is added into row 1 / 3 (row_bcast:15) and lane 31
// into rows 2 and 3 (row_bcast:31).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += dpp_or_zero<0x111, 0xf>(x);  // row_shr:1
    x += dpp_or_zero<0x112, 0xf>(x);  // row_shr:2
    x += dpp_or_zero<0x114, 0xf>(x);  // row_shr:4
    x += dpp_or_zero<0x118, 0xf>(x);  // row_shr:8
    x += dpp_or_zero<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t x) {
    uint32_t y;
    y = dpp_or_zero<0x111, 0xf>(x); x = x < y ? y : x;
    y = dpp_or_zero<0x112, 0xf>(x); x = x < y ? y : x;
    y = dpp_or_zero<0x114, 0xf>(x); x = x < y ? y : x;
    y = dpp_or_zero<0x118, 0xf>(x); x = x < y ? y : x;
    y = dpp_or_zero<0x142, 0xa>(x); x = x < y ? y : x;
    y = dpp_or_zero<0x143, 0xc>(x); x = x < y ? y : x;
    return x;
}

// Slab tests of the 8 octants of node `rec` for this lane's ray: bit o set iff octant o is non-empty and passes
// (collision_detection.h:71-86 per octant; the plane products are shared between octants, see NodeRec).
__device__ __forceinline__ uint32_t lane_node_pass(const NodeRec* __restrict__ nodes, uint32_t rec, f3 o, f3 inv) {
    const float4* p = reinterpret_cast<const float4*>(nodes + rec);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const float lo[3][2] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}};
    const float hi[3][2] = {{q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}};
    const uint32_t nonempty = __float_as_uint(q3.x);
    float tn[3][2], tf[3][2];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float oo = comp(o, ax), ii = comp(inv, ax);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float t0 = (lo[ax][b] - oo) * ii;
            float t1 = (hi[ax][b] - oo) * ii;
            if (ii < 0.0f) { const float s = t0; t0 = t1; t1 = s; }
            tn[ax][b] = t0;
            tf[ax][b] = t1;
        }
    }
    uint32_t pass = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
        float t_min = slab_max(-RT_INF, tn[0][bx]), t_max = slab_min(RT_INF, tf[0][bx]);
        t_min = slab_max(t_min, tn[1][by]);
        t_max = slab_min(t_max, tf[1][by]);
        t_min = slab_max(t_min, tn[2][bz]);
        t_max = slab_min(t_max, tf[2][bz]);
        pass |= (t_max >= t_min) ? (1u << c) : 0u;
    }
    return pass & nonempty;
}

// valid: this lane carries a ray.  For ANY, t_limit is the distance to the light.  `stk` is the wave's LDS stack.
// FAST (any-hit only): unit-length directions in a scene within the bound of any_hit_exec.
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void traverse_lane(const DevOctree& T, LaneStackLds& stk, bool valid, f3 o, f3 d, float t_limit, HitState& h) {
    h.t = RT_INF;
    h.slot = 0xffffffffu;
    h.tri = 0xffffffffu;
    h.found = 0;
    if (T.root_info1 == 0 || ballot64(valid) == 0) return;  // empty tree / no rays
    const uint32_t tid = __lane_id();
    const f3 inv = reciprocal3(d, valid);  // collision_detection.h:76
    bool active = valid;
    bool found = false;
    constexpr bool kTwo = ANY ? (RT_LANE_TWO_LEAVES_ANY != 0) : (RT_LANE_TWO_LEAVES_CLOSEST != 0);
    uint32_t leaf_slot = 0, leaf_left = 0;    // the leaf the lane holds
    uint32_t leaf_slot2 = 0, leaf_left2 = 0;  // a second one (kTwo)
    uint32_t cur_rec = 0, cur_pending = 0;
    int sp = 0;
    if (!(T.root_info1 & kChildInternal)) {  // the root is a leaf: all triangles, no box test (octree.cc:56-63,186-191)
        leaf_slot = T.root_info0;
        leaf_left = active ? T.root_info1 : 0u;
    } else {
        cur_rec = T.root_info0;
        cur_pending = active ? lane_node_pass(T.nodes, cur_rec, o, inv) : 0u;
    }
#ifdef RT_LANE_PROFILE
    unsigned long long pa = 0, pb = 0, psteps = 0, plan = 0, pch = 0, ppairs = 0, prounds = 0;
#endif
    for (;;) {
#ifdef RT_LANE_PROFILE
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- phase A: active lanes without a leaf take descent steps.  The phase ends when they all hold a leaf
        // (or are finished), or earlier once few of them are still searching: those resume in the next round.
        for (;;) {
            const bool can_search = active && (kTwo ? leaf_left2 == 0 : leaf_left == 0);
            const lanemask searching = ballot64(can_search);
            if (searching == 0) break;
            const lanemask holding = ballot64(leaf_left != 0 && !can_search);  // lanes that only wait for the leaf phase
            if (__builtin_popcountll(searching) * RT_LANE_A_NUM < __builtin_popcountll(holding) * RT_LANE_A_DEN) break;
#ifdef RT_LANE_PROFILE
            ++psteps; plan += __builtin_popcountll(searching);
#endif
            if (can_search) {
                if (cur_pending == 0) {
                    if (sp == 0) {
                        active = false;
                    } else {
                        --sp;
                        const uint32_t e = stk.e[sp][tid];
                        cur_rec = e >> 8;
                        cur_pending = e & 0xffu;
                    }
                } else {
                    const uint32_t c = (uint32_t)__builtin_ctz(cur_pending);
                    cur_pending &= cur_pending - 1;
                    const uint2 info = *reinterpret_cast<const uint2*>(&T.nodes[cur_rec].info[c][0]);
                    if (info.y & kChildInternal) {
                        if (cur_pending != 0) {
                            stk.e[sp][tid] = (cur_rec << 8) | cur_pending;
                            ++sp;
                        }
                        cur_rec = info.x;
                        cur_pending = lane_node_pass(T.nodes, cur_rec, o, inv);
                    } else if (kTwo && leaf_left != 0) {
                        leaf_slot2 = info.x;
                        leaf_left2 = info.y;
                    } else {
                        leaf_slot = info.x;
                        leaf_left = info.y;
                    }
                }
            }
        }
#ifdef RT_LANE_PROFILE
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        pa += t1 - t0;
#endif
        if (ballot64(active || leaf_left != 0) == 0) break;  // a finished lane may still hold leaves to test
        // ---- phase B: all (ray, triangle) pairs of the leaves the lanes hold, 64 pairs per step whatever lane owns
        // them.  Pair p of the round belongs to the lane whose leaf covers position p of the concatenated leaves;
        // lane i of the wave tests pair base + i: it fetches the owner's ray through lane shuffles and the triangle
        // record itself -- neighbouring lanes read neighbouring records, so the loads coalesce -- and the owners
        // collect the results from the ballot (any-hit) or from the hit lanes of their range (closest hit; ranges
        // are in slot order, so the tie rule is the lane order).
        {
            const uint32_t lane = __lane_id();
            const uint32_t cnt = leaf_left + (kTwo ? leaf_left2 : 0u);
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            lds_vu32* hd = (lds_vu32*)stk.hd;
            lanemask decided = 0;  // any-hit: owners that found an occluder earlier in this round
            for (uint32_t base = 0; base < total; base += 64) {
                hd[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                if (cnt != 0 && excl >= base && excl < base + 64) hd[excl - base] = lane + 1;
                __builtin_amdgcn_wave_barrier();
                uint32_t own = hd[lane];
                __builtin_amdgcn_wave_barrier();
                const lanemask cont = ballot64(cnt != 0 && excl < base && incl > base);  // a leaf continuing from the last chunk
                const uint32_t carry = cont ? (uint32_t)__builtin_ctzll(cont) + 1u : 0u;
                own = wave_inclusive_max(own);
                own = own < carry ? carry : own;
                const uint32_t p = base + lane;
                const bool pair = p < total;
                const uint32_t L = pair ? own - 1u : lane;
                const uint32_t o_excl = __shfl(excl, (int)L, 64), o_slot = __shfl(leaf_slot, (int)L, 64);
                const f3 ro = mk(__shfl(o.x, (int)L, 64), __shfl(o.y, (int)L, 64), __shfl(o.z, (int)L, 64));
                const f3 rd = mk(__shfl(d.x, (int)L, 64), __shfl(d.y, (int)L, 64), __shfl(d.z, (int)L, 64));
                uint32_t slot = o_slot + (p - o_excl);
                if (kTwo) {  // past the owner's first leaf: its second one
                    const uint32_t first_n = __shfl(leaf_left, (int)L, 64), o_slot2 = __shfl(leaf_slot2, (int)L, 64);
                    if (p - o_excl >= first_n) slot = o_slot2 + (p - o_excl - first_n);
                }
                const bool on = pair && !((decided >> L) & 1ull);
                const float* g = T.slot_geom + (size_t)(on ? slot : 0u) * 9;
                const f3 a = mk(g[0], g[1], g[2]), e1 = mk(g[3], g[4], g[5]), e2 = mk(g[6], g[7], g[8]);
                float t = RT_INF;
                lanemask hit;
                if constexpr (ANY && FAST && kHandScheduledTests) {
                    hit = any_hit_exec_pairs<CULL>(ballot64(on), ro, rd, a, e1, e2, __shfl(t_limit, (int)L, 64));  // renderer.cc:302 included
                } else {
                    hit = ray_triangle<CULL>(ballot64(on), ro, rd, a, e1, e2, t);
                }
                if (hit == 0) continue;
                // this lane's leaf covers positions [lo, hi) of the chunk
                const bool here = cnt != 0 && excl < base + 64 && incl > base;
                const uint32_t lo = excl > base ? excl - base : 0u;
                const uint32_t hi = incl - base < 64u ? incl - base : 64u;
                const lanemask below_hi = hi >= 64u ? ~0ull : ((1ull << hi) - 1ull);
                const lanemask range = here ? (below_hi & ~((1ull << lo) - 1ull)) : 0ull;
                if (ANY) {
                    lanemask good = hit;
                    if constexpr (!(FAST && kHandScheduledTests)) good = hit & ballot64(t < __shfl(t_limit, (int)L, 64));  // renderer.cc:302
                    const bool f = (good & range) != 0;
                    found = found || f;
                    decided |= ballot64(f);
                } else {
                    lanemask m = hit & range;
                    while (ballot64(m != 0) != 0) {
                        const bool has = m != 0;
                        const uint32_t c = has ? (uint32_t)__builtin_ctzll(m) : lane;
                        m &= m - 1ull;
                        const float tc = __shfl(t, (int)c, 64);
                        const uint32_t sc = __shfl(slot, (int)c, 64);
                        // renderer.cc:130: strictly smaller t wins; equal t keeps the smaller global slot
                        const bool b = has && (tc < h.t || (tc == h.t && sc < h.slot));
                        h.t = b ? tc : h.t;
                        h.slot = b ? sc : h.slot;
                    }
                }
            }
            if (ANY && found) active = false;  // this ray is decided
            leaf_left = 0;
            leaf_left2 = 0;
#ifdef RT_LANE_PROFILE
            pb += __builtin_amdgcn_s_memtime() - t1;
            pch += (total + 63) / 64; ppairs += total; ++prounds;
#endif
        }
    }
    if (ANY) h.found = ballot64(found);
    else if (h.slot != 0xffffffffu) h.tri = T.slots[h.slot].tri;  // the winner's triangle index, read once
#ifdef RT_LANE_PROFILE
    RT_PROF_ADD(0, pa); RT_PROF_ADD(1, pb); RT_PROF_ADD(2, psteps); RT_PROF_ADD(3, plan); RT_PROF_ADD(4, pch); RT_PROF_ADD(5, ppairs); RT_PROF_ADD(6, 1); RT_PROF_ADD(7, prounds);
#endif
}

}  // namespace rt
