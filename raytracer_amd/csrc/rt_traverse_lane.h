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
// [ANY][k]: 0 cycles in descent, 1 cycles in leaf work, 2 descent steps, 3 lanes stepping (sum), 4 leaf chunks, 5 pairs, 6 waves, 7 rounds,
// sums over the descent steps of the lanes that 8 pop, 9 enter an internal node, 10 take a leaf, 11 hold leaves and wait, 12 are finished
__device__ unsigned long long g_lane_prof[2][16];
#define RT_PROF_ADD(k, v) do { if (__lane_id() == 0) atomicAdd(&g_lane_prof[ANY ? 1 : 0][k], (unsigned long long)(v)); } while (0)
#endif

// Cross-lane traffic through LDS goes through volatile pointers in the LDS address space (ds_read / ds_write); a
// volatile generic pointer would compile to flat accesses with a wait after each.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;

// The per-lane stack of one wave, in dynamic LDS: one entry per level of internal nodes ([level][lane]) -- the record
// index << 8 | the octants of that node that passed the slab test and are still to be visited -- and, for the pruned
// closest-hit walk, a lower bound of the entry distances of those octants.  hd: which lane's leaf starts at each position
// of the current 64-pair chunk.
struct LaneStack {
    lds_vu32* e;
    lds_vu32* lb;
    lds_vu32* hd;
    unsigned long long* best;  // closest hit: per home lane the smallest (bits of t) << 32 | slot published so far
};
// Words of dynamic LDS one wave needs for a tree with `levels` levels of internal nodes.
__host__ __device__ constexpr uint32_t lane_stack_words(uint32_t levels, bool closest) { return 64u + (closest ? 128u : 0u) + levels * 64u * (closest ? 2u : 1u); }
__device__ __forceinline__ LaneStack lane_stack(uint32_t* wave_base, uint32_t levels, bool closest) {
    LaneStack s;
    s.best = (unsigned long long*)wave_base;  // (closest-hit launches only; so is lb)
    uint32_t* rest = wave_base + (closest ? 128u : 0u);
    s.hd = (lds_vu32*)rest;
    s.e = (lds_vu32*)(rest + 64);
    s.lb = (lds_vu32*)(rest + 64 + levels * 64u);
    return s;
}

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

// Slab tests of the 8 octants of node `rec` for this lane's ray (collision_detection.h:71-86 per octant; the plane
// products are shared between octants, see NodeRec), in the ray's own order: bit k of the result stands for octant
// c = k ^ flip, where flip has the bit of every axis along which the ray runs backwards (1 / dir < 0) -- so that
// ascending k is front to back.  The reference swaps the two products of an axis when 1 / dir < 0; picking, per axis,
// the near and the far plane of the nearer and of the farther child BEFORE the multiplication gives the same products
// from the same operands at the same number of selects.  [lo, hi]: the part of the ray that is walked (Span).
// LB: *lb_rest = a lower bound of the entry distance of every passing octant but the nearest (see below).
// The same tests with the octants in their own numbering (bit o = octant o), for the task-pool form.
__device__ __forceinline__ uint32_t lane_node_pass_natural(const NodeRec* __restrict__ nodes, uint32_t rec, f3 o, f3 inv, float lo, float hi) {
    const float4* p = reinterpret_cast<const float4*>(nodes + rec);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const float plo[3][2] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}};
    const float phi[3][2] = {{q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}};
    const uint32_t nonempty = __float_as_uint(q3.x);
    float tn[3][2], tf[3][2];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float oo = comp(o, ax), ii = comp(inv, ax);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float t0 = (plo[ax][b] - oo) * ii;
            float t1 = (phi[ax][b] - oo) * ii;
            if (ii < 0.0f) { const float s = t0; t0 = t1; t1 = s; }
            tn[ax][b] = t0;
            tf[ax][b] = t1;
        }
    }
    uint32_t pass = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
        float t_min = slab_max(lo, tn[0][bx]), t_max = slab_min(hi, tf[0][bx]);
        t_min = slab_max(t_min, tn[1][by]);
        t_max = slab_min(t_max, tf[1][by]);
        t_min = slab_max(t_min, tn[2][bz]);
        t_max = slab_min(t_max, tf[2][bz]);
        pass |= (t_max >= t_min) ? (1u << c) : 0u;
    }
    return pass & nonempty;
}

__device__ __forceinline__ uint32_t ray_flip(f3 inv) { return (inv.x < 0.0f ? 4u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 1u : 0u); }
template <bool LB>
__device__ __forceinline__ uint32_t lane_node_pass(const NodeRec* __restrict__ nodes, uint32_t rec, f3 o, f3 inv, uint32_t flip, float lo, float hi,
                                                   float* lb_rest) {
    const float4* p = reinterpret_cast<const float4*>(nodes + rec);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const float plo[3][2] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}};
    const float phi[3][2] = {{q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}};
    // the non-empty octants in this ray's order: NodeRec::ordered[flip]
    const uint32_t table = (flip & 4u) ? __float_as_uint(q3.w) : __float_as_uint(q3.z);
    const uint32_t nonempty = (table >> ((flip & 3u) * 8u)) & 0xffu;
    float tn[3][2], tf[3][2];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float oo = comp(o, ax), ii = comp(inv, ax);
        const bool neg = (flip & (4u >> ax)) != 0u;  // = (ii < 0), see ray_flip
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float near_plane = neg ? phi[ax][k ^ 1] : plo[ax][k];
            const float far_plane = neg ? plo[ax][k ^ 1] : phi[ax][k];
            tn[ax][k] = (near_plane - oo) * ii;
            tf[ax][k] = (far_plane - oo) * ii;
        }
    }
    uint32_t pass = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int kx = (k >> 2) & 1, ky = (k >> 1) & 1, kz = k & 1;
        float t_min = slab_max(lo, tn[0][kx]), t_max = slab_min(hi, tf[0][kx]);
        t_min = slab_max(t_min, tn[1][ky]);
        t_max = slab_min(t_max, tf[1][ky]);
        t_min = slab_max(t_min, tn[2][kz]);
        t_max = slab_min(t_max, tf[2][kz]);
        pass |= (t_max >= t_min) ? (1u << k) : 0u;
    }
    pass &= nonempty;
    if (LB) {
        // Per axis the near product of the farther child is never below the nearer child's (same multiplier, ordered
        // planes; a NaN product counts as -inf on both sides of the comparison), so an octant's entry distance
        // t_min(k) = max(lo, tn[0][kx], tn[1][ky], tn[2][kz]) does not decrease when a bit of k is set.  The octants that
        // remain after the nearest passing one therefore all enter at or beyond t_min(AND of their indices): one more
        // evaluation instead of a minimum over eight.  An octant with t_min > hi' fails the slab test under any upper
        // bound hi' -- that is what lets a stack entry be dropped unseen once the best hit is closer than its bound.
        const uint32_t rest = pass & (pass - 1u);
        const bool ux = (rest & 0x0fu) == 0u, uy = (rest & 0x33u) == 0u, uz = (rest & 0x55u) == 0u;
        float b = slab_max(lo, ux ? tn[0][1] : tn[0][0]);
        b = slab_max(b, uy ? tn[1][1] : tn[1][0]);
        b = slab_max(b, uz ? tn[2][1] : tn[2][0]);
        *lb_rest = b;
    }
    return pass;
}

// In-wave work stealing (RT_LANE_STEAL).  The rays of a wave need very different numbers of descent
// steps -- a shadow ray that skims a finely tessellated surface passes hundreds of small octants, its neighbours a
// dozen -- and a wave lasts as long as its slowest lane: on a 1.3 M-triangle scene 38 of 64 lanes sat finished through
// the average descent step.  The octants a ray still has to visit are independent subtrees (an any-hit answer is an OR
// over them), so a finished lane takes over the OLDEST stack entry of a busy lane -- the subtree nearest the root --
// together with a copy of that lane's ray, and walks it with its own stack.  Answers are kept per home lane of a ray
// (`owner`): an occluder found by any lane decides the ray for all lanes that work on it; closest hits are merged with a
// 64-bit minimum per home lane in LDS ((bits of t) << 32 | leaf slot: the order of the keys is the tie rule).
#ifndef RT_LANE_STEAL
#define RT_LANE_STEAL 1
#endif
#ifndef RT_LANE_STEAL_MIN_IDLE
#define RT_LANE_STEAL_MIN_IDLE 8
#endif
// Closest-hit rays: off.  Their ordered, pruned walks are short and alike (47 of 64 lanes take part in the average
// descent step without it), and the merge of the lanes' results costs 12 vector registers -- 5 instead of 7 waves per
// SIMD: 2.24 against 2.04 ms for the closest-hit passes of a 4K frame of a 1.3 M-triangle scene.
#ifndef RT_LANE_STEAL_CLOSEST
#define RT_LANE_STEAL_CLOSEST 0
#endif

__device__ __forceinline__ uint32_t lanes_below(lanemask m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Any-hit walk of the rays with `valid` set; returns the lanes whose ray has an occluder.  t_limit: distance to the light.
// BASES: the lanes walk different copies of the tree that lie in one pair of arrays (per-light trees, one light per
// lane): rec_base / slot_base are added to every record index / leaf slot index a lane reads from its copy, and travel
// with the ray when another lane takes over part of its walk.
template <bool CULL, bool FAST, bool BASES = false>
__device__ __forceinline__ lanemask traverse_lane_any(const DevOctree& T, const LaneStack& stk, bool valid, f3 o, f3 d, float t_limit, Span sp,
                                                      uint32_t rec_base = 0, uint32_t slot_base = 0) {
    [[maybe_unused]] constexpr bool ANY = true;  // (the profiling macros index by it)
    const uint32_t tid = __lane_id();
    if constexpr (!BASES) rec_base = slot_base = 0;
    f3 inv = reciprocal3(d, valid);  // collision_detection.h:76
    uint32_t flip = ray_flip(inv);
    uint32_t owner = tid;    // the home lane of the ray this lane works on
    lanemask decided = 0;    // home lanes whose ray has an occluder
    bool active = valid;
    constexpr bool kTwo = RT_LANE_TWO_LEAVES_ANY != 0;
    uint32_t leaf_slot = 0, leaf_left = 0;    // the leaf the lane holds
    uint32_t leaf_slot2 = 0, leaf_left2 = 0;  // a second one (kTwo)
    uint32_t cur_rec = 0, cur_pending = 0;
    int sp_n = 0, sb = 0;  // the lane's stack entries are [sb, sp_n)
    if (!(T.root_info1 & kChildInternal)) {  // the root is a leaf: all triangles, no box test (octree.cc:56-63,186-191)
        leaf_slot = T.root_info0 + slot_base;
        leaf_left = active ? T.root_info1 : 0u;
    } else {
        cur_rec = T.root_info0 + rec_base;
        cur_pending = active ? lane_node_pass<false>(T.nodes, cur_rec, o, inv, flip, sp.lo, sp.hi, nullptr) : 0u;
    }
#ifdef RT_LANE_PROFILE
    unsigned long long pa = 0, pb = 0, psteps = 0, plan = 0, pch = 0, ppairs = 0, prounds = 0, ppop = 0, pint = 0, pleaf = 0, phold = 0, pdone = 0;
#endif
    for (;;) {
#ifdef RT_LANE_PROFILE
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- phase A: active lanes without a leaf take descent steps (see traverse_lane_walk)
        for (;;) {
            const bool can_search = active && (kTwo ? leaf_left2 == 0 : leaf_left == 0);
            const lanemask searching = ballot64(can_search);
            if (searching == 0) break;
            const lanemask holding = ballot64(leaf_left != 0 && !can_search);  // lanes that only wait for the leaf phase
            if (__builtin_popcountll(searching) * RT_LANE_A_NUM < __builtin_popcountll(holding) * RT_LANE_A_DEN) break;
#if RT_LANE_STEAL
            {
                const bool idle = !active && leaf_left == 0;
                const lanemask idle_m = ballot64(idle);
                if (__builtin_popcountll(idle_m) >= RT_LANE_STEAL_MIN_IDLE) {
                    const bool donor = active && sp_n > sb;
                    const lanemask donor_m = ballot64(donor);
                    if (donor_m != 0) {
                        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_m), n_donor = (uint32_t)__builtin_popcountll(donor_m);
                        const uint32_t rank_d = lanes_below(donor_m), rank_i = lanes_below(idle_m);
                        const bool gives = donor && rank_d < n_idle, thief = idle && rank_i < n_donor;
                        if (gives) stk.hd[rank_d] = tid;  // the k-th donor serves the k-th idle lane
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t from = thief ? stk.hd[rank_i] : tid;
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t e = stk.e[(uint32_t)__shfl(sb, (int)from, 64) * 64u + from];  // (a lane's own column if it steals nothing)
                        const f3 t_o = mk(__shfl(o.x, (int)from, 64), __shfl(o.y, (int)from, 64), __shfl(o.z, (int)from, 64));
                        const f3 t_d = mk(__shfl(d.x, (int)from, 64), __shfl(d.y, (int)from, 64), __shfl(d.z, (int)from, 64));
                        const f3 t_inv = mk(__shfl(inv.x, (int)from, 64), __shfl(inv.y, (int)from, 64), __shfl(inv.z, (int)from, 64));
                        const uint32_t t_flip = __shfl(flip, (int)from, 64), t_owner = __shfl(owner, (int)from, 64);
                        const float t_tl = __shfl(t_limit, (int)from, 64), t_lo = __shfl(sp.lo, (int)from, 64), t_hi = __shfl(sp.hi, (int)from, 64);
                        uint32_t t_rb = 0, t_sb = 0;
                        if constexpr (BASES) {
                            t_rb = __shfl(rec_base, (int)from, 64);
                            t_sb = __shfl(slot_base, (int)from, 64);
                        }
                        if (thief) {
                            o = t_o; d = t_d; inv = t_inv; flip = t_flip; owner = t_owner;
                            t_limit = t_tl; sp.lo = t_lo; sp.hi = t_hi;
                            if constexpr (BASES) { rec_base = t_rb; slot_base = t_sb; }
                            cur_rec = e >> 8;
                            cur_pending = e & 0xffu;
                            active = true;
                            sp_n = sb = 0;
                        }
                        __builtin_amdgcn_wave_barrier();  // the entries have been read before their lanes may reuse the places
                        if (gives) {
                            ++sb;
                            if (sb == sp_n) sb = sp_n = 0;
                        }
                        continue;
                    }
                }
            }
#endif
#ifdef RT_LANE_PROFILE
            ++psteps; plan += __builtin_popcountll(searching);
            ppop += __builtin_popcountll(ballot64(can_search && cur_pending == 0));
            phold += __builtin_popcountll(holding);
            pdone += __builtin_popcountll(ballot64(!active && leaf_left == 0));
#endif
            if (can_search) {
                // a lane that has used up its node first returns to the last node with octants left, in the same step
                if (cur_pending == 0 && sp_n != sb) {
                    --sp_n;
                    const uint32_t e = stk.e[sp_n * 64 + tid];
                    cur_rec = e >> 8;
                    cur_pending = e & 0xffu;
                    if (sp_n == sb) sb = sp_n = 0;
                }
                if (cur_pending == 0) {
                    active = false;  // (the stack is empty: the pop above would have refilled the node)
                } else {
                    const uint32_t k = (uint32_t)__builtin_ctz(cur_pending);  // the nearest pending octant
                    cur_pending &= cur_pending - 1;
                    const uint2 info = *reinterpret_cast<const uint2*>(&T.nodes[cur_rec].info[k ^ flip][0]);
#ifdef RT_LANE_PROFILE
                    pint += __builtin_popcountll(ballot64((info.y & kChildInternal) != 0));
                    pleaf += __builtin_popcountll(ballot64((info.y & kChildInternal) == 0));
#endif
                    if (info.y & kChildInternal) {
                        if (cur_pending != 0) {
                            stk.e[sp_n * 64 + tid] = (cur_rec << 8) | cur_pending;
                            ++sp_n;
                        }
                        cur_rec = info.x + rec_base;
                        cur_pending = lane_node_pass<false>(T.nodes, cur_rec, o, inv, flip, sp.lo, sp.hi, nullptr);
                    } else if (kTwo && leaf_left != 0) {
                        leaf_slot2 = info.x + slot_base;
                        leaf_left2 = info.y;
                    } else {
                        leaf_slot = info.x + slot_base;
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
        // ---- phase B: the (ray, triangle) pairs of the held leaves, 64 per step (see traverse_lane_walk)
        {
            const uint32_t lane = tid;
            const uint32_t cnt = leaf_left + (kTwo ? leaf_left2 : 0u);
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            lds_vu32* hd = stk.hd;
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
                const uint32_t L = pair ? own - 1u : lane;  // the lane that holds the pair's leaf
                const uint32_t o_excl = __shfl(excl, (int)L, 64), o_slot = __shfl(leaf_slot, (int)L, 64);
                const f3 ro = mk(__shfl(o.x, (int)L, 64), __shfl(o.y, (int)L, 64), __shfl(o.z, (int)L, 64));
                const f3 rd = mk(__shfl(d.x, (int)L, 64), __shfl(d.y, (int)L, 64), __shfl(d.z, (int)L, 64));
                const uint32_t r_owner = __shfl(owner, (int)L, 64);
                uint32_t slot = o_slot + (p - o_excl);
                if (kTwo) {  // past the holder's first leaf: its second one
                    const uint32_t first_n = __shfl(leaf_left, (int)L, 64), o_slot2 = __shfl(leaf_slot2, (int)L, 64);
                    if (p - o_excl >= first_n) slot = o_slot2 + (p - o_excl - first_n);
                }
                const bool on = pair && !((decided >> r_owner) & 1ull);
                const float* g = T.slot_geom + (size_t)(on ? slot : 0u) * 9;
                const f3 a = mk(g[0], g[1], g[2]), e1 = mk(g[3], g[4], g[5]), e2 = mk(g[6], g[7], g[8]);
                float t = RT_INF;
                lanemask good;
                if constexpr (FAST && kHandScheduledTests) {
                    good = any_hit_exec_pairs<CULL>(ballot64(on), ro, rd, a, e1, e2, __shfl(t_limit, (int)L, 64));  // renderer.cc:302 included
                } else {
                    good = ray_triangle<CULL>(ballot64(on), ro, rd, a, e1, e2, t);
                    if (good != 0) good &= ballot64(t < __shfl(t_limit, (int)L, 64));  // renderer.cc:302
                }
                // the occluders' rays are decided, whichever lanes work on them
                while (good != 0) {
                    const int gl = __builtin_ctzll(good);
                    good &= good - 1ull;
                    decided |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)r_owner, gl);
                }
            }
            leaf_left = 0;
            leaf_left2 = 0;
            if ((decided >> owner) & 1ull) {  // this lane's ray is decided: drop what is left of it
                active = false;
                cur_pending = 0;
                sp_n = sb = 0;
            }
#ifdef RT_LANE_PROFILE
            pb += __builtin_amdgcn_s_memtime() - t1;
            pch += (total + 63) / 64; ppairs += total; ++prounds;
#endif
        }
    }
#ifdef RT_LANE_PROFILE
    RT_PROF_ADD(0, pa); RT_PROF_ADD(1, pb); RT_PROF_ADD(2, psteps); RT_PROF_ADD(3, plan); RT_PROF_ADD(4, pch); RT_PROF_ADD(5, ppairs); RT_PROF_ADD(6, 1); RT_PROF_ADD(7, prounds);
    RT_PROF_ADD(8, ppop); RT_PROF_ADD(9, pint); RT_PROF_ADD(10, pleaf); RT_PROF_ADD(11, phold); RT_PROF_ADD(12, pdone);
#endif
    return decided;
}

// Closest-hit walk of the rays with `valid` set.  lo: lower end of the part of the ray that is walked; prune: the upper
// end follows the ray's best distance (the smaller of this lane's own best and the best any lane has published for the
// ray).  Returns in `ties` the home lanes whose best distance was met by two distinct triangles.
template <bool CULL>
__device__ __forceinline__ void traverse_lane_closest(const DevOctree& T, const LaneStack& stk, bool valid, f3 o, f3 d, float lo, bool prune,
                                                      HitState& h, lanemask& ties_out) {
    [[maybe_unused]] constexpr bool ANY = false;  // (the profiling macros index by it)
    constexpr bool kSteal = (RT_LANE_STEAL != 0) && (RT_LANE_STEAL_CLOSEST != 0);
    const uint32_t tid = __lane_id();
    f3 inv = reciprocal3(d, valid);  // collision_detection.h:76
    uint32_t flip = ray_flip(inv);
    float pad_abs = span_pad_abs(o);
    uint32_t owner = tid;  // the home lane of the ray this lane works on
    lanemask ties = 0;     // home lanes with two distinct triangles at the best distance
    bool active = valid;
    float bt = RT_INF;     // this lane's best candidate for that ray: distance, leaf slot, triangle
    uint32_t bslot = 0xffffffffu, btri = 0xffffffffu;
    float bound = RT_INF;  // the best distance published for the ray by any lane
    bool improved = false, tie_l = false;
    constexpr bool kTwo = RT_LANE_TWO_LEAVES_CLOSEST != 0;
    uint32_t leaf_slot = 0, leaf_left = 0;    // the leaf the lane holds
    uint32_t leaf_slot2 = 0, leaf_left2 = 0;  // a second one (kTwo)
    uint32_t cur_rec = 0, cur_pending = 0;
    float cur_lb = -RT_INF;  // no pending octant of cur_rec is entered before this distance
    int sp_n = 0, sb = 0;    // the lane's stack entries are [sb, sp_n)
    if (kSteal) stk.best[tid] = ~0ull;
    if (!(T.root_info1 & kChildInternal)) {  // the root is a leaf: all triangles, no box test (octree.cc:56-63,186-191)
        leaf_slot = T.root_info0;
        leaf_left = active ? T.root_info1 : 0u;
    } else {
        cur_rec = T.root_info0;
        cur_pending = active ? lane_node_pass<true>(T.nodes, cur_rec, o, inv, flip, lo, RT_INF, &cur_lb) : 0u;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef RT_LANE_PROFILE
    unsigned long long pa = 0, pb = 0, psteps = 0, plan = 0, pch = 0, ppairs = 0, prounds = 0, ppop = 0, pint = 0, pleaf = 0, phold = 0, pdone = 0;
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
            if constexpr (kSteal) {
                const bool idle = !active && leaf_left == 0;
                const lanemask idle_m = ballot64(idle);
                if (__builtin_popcountll(idle_m) >= RT_LANE_STEAL_MIN_IDLE) {
                    const bool donor = active && sp_n > sb;
                    const lanemask donor_m = ballot64(donor);
                    if (donor_m != 0) {
                        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_m), n_donor = (uint32_t)__builtin_popcountll(donor_m);
                        const uint32_t rank_d = lanes_below(donor_m), rank_i = lanes_below(idle_m);
                        const bool gives = donor && rank_d < n_idle, thief = idle && rank_i < n_donor;
                        if (gives) stk.hd[rank_d] = tid;  // the k-th donor serves the k-th idle lane
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t from = thief ? stk.hd[rank_i] : tid;
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t at = (uint32_t)__shfl(sb, (int)from, 64) * 64u + from;  // (a lane's own column if it steals nothing)
                        const uint32_t e = stk.e[at], e_lb = stk.lb[at];
                        const f3 t_o = mk(__shfl(o.x, (int)from, 64), __shfl(o.y, (int)from, 64), __shfl(o.z, (int)from, 64));
                        const f3 t_d = mk(__shfl(d.x, (int)from, 64), __shfl(d.y, (int)from, 64), __shfl(d.z, (int)from, 64));
                        const f3 t_inv = mk(__shfl(inv.x, (int)from, 64), __shfl(inv.y, (int)from, 64), __shfl(inv.z, (int)from, 64));
                        const uint32_t t_flip = __shfl(flip, (int)from, 64), t_owner = __shfl(owner, (int)from, 64);
                        const float t_lo = __shfl(lo, (int)from, 64), t_pad = __shfl(pad_abs, (int)from, 64);
                        if (thief) {  // (an idle lane has published all it found for its last ray)
                            o = t_o; d = t_d; inv = t_inv; flip = t_flip; owner = t_owner;
                            lo = t_lo; pad_abs = t_pad;
                            cur_rec = e >> 8;
                            cur_pending = e & 0xffu;
                            cur_lb = __uint_as_float(e_lb);
                            active = true;
                            sp_n = sb = 0;
                            bt = RT_INF;
                            bslot = btri = 0xffffffffu;
                            const uint32_t tb = (uint32_t)(__hip_atomic_load(&stk.best[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 32);
                            bound = tb == 0xffffffffu ? RT_INF : __uint_as_float(tb);
                        }
                        __builtin_amdgcn_wave_barrier();  // the entries have been read before their lanes may reuse the places
                        if (gives) {
                            ++sb;
                            if (sb == sp_n) sb = sp_n = 0;
                        }
                        continue;
                    }
                }
            }
#ifdef RT_LANE_PROFILE
            ++psteps; plan += __builtin_popcountll(searching);
            ppop += __builtin_popcountll(ballot64(can_search && cur_pending == 0));
            phold += __builtin_popcountll(holding);
            pdone += __builtin_popcountll(ballot64(!active && leaf_left == 0));
#endif
            if (can_search) {
                const float hi = prune ? span_hi_closest(bt < bound ? bt : bound, pad_abs) : RT_INF;
                // a lane that has used up its node first returns to the last node with octants left, in the same step
                if (cur_pending == 0 && sp_n != sb) {
                    --sp_n;
                    const uint32_t e = stk.e[sp_n * 64 + tid];
                    cur_rec = e >> 8;
                    cur_pending = e & 0xffu;
                    cur_lb = __uint_as_float(stk.lb[sp_n * 64 + tid]);
                    if (cur_lb > hi) cur_pending = 0;  // every octant of the entry lies beyond the best hit
                    if (sp_n == sb) sb = sp_n = 0;
                }
                if (cur_pending == 0) {
                    if (sp_n == sb) active = false;
                } else {
                    const uint32_t k = (uint32_t)__builtin_ctz(cur_pending);  // the nearest pending octant
                    cur_pending &= cur_pending - 1;
                    const uint2 info = *reinterpret_cast<const uint2*>(&T.nodes[cur_rec].info[k ^ flip][0]);
#ifdef RT_LANE_PROFILE
                    pint += __builtin_popcountll(ballot64((info.y & kChildInternal) != 0));
                    pleaf += __builtin_popcountll(ballot64((info.y & kChildInternal) == 0));
#endif
                    if (info.y & kChildInternal) {
                        if (cur_pending != 0) {
                            stk.e[sp_n * 64 + tid] = (cur_rec << 8) | cur_pending;
                            stk.lb[sp_n * 64 + tid] = __float_as_uint(cur_lb);
                            ++sp_n;
                        }
                        cur_rec = info.x;
                        cur_pending = lane_node_pass<true>(T.nodes, cur_rec, o, inv, flip, lo, hi, &cur_lb);
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
        // ---- phase B: all (ray, triangle) pairs of the leaves the lanes hold, 64 pairs per step whatever lane holds
        // them.  Pair p of the round belongs to the lane whose leaf covers position p of the concatenated leaves;
        // lane i of the wave tests pair base + i: it fetches the holder's ray through lane shuffles and the triangle
        // record itself -- neighbouring lanes read neighbouring records, so the loads coalesce -- and the holders
        // collect the results from the hit lanes of their range (ranges are in slot order, so the tie rule is the lane
        // order).
        {
            const uint32_t lane = tid;
            const uint32_t cnt = leaf_left + (kTwo ? leaf_left2 : 0u);
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            lds_vu32* hd = stk.hd;
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
                if (kTwo) {  // past the holder's first leaf: its second one
                    const uint32_t first_n = __shfl(leaf_left, (int)L, 64), o_slot2 = __shfl(leaf_slot2, (int)L, 64);
                    if (p - o_excl >= first_n) slot = o_slot2 + (p - o_excl - first_n);
                }
                // the whole record: the triangle index tells a second listing of the best triangle from a tie
                const LeafTri* g = T.slots + (pair ? slot : 0u);
                const float4 r0 = reinterpret_cast<const float4*>(g)[0], r1 = reinterpret_cast<const float4*>(g)[1];
                const float2 r2 = reinterpret_cast<const float2*>(g)[4];
                const f3 a = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
                const uint32_t tri = __float_as_uint(r2.y);
                float t = RT_INF;
                const lanemask hit = ray_triangle<CULL>(ballot64(pair), ro, rd, a, e1, e2, t);
                if (hit == 0) continue;
                // this lane's leaves cover positions [rlo, rhi) of the chunk
                const bool here = cnt != 0 && excl < base + 64 && incl > base;
                const uint32_t rlo = excl > base ? excl - base : 0u;
                const uint32_t rhi = incl - base < 64u ? incl - base : 64u;
                const lanemask below_hi = rhi >= 64u ? ~0ull : ((1ull << rhi) - 1ull);
                const lanemask range = here ? (below_hi & ~((1ull << rlo) - 1ull)) : 0ull;
                lanemask m = hit & range;
                while (ballot64(m != 0) != 0) {
                    const bool has = m != 0;
                    const uint32_t c = has ? (uint32_t)__builtin_ctzll(m) : lane;
                    m &= m - 1ull;
                    const float tc = __shfl(t, (int)c, 64);
                    const uint32_t sc = __shfl(slot, (int)c, 64);
                    const uint32_t trc = __shfl(tri, (int)c, 64);
                    // renderer.cc:130: strictly smaller t wins; equal t keeps the smaller global slot
                    const bool less = has && tc < bt, same = has && tc == bt;
                    tie_l = (tie_l && !less) || (same && trc != btri);  // two distinct triangles at the best distance
                    const bool b = less || (same && sc < bslot);
                    bt = b ? tc : bt;
                    bslot = b ? sc : bslot;
                    btri = b ? trc : btri;
                    improved = improved || b;
                }
            }
            leaf_left = 0;
            leaf_left2 = 0;
            // publish: what this lane found joins the ray's best, and every lane learns the ray's best distance
            if constexpr (kSteal) {
            if (improved) {
                const unsigned long long old = atomicMin(&stk.best[owner], ((unsigned long long)__float_as_uint(bt) << 32) | (unsigned long long)bslot);
                // the same distance from another slot: a second listing of the same triangle, or a tie of two triangles
                if ((uint32_t)(old >> 32) == __float_as_uint(bt) && (uint32_t)old != bslot && T.slots[(uint32_t)old].tri != btri) tie_l = true;
                improved = false;
            }
            lanemask tm = ballot64(tie_l);
            tie_l = false;
            while (tm != 0) {
                const int gl = __builtin_ctzll(tm);
                tm &= tm - 1ull;
                ties |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)owner, gl);
            }
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t tb = (uint32_t)(__hip_atomic_load(&stk.best[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 32);
                bound = tb == 0xffffffffu ? RT_INF : __uint_as_float(tb);
            }
            }
#ifdef RT_LANE_PROFILE
            pb += __builtin_amdgcn_s_memtime() - t1;
            pch += (total + 63) / 64; ppairs += total; ++prounds;
#endif
        }
    }
    if constexpr (kSteal) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long key = __hip_atomic_load(&stk.best[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (valid && key != ~0ull) {
            h.t = __uint_as_float((uint32_t)(key >> 32));
            h.slot = (uint32_t)key;
            h.tri = T.slots[h.slot].tri;
        }
        __builtin_amdgcn_wave_barrier();  // (a second walk of this wave resets best[])
        ties_out = ties;
    } else {  // every lane has walked its own ray
        h.t = bt;
        h.slot = bslot;
        h.tri = btri;
        ties_out = ballot64(tie_l);
    }
#ifdef RT_LANE_PROFILE
    RT_PROF_ADD(0, pa); RT_PROF_ADD(1, pb); RT_PROF_ADD(2, psteps); RT_PROF_ADD(3, plan); RT_PROF_ADD(4, pch); RT_PROF_ADD(5, ppairs); RT_PROF_ADD(6, 1); RT_PROF_ADD(7, prounds);
    RT_PROF_ADD(8, ppop); RT_PROF_ADD(9, pint); RT_PROF_ADD(10, pleaf); RT_PROF_ADD(11, phold); RT_PROF_ADD(12, pdone);
#endif
}

// valid: this lane carries a ray.  clip: see traverse<> in rt_traverse.h -- any-hit rays walk the segment up to the
// light; closest-hit rays walk front to back up to their best hit, and a ray whose best distance two distinct
// triangles share is walked once more like the reference walks it (whole line, no pruning), which settles the tie by
// the reference's own rule.
template <bool ANY, bool CULL, bool FAST = false, bool BASES = false>
__device__ __forceinline__ void traverse_lane(const DevOctree& T, const LaneStack& stk, bool valid, f3 o, f3 d, float t_limit, bool clip,
                                              HitState& h, uint32_t rec_base = 0, uint32_t slot_base = 0) {
    hit_state_reset(h);
    if (T.root_info1 == 0 || ballot64(valid) == 0) return;  // empty tree / no rays
    if constexpr (ANY) {
        h.found = traverse_lane_any<CULL, FAST, BASES>(T, stk, valid, o, d, t_limit, span_segment(o, t_limit, clip), rec_base, slot_base);
    } else {
        bool todo = valid, pruned = clip;
#pragma nounroll
        for (;;) {
            HitState w;
            hit_state_reset(w);
            lanemask ties = 0;
            traverse_lane_closest<CULL>(T, stk, todo, o, d, span_lo_closest(o, pruned), pruned, w, ties);
            h.t = todo ? w.t : h.t;
            h.slot = todo ? w.slot : h.slot;
            h.tri = todo ? w.tri : h.tri;
            if (!pruned || ties == 0) break;
            todo = lane_of(ties);
            pruned = false;
        }
    }
}

}  // namespace rt
