// rt_traverse_pool.h -- octree traversal with a wave-wide task pool, for gfx950 (device only).
//
// The per-lane form (rt_traverse_lane.h) ties a ray's descent to its lane: a descent step runs with the lanes that
// happen to need one (some 40 % on a large tree), the wave needs as many steps as its slowest ray, and every step is
// one more dependent memory round trip.  Which candidates a ray collects does not depend on the order of its node
// visits, though: the result is a minimum (closest hit) or an OR (any hit) over all leaves behind passing slab tests.
// So here the descent of all rays of a wave is one pool of (ray, node) tasks in LDS.  A node step hands the top 64
// tasks to the 64 lanes, whichever rays they belong to: each lane slab-tests the 8 octants of its node for the task's
// ray (fetched from the owner lane with shuffles) and appends the passing internal octants as new tasks and the
// passing leaves to a list of (ray, leaf) items.  Whenever that list holds 64 items, or the pool is empty, the
// triangles of the listed leaves are tested as (ray, triangle) pairs, 64 pairs per step with coalesced record loads
// (as in the per-lane form).  All lanes work in nearly every step, and the number of dependent round trips of a wave
// is about (node visits of all its rays) / 64 instead of the maximum over its rays.
//
// Results.  Closest hit: hit pairs take a 64-bit minimum on best[ray] in LDS with the key (bits of t) << 32 | slot
// -- t is positive and finite there, so the order of the keys is "smaller t first, equal t: smaller global leaf
// slot", the same rule as in rt_traverse.h.  Any hit: a bit per decided ray; its remaining tasks are dropped when
// they come up.
//
// The pool is a stack (last in, first out), so it grows like a depth-first walk: a step takes fewer tasks when the
// pool is nearly full and a single one beyond the soft limit, after which the growth is bounded by 7 per tree level.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_traverse.h"
#include "rt_traverse_lane.h"

namespace rt {

#ifndef RT_POOL_ANY_TRIGGER
#define RT_POOL_ANY_TRIGGER 64
#endif
#ifndef RT_POOL_NODE_CAP
#define RT_POOL_NODE_CAP 512
#define RT_POOL_LEAF_CAP 448
#endif
constexpr uint32_t kPoolNodeCap = RT_POOL_NODE_CAP;  // node tasks: record index (24 bits) | ray << 24
constexpr uint32_t kPoolNodeSoft = kPoolNodeCap - 8u * (uint32_t)kLaneMaxLevels;
constexpr uint32_t kPoolLeafCap = RT_POOL_LEAF_CAP;   // leaf items: first slot, triangle count (26 bits) | ray << 26
constexpr uint32_t kPoolRecBits = 24, kPoolCountBits = 26;
static_assert(kPoolLeafCap >= 64 + 8 * 8, "a node step must fit behind 63 waiting leaf items");
static_assert(kPoolNodeSoft >= 64, "the 64 root tasks of a wave must fit under the soft limit");

#ifdef RT_LANE_PROFILE
// [ANY][k]: 0 cycles in node steps, 1 cycles in leaf work, 2 node steps, 3 tasks taken, 4 leaf chunks, 5 pairs, 6 waves, 7 leaf phases
__device__ unsigned long long g_pool_prof[2][8];
#define RT_PPROF_ADD(k, v) do { if (__lane_id() == 0) atomicAdd(&g_pool_prof[ANY ? 1 : 0][k], (unsigned long long)(v)); } while (0)
#endif

struct PoolLds {  // of one wave
    uint32_t nodes[kPoolNodeCap];
    uint32_t leaf_slot[kPoolLeafCap];
    uint32_t leaf_count[kPoolLeafCap];
    unsigned long long best[64];
    uint32_t hd[64];
};

// One walk of the rays with `valid` set.  For ANY, t_limit is the distance to the light.
// sp: the part of the ray whose boxes are walked; prune (closest hit): its upper end follows the ray's best distance
// (read from best[] when a task comes up) and `tie` returns the rays whose best distance two distinct triangles share.
template <bool ANY, bool CULL, bool FAST>
__device__ __forceinline__ void traverse_pool_walk(const DevOctree& T, PoolLds& lds, bool valid, f3 o, f3 d, float t_limit, Span sp, bool prune,
                                                   HitState& h, lanemask& tie) {
    const lanemask vm = ballot64(valid);
    tie = 0;
    if (vm == 0) return;
    const uint32_t lane = __lane_id();
    const f3 inv = reciprocal3(d, valid);  // collision_detection.h:76
    const float pad_abs = span_pad_abs(o);
    lds_vu32* nstk = (lds_vu32*)lds.nodes;
    lds_vu32* lslot = (lds_vu32*)lds.leaf_slot;
    lds_vu32* lcount = (lds_vu32*)lds.leaf_count;
    lds_vu32* hd = (lds_vu32*)lds.hd;
    uint32_t ns = (uint32_t)__builtin_popcountll(vm), nl = 0;  // wave-uniform fill levels
    if (valid) nstk[__builtin_popcountll(vm & ((1ull << lane) - 1ull))] = T.root_info0 | (lane << kPoolRecBits);
    if (!ANY) lds.best[lane] = ~0ull;
    lanemask decided = 0;  // any-hit: rays that found an occluder
    __builtin_amdgcn_wave_barrier();
#ifdef RT_LANE_PROFILE
    unsigned long long pa = 0, pb = 0, psteps = 0, ptasks = 0, pch = 0, ppairs = 0, pphases = 0;
#endif
    for (;;) {
#ifdef RT_LANE_PROFILE
        const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
        if (nl >= (ANY ? (uint32_t)RT_POOL_ANY_TRIGGER : 64u) || (ns == 0 && nl != 0)) {
            // ---- leaf work: the last up to 64 items of the list, lane i holds one
            const uint32_t m = nl < 64u ? nl : 64u;
            uint32_t my_slot = 0, cnt = 0, my_ray = 0;
            if (lane < m) {
                my_slot = lslot[nl - m + lane];
                const uint32_t w = lcount[nl - m + lane];
                cnt = w & ((1u << kPoolCountBits) - 1u);
                my_ray = w >> kPoolCountBits;
                if (ANY && ((decided >> my_ray) & 1ull)) cnt = 0;
            }
            nl -= m;
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            for (uint32_t base = 0; base < total; base += 64) {
                // which item covers position base + lane of the concatenated leaves
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
                const uint32_t o_excl = __shfl(excl, (int)L, 64), o_slot = __shfl(my_slot, (int)L, 64);
                const uint32_t ray = __shfl(my_ray, (int)L, 64);
                const f3 ro = mk(__shfl(o.x, (int)ray, 64), __shfl(o.y, (int)ray, 64), __shfl(o.z, (int)ray, 64));
                const f3 rd = mk(__shfl(d.x, (int)ray, 64), __shfl(d.y, (int)ray, 64), __shfl(d.z, (int)ray, 64));
                const uint32_t slot = o_slot + (p - o_excl);
                const bool on = pair && !(ANY && ((decided >> ray) & 1ull));
                f3 a, e1, e2;
                uint32_t tri = 0;
                if (ANY) {  // geometry only (36 bytes)
                    const float* g = T.slot_geom + (size_t)(on ? slot : 0u) * 9;
                    a = mk(g[0], g[1], g[2]); e1 = mk(g[3], g[4], g[5]); e2 = mk(g[6], g[7], g[8]);
                } else {  // the whole record: the triangle index tells a second listing of the best triangle from a tie
                    const LeafTri* g = T.slots + (on ? slot : 0u);
                    const float4 r0 = reinterpret_cast<const float4*>(g)[0], r1 = reinterpret_cast<const float4*>(g)[1];
                    const float2 r2 = reinterpret_cast<const float2*>(g)[4];
                    a = mk(r0.x, r0.y, r0.z); e1 = mk(r0.w, r1.x, r1.y); e2 = mk(r1.z, r1.w, r2.x);
                    tri = __float_as_uint(r2.y);
                }
                float t = RT_INF;
                lanemask hit;
                if constexpr (ANY && FAST && kHandScheduledTests) {
                    hit = any_hit_exec_pairs<CULL>(ballot64(on), ro, rd, a, e1, e2, __shfl(t_limit, (int)ray, 64));  // renderer.cc:302 included
                } else {
                    hit = ray_triangle<CULL>(ballot64(on), ro, rd, a, e1, e2, t);
                }
                if (hit == 0) continue;
                if (ANY) {
                    lanemask good = hit;
                    if constexpr (!(FAST && kHandScheduledTests)) good = hit & ballot64(t < __shfl(t_limit, (int)ray, 64));  // renderer.cc:302
                    while (good != 0) {
                        const int g = __builtin_ctzll(good);
                        good &= good - 1ull;
                        decided |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)ray, g);
                    }
                } else {
                    bool shared = false;
                    if (lane_of(hit) && t < RT_INF) {
                        // renderer.cc:130: strictly smaller t wins; equal t keeps the smaller global slot
                        const unsigned long long old = atomicMin(&lds.best[ray], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)slot);
                        // the same distance from another slot: a second listing of the same triangle, or a tie of two triangles
                        if ((uint32_t)(old >> 32) == __float_as_uint(t) && (uint32_t)old != slot) shared = T.slots[(uint32_t)old].tri != tri;
                    }
                    lanemask sh = ballot64(shared);
                    while (sh != 0) {
                        const int g = __builtin_ctzll(sh);
                        sh &= sh - 1ull;
                        tie |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)ray, g);
                    }
                }
            }
#ifdef RT_LANE_PROFILE
            pb += __builtin_amdgcn_s_memtime() - pt0; pch += (total + 63) / 64; ppairs += total; ++pphases;
#endif
            continue;
        }
        if (ns == 0) break;
        // ---- node step: the top k tasks, lane i takes one
        uint32_t k = ns < 64u ? ns : 64u;
        {
            const uint32_t room_nodes = ns < kPoolNodeSoft ? (kPoolNodeSoft - ns) / 7u : 0u;
            const uint32_t room_leaves = (kPoolLeafCap - nl) / 8u;  // nl < 64 here
            k = k < room_nodes ? k : room_nodes;
            k = k < room_leaves ? k : room_leaves;
            k = k == 0 ? 1u : k;
        }
        uint32_t task = 0;
        if (lane < k) task = nstk[ns - 1u - lane];
        ns -= k;
        const uint32_t ray = task >> kPoolRecBits, rec = task & ((1u << kPoolRecBits) - 1u);
        const bool work = lane < k && !(ANY && ((decided >> ray) & 1ull));
        const f3 ro = mk(__shfl(o.x, (int)ray, 64), __shfl(o.y, (int)ray, 64), __shfl(o.z, (int)ray, 64));
        const f3 ri = mk(__shfl(inv.x, (int)ray, 64), __shfl(inv.y, (int)ray, 64), __shfl(inv.z, (int)ray, 64));
        const float r_lo = __shfl(sp.lo, (int)ray, 64);
        float r_hi;
        if (!ANY && prune) {  // the ray's best distance so far (keys are (bits of t) << 32 | slot; all ones = none)
            const uint32_t tb = (uint32_t)(__hip_atomic_load(&lds.best[ray], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 32);
            r_hi = tb == 0xffffffffu ? RT_INF : span_hi_closest(__uint_as_float(tb), __shfl(pad_abs, (int)ray, 64));
        } else {
            r_hi = __shfl(sp.hi, (int)ray, 64);
        }
        uint32_t pass_internal = 0, pass_leaf = 0;
        uint4 i0 = make_uint4(0, 0, 0, 0), i1 = i0, i2 = i0, i3 = i0;  // info[0..1], [2..3], [4..5], [6..7]
        if (work) {
            const uint4* ip = reinterpret_cast<const uint4*>(T.nodes + rec) + 4;
            i0 = ip[0]; i1 = ip[1]; i2 = ip[2]; i3 = ip[3];
            const uint32_t pass = lane_node_pass_natural(T.nodes, rec, ro, ri, r_lo, r_hi);
            const uint32_t internal = ((i0.y >> 31) & 1u) | ((i0.w >> 31) << 1) | ((i1.y >> 31) << 2) | ((i1.w >> 31) << 3) |
                                      ((i2.y >> 31) << 4) | ((i2.w >> 31) << 5) | ((i3.y >> 31) << 6) | ((i3.w >> 31) << 7);
            pass_internal = pass & internal;
            pass_leaf = pass & ~internal;
        }
        const uint32_t packed = (uint32_t)__builtin_popcount(pass_internal) | ((uint32_t)__builtin_popcount(pass_leaf) << 16);
        const uint32_t incl = wave_inclusive_sum(packed);
        const uint32_t excl = incl - packed;
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t at_n = ns + (excl & 0xffffu), at_l = nl + (excl >> 16);
        const uint32_t ray_n = ray << kPoolRecBits, ray_l = ray << kPoolCountBits;
#define RT_POOL_EMIT(c, ix, iy)                                 \
        if ((pass_internal >> (c)) & 1u) {                      \
            nstk[at_n] = (ix) | ray_n;                          \
            ++at_n;                                             \
        } else if ((pass_leaf >> (c)) & 1u) {                   \
            lslot[at_l] = (ix);                                 \
            lcount[at_l] = (iy) | ray_l;                        \
            ++at_l;                                             \
        }
        RT_POOL_EMIT(0, i0.x, i0.y)
        RT_POOL_EMIT(1, i0.z, i0.w)
        RT_POOL_EMIT(2, i1.x, i1.y)
        RT_POOL_EMIT(3, i1.z, i1.w)
        RT_POOL_EMIT(4, i2.x, i2.y)
        RT_POOL_EMIT(5, i2.z, i2.w)
        RT_POOL_EMIT(6, i3.x, i3.y)
        RT_POOL_EMIT(7, i3.z, i3.w)
#undef RT_POOL_EMIT
        ns += tot & 0xffffu;
        nl += tot >> 16;
        __builtin_amdgcn_wave_barrier();
#ifdef RT_LANE_PROFILE
        pa += __builtin_amdgcn_s_memtime() - pt0; ++psteps; ptasks += k;
#endif
    }
#ifdef RT_LANE_PROFILE
    RT_PPROF_ADD(0, pa); RT_PPROF_ADD(1, pb); RT_PPROF_ADD(2, psteps); RT_PPROF_ADD(3, ptasks); RT_PPROF_ADD(4, pch); RT_PPROF_ADD(5, ppairs); RT_PPROF_ADD(6, 1); RT_PPROF_ADD(7, pphases);
#endif
    if (ANY) {
        h.found = decided & vm;
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long key = __hip_atomic_load(&lds.best[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (valid && key != ~0ull) {
            h.t = __uint_as_float((uint32_t)(key >> 32));
            h.slot = (uint32_t)key;
            h.tri = T.slots[h.slot].tri;
        }
        tie &= vm;
        __builtin_amdgcn_wave_barrier();  // (the next walk of this wave reuses the pool)
    }
}

// valid: this lane carries a ray.  clip: see traverse<> in rt_traverse.h.  The root must be an internal node and leaves
// must hold fewer than 2^26 triangles (the host selects this form only then).
// FAST (any-hit only): unit-length directions in a scene within the bound of any_hit_exec.
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void traverse_pool(const DevOctree& T, PoolLds& lds, bool valid, f3 o, f3 d, float t_limit, bool clip, HitState& h) {
    hit_state_reset(h);
    if (T.root_info1 == 0 || ballot64(valid) == 0) return;  // empty tree / no rays
    lanemask tie = 0;
    if (ANY) {
        traverse_pool_walk<ANY, CULL, FAST>(T, lds, valid, o, d, t_limit, span_segment(o, t_limit, clip), false, h, tie);
        return;
    }
    bool todo = valid, pruned = clip;
#pragma nounroll
    for (;;) {
        HitState w;
        hit_state_reset(w);
        traverse_pool_walk<ANY, CULL, FAST>(T, lds, todo, o, d, t_limit, Span{span_lo_closest(o, pruned), RT_INF}, pruned, w, tie);
        h.t = todo ? w.t : h.t;
        h.slot = todo ? w.slot : h.slot;
        h.tri = todo ? w.tri : h.tri;
        if (!pruned || tie == 0) break;
        todo = lane_of(tie);
        pruned = false;
    }
}


}  // namespace rt
