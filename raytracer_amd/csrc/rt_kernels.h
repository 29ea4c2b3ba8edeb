// rt_kernels.h -- gfx950 kernels of the ray-scene query path (device code, included by rt_capi.hip).
//
// Kernel classes
//   k_trace_closest / k_trace_any : batched Search + closest-hit / any-hit loops for caller-supplied rays
//   k_closest                     : one recursion level of the renderer: ray generation (level 0) or queue
//                                   read, Search + closest hit, compaction of the hits into a work list
//   k_shadow                      : the any-hit pass: per hit, every area-light shadow sample of every light ->
//                                   one visibility bit each
//   k_shade                       : per hit: local illumination from the visibility bits, recursion gate,
//                                   reflection / refraction ray construction
//   k_resolve / k_resolve_pixels  : bottom-up evaluation of the recursion tree in the reference's order
//   k_search_count / k_search_fill: Ads::Search restated (breadth-first candidate lists)
#pragma once

// Minimum waves per SIMD the register allocator must leave room for in the wave-together forms of the two level kernels
// (their scalar register count sits at the edge of 8 waves; values live across the traversal are then kept in spare
// vector-register lanes).  0 = no constraint.
#ifndef RT_CLOSEST_WAVES
#define RT_CLOSEST_WAVES 8
#endif
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 0
#endif
#if RT_CLOSEST_WAVES
#define RT_CLOSEST_OCCUPANCY(MODE) __attribute__((amdgpu_waves_per_eu(((MODE) == 0 || (MODE) == 4) ? RT_CLOSEST_WAVES : 4, 8)))
#else
#define RT_CLOSEST_OCCUPANCY(MODE)
#endif
#if RT_SHADOW_WAVES
#define RT_SHADOW_OCCUPANCY(MODE) __attribute__((amdgpu_waves_per_eu(((MODE) == 0 || (MODE) == 3) ? RT_SHADOW_WAVES : 4, 8)))
#else
#define RT_SHADOW_OCCUPANCY(MODE)
#endif

#ifndef RT_XCD_CHUNK
#define RT_XCD_CHUNK 16
#endif

#include <hip/hip_runtime.h>

#include "rt_traverse.h"
#include "rt_traverse_lane.h"
#include "rt_traverse_pool.h"
#include "rt_stats.h"

namespace rt {

// How a kernel walks the tree.  MODE_WAVE: the wave together (rt_traverse.h); MODE_LANE: every lane its own ray
// (rt_traverse_lane.h); MODE_POOL: the rays of a wave share a pool of descent tasks (rt_traverse_pool.h).  The
// per-wave stacks of the last two are in LDS; a kernel declares one slot per wave of its block.  MODE_CLUSTER: the wave
// together over the clustered form of the tree (any-hit kernels only; a kernel of its own so that its extra registers
// do not cost the other wave-together kernels their occupancy).
enum { MODE_WAVE = 0, MODE_LANE = 1, MODE_POOL = 2, MODE_CLUSTER = 3, MODE_WAVE_PRUNED = 4 };
template <int MODE>
struct TraverseLds {
    uint32_t unused;
};
template <>
struct TraverseLds<MODE_POOL> {
    PoolLds s;
};
// The per-lane form keeps its stacks in dynamic LDS, sized at launch for the tree's depth (lane_stack_words).
extern __shared__ __attribute__((aligned(16))) uint32_t rt_dyn_lds[];  // (64-bit LDS atomics of the per-lane closest-hit walk)

// Launch options of the traversal kernels, packed into one kernel argument (the wave-together forms sit at the edge of
// their scalar register budget): bits 0..3 `sparse` (see ray_index), bit 4 `clip` (walk only the part of a ray that can
// contribute a result, see Span), bits 8..15 the tree's levels of internal nodes (depth of the per-lane stacks).
__host__ __device__ constexpr uint32_t trace_opts(uint32_t sparse, bool clip, uint32_t levels) { return sparse | (clip ? 16u : 0u) | (levels << 8); }
constexpr uint32_t kOptHitLists = 32u;  // k_shadow: the frame's camera is near enough for LightForm::hit_lists
__device__ __forceinline__ uint32_t opts_sparse(uint32_t opts) { return opts & 15u; }
__device__ __forceinline__ bool opts_clip(uint32_t opts) { return (opts & 16u) != 0u; }
__device__ __forceinline__ uint32_t opts_levels(uint32_t opts) { return (opts >> 8) & 0xffu; }
// a, e1, e2 of every leaf slot as 9 consecutive floats (DevOctree::slot_geom).
__global__ void __launch_bounds__(256) k_pack_slot_geometry(const LeafTri* __restrict__ slots, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LeafTri s = slots[i];
    float* g = out + (size_t)i * 9;
    g[0] = s.ax; g[1] = s.ay; g[2] = s.az;
    g[3] = s.e1x; g[4] = s.e1y; g[5] = s.e1z;
    g[6] = s.e2x; g[7] = s.e2y; g[8] = s.e2z;
}

// Thread index -> ray index: one ray per lane, and only the first 64 >> sparse lanes of a wave carry rays (sparse > 0:
// small launches of the pool form, whose idle lanes still work on the wave's tasks -- more, shorter waves).
// ok: the lane carries a ray.
__device__ __forceinline__ uint64_t ray_index(uint64_t g, uint32_t sparse, bool& ok) {
    const uint32_t lane = (uint32_t)g & 63u;
    ok = lane < (64u >> sparse);
    return ((g >> 6) << (6u - sparse)) + lane;
}
// FAST: any-hit rays with unit-length directions in a scene within the bound of any_hit_exec.
// BASES (per-lane any-hit walks only): see traverse_lane_any.
template <bool ANY, bool CULL, int MODE, bool FAST = false, bool BASES = false>
__device__ __forceinline__ void trace(const DevOctree& T, TraverseLds<MODE>& lds, uint32_t opts, bool valid, f3 o, f3 d, float t_limit, HitState& h,
                                      uint32_t rec_base = 0, uint32_t slot_base = 0) {
    static_assert(!BASES || (ANY && MODE == MODE_LANE), "");
    const bool clip = opts_clip(opts);
    if constexpr (MODE == MODE_LANE) {
        const uint32_t levels = opts_levels(opts);
        const LaneStack stk = lane_stack(rt_dyn_lds + (threadIdx.x >> 6) * lane_stack_words(levels, !ANY), levels, !ANY);
        traverse_lane<ANY, CULL, ANY && FAST, BASES>(T, stk, valid, o, d, t_limit, clip, h, rec_base, slot_base);
    } else if constexpr (MODE == MODE_POOL) {
        traverse_pool<ANY, CULL, ANY && FAST>(T, lds.s, valid, o, d, t_limit, clip, h);
    } else if constexpr (MODE == MODE_CLUSTER) {
        static_assert(ANY, "the clustered form serves any-hit queries");
        traverse_any_clustered<CULL, FAST>(T, valid, o, d, t_limit, clip, h);
    } else {
        // Closest-hit walks of the wave are pruned by the best hit only in an instantiation of their own
        // (MODE_WAVE_PRUNED, closest-hit kernels only): the ordered, pruned walk more than doubles the kernel's code, and
        // the small trees that MODE_WAVE serves by default never use it.
        static_assert(MODE != MODE_WAVE_PRUNED || !ANY, "");
        traverse<ANY, CULL, FAST>(T, valid, o, d, t_limit, ANY ? clip : (MODE == MODE_WAVE_PRUNED), h);
    }
}

struct alignas(16) TriShade {  // per triangle in Build order
    float nx, ny, nz;          // face normal (src/triangle.h:28-32)
    int32_t mat;               // Triangle::material_id
    int32_t light_eq;          // first light l with lights[l] == triangle (src/triangle.h:15-19), else -1
    uint32_t spawn;            // secondary rays a hit can spawn at most: non-zero specular + non-zero transmittance (renderer.cc:179,185)
    uint32_t dark;             // kDark* bits: which shadow rays of a hit on this triangle need no answer (shadow_ray_idle)
    int32_t pad;
};
static_assert(sizeof(TriShade) == 32, "");

struct alignas(16) MatDev {  // src/material.h:7-16, plus the two epsilon-compare results of renderer.cc:179,185
    float diffuse[3], specular[3], transmittance[3], emission[3];
    float shininess, ior;
    uint32_t has_specular, has_transmittance;
};
static_assert(sizeof(MatDev) == 64, "");

struct alignas(16) LightDev {  // one emissive triangle (obj_loader.cc:188-191)
    float a[3], e1[3], e2[3];  // vertex a, b - a, c - a (src/triangle.h:47-51)
    float n[3];                // its face normal
    float area;                // src/triangle.h:34-40
    float emission[3];
};
static_assert(sizeof(LightDev) == 64, "");

// What the any-hit passes towards ONE light walk instead of the scene's full small-tree / clustered arrays: the same
// arrays without the triangles that can never occlude that light when back faces are culled (make_light_forms in
// rt_capi.hip states the condition).  Null pointers: that form does not exist for the tree.
struct alignas(16) LightForm {
    const LeafTri* flat_tris;  // small-tree form: the light's triangle list (whole 4-record blocks, at least one)
    const NodeRec* cl_nodes;   // clustered form: the node records with the light's list positions ...
    const LeafTri* cl_tris;    // ... and its cluster lists
    uint32_t n_flat_tris;
    // per-lane form (big trees): the node records with the light's leaf ranges -- octants and subtrees left without a
    // triangle are marked empty -- and the geometry of the slots that remain (DevOctree::nodes / slot_geom / root_info)
    uint32_t lane_root_info1;
    const NodeRec* lane_nodes;
    const float* lane_slot_geom;
    // the same copies seen from light 0's arrays (all lights' copies lie one after the other): what a lane adds to the
    // record and slot indices of its light's copy when the lanes of a wave walk different lights' trees
    uint32_t lane_rec_base, lane_slot_base;
    // small-tree form, per hit triangle h: where (records after flat_tris) and how long the light's list is without the
    // triangles that have all of h behind them (make_light_forms); nullptr: no such lists
    const uint2* hit_lists;
};
static_assert(sizeof(LightForm) == 64, "");

struct alignas(16) RayRec {  // one node of the recursion tree waiting to be traced
    float o[3], d[3];
    uint32_t parent;   // index of the parent node in the previous level
    uint32_t pixel;    // global pixel index (key of the random generator)
    uint32_t node_id;  // 1 = camera ray, 2n = reflection child of n, 2n+1 = refraction child of n
    uint32_t pad[3];
};
static_assert(sizeof(RayRec) == 48, "");

struct alignas(16) NodeOut {  // what ray_color (renderer.cc:114-195) knows about one node before its children return
    float c[3];               // clamped local colour (renderer.cc:170), or the background on a miss
    uint32_t meta;            // bit 0: has a reflection child, bit 1: a refraction child; bits 2..: material of the hit
};
static_assert(sizeof(NodeOut) == 16, "");
// The colours the children return (I_R, I_T of renderer.cc:180-189) live beside the nodes, 6 floats per node, and are
// only touched for nodes that have children -- a level-0 node is written once per hit and read once per pixel, and
// 16 bytes instead of 64 is what those two passes move.
constexpr uint32_t kNodeChildFloats = 6;

struct SceneDev {
    DevOctree tree;
    const LeafTri* tri_geom;   // per triangle in Build order: a, b-a, c-a
    const TriShade* tri_shade;
    const float* tri_vnorm;    // 9 floats per triangle (vertex normals) or nullptr
    const MatDev* mats;
    const LightDev* lights;
    uint32_t n_lights;
    uint32_t lights_merged;        // all lights share one set of forms (light_forms[l] are equal): see make_light_forms
    const LightForm* light_forms;  // per light, or nullptr (culling off at creation, too many lights, no such forms)
};

struct FrameParams {
    f3 cam_pos, p00, qw, qh, bg;
    int32_t width, height, tiles_x;
    int32_t band_rank, band_world;
    int32_t render_type, shading_type, cull;
    int32_t max_depth, spt;
    float max_distance;
    uint32_t seed, spp_sample;
    int32_t spp;
};

// Per-level counters.  Each sits in its own 128-byte line: atomics on one line are served one after the other.
// CNT_WORK: hits of the level (length of its work list); CNT_NEXT: rays appended to the next level; CNT_SHADOW: shadow
// rays the hits cost (64 bits); CNT_SPAWN: upper bound of CNT_NEXT known after the closest-hit pass (hits whose material
// reflects or refracts); CNT_TICKET: blocks of the closest-hit pass that have finished.  These are zeroed for every pixel
// sample.  CNT_SEQ (launches of the level's closest-hit pass so far) and CNT_SLOT (address of the level's LevelSlot in
// pinned host memory, 64 bits, 0 = nobody listens) are set up by k_frame_begin and live for the whole frame; keeping them
// here rather than in kernel arguments keeps the traversal kernels' scalar register count where it was.
// CNT_DEEPEST (k_tail): the deepest recursion level a path of the fused tail reached, + 1.
// CNT_IDLE: shadow rays of the level that were not traced (shadow_ray_idle); they are part of CNT_SHADOW.  Every wave
// of the any-hit pass may add to it, so it is kCntIdleLines partial sums in lines of their own (a wave picks one by its
// block index).
constexpr uint32_t kCntIdleLines = 8;
enum { CNT_WORK = 0, CNT_NEXT = 32, CNT_SHADOW_LO = 64, CNT_SHADOW_HI = 65, CNT_SPAWN = 96, CNT_TICKET = 128, CNT_SEQ = 129,
       CNT_SLOT_LO = 130, CNT_SLOT_HI = 131, CNT_DEEPEST = 132, CNT_IDLE = 160, CNT_WORDS = 160 + 32 * kCntIdleLines };

// What the host needs to know about a level to size the launches that follow, written by the last block of the
// level's closest-hit pass straight into pinned host memory (flag last, system-scope release): the host polls the
// flag instead of synchronising the stream, while the device already runs the level's any-hit pass.
struct LevelSlot {
    uint32_t n;          // rays of the level
    uint32_t n_work;     // hits
    uint32_t spawn;      // upper bound of the next level's rays
    uint32_t shadow_lo, shadow_hi;
    uint32_t pad[2];
    uint32_t flag;       // CNT_SEQ of the launch that filled the slot
};
static_assert(sizeof(LevelSlot) == 32, "");

// Frame totals kept on the device between the samples of a frame (k_roll_counters) and mirrored into pinned host memory.
struct FrameTotals {
    unsigned long long secondary_rays, shadow_rays, deepest_level, idle_shadow_rays;
};

// ---- helpers ---------------------------------------------------------------------------------------------------

// Level-0 node index -> pixel.  64 consecutive nodes are one 8x8 pixel tile = one wavefront.
__device__ __forceinline__ bool node0_pixel(const FrameParams& F, uint32_t node, int& x, int& y) {
    const uint32_t tile = node >> 6, lane = node & 63;
    const uint32_t local_band = tile / (uint32_t)F.tiles_x, tx = tile % (uint32_t)F.tiles_x;
    const uint32_t band = local_band * (uint32_t)F.band_world + (uint32_t)F.band_rank;
    x = (int)(tx * 8 + (lane & 7));
    y = (int)(band * 8 + (lane >> 3));
    return x < F.width && y < F.height;
}

// src/Renderer/renderer.cc:104-106 (+ :44-46 for the jittered samples of fuzzysampling).
__device__ __forceinline__ f3 primary_dir(const FrameParams& F, int x, int y, uint32_t pixel) {
    const f3 pxy = (F.p00 + (F.qw * (float)x)) - (F.qh * (float)y);
    f3 dir = div_guard(pxy, length(pxy));
    if (F.spp_sample > 0) {
        const uint32_t pre = rng_prefix(F.seed, pixel, F.spp_sample, 0);
        const float xj = u31_to_float(rng_finish(pre, 0, 0, 0)) * 0.001f;
        const float yj = u31_to_float(rng_finish(pre, 0, 0, 1)) * 0.001f;
        dir = dir + mk(xj, yj, 0.0f);
    }
    return dir;
}

// Hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2.  This maps the hardware
// block index to a logical one such that an XCD receives runs of RT_XCD_CHUNK consecutive logical blocks: neighbouring
// rays (which walk the same part of the tree) then share an L2 instead of spreading over all eight.  Short runs keep
// the eight XCDs equally loaded when cost varies over the image.
__device__ __forceinline__ uint32_t xcd_block() {
#if RT_XCD_CHUNK > 1
    constexpr uint32_t kGroup = 8u * RT_XCD_CHUNK;
    const uint32_t b = blockIdx.x;
    const uint32_t g = b / kGroup;
    if ((g + 1u) * kGroup > gridDim.x) return b;  // the incomplete last group keeps the hardware order
    const uint32_t r = b - g * kGroup;
    return g * kGroup + (r & 7u) * RT_XCD_CHUNK + (r >> 3);
#else
    return blockIdx.x;
#endif
}

// Appends one item per lane with `want` set to a global list; returns the lane's slot.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool want) {
    const uint64_t m = ballot64(want);
    if (m == 0) return 0;
    uint32_t base = 0;
    const uint32_t lane = __lane_id();
    const int leader = __builtin_ctzll(m);
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    const uint64_t below = m & ((1ull << lane) - 1ull);
    return base + (uint32_t)__builtin_popcountll(below);
}

// The same for a whole block (up to 1024 threads) with ONE atomic per block: a single counter word sustains only some
// tens of atomics per microsecond, and a 1080p level has 32k waves.  Every thread of the block must call it (once per
// kernel).  `weight` is summed over the block into the 64-bit counter CNT_SHADOW and `spawn` into CNT_SPAWN (one atomic
// each per block).  The block that finishes last publishes the level's totals in the level's LevelSlot, if it has one.
constexpr int kMaxBlockWaves = 16;
__device__ __forceinline__ uint32_t block_append(uint32_t* counters, bool want, uint32_t weight, uint32_t spawn, uint32_t n_level) {
    __shared__ uint32_t s_count[kMaxBlockWaves];
    __shared__ uint32_t s_weight[kMaxBlockWaves];
    __shared__ uint32_t s_spawn[kMaxBlockWaves];
    __shared__ uint32_t s_base;
    const uint64_t m = ballot64(want);
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    uint32_t w = weight, sp = spawn;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        w += __shfl_xor(w, d, 64);
        sp += __shfl_xor(sp, d, 64);
    }
    if (lane == 0) {
        s_count[wave] = (uint32_t)__builtin_popcountll(m);
        s_weight[wave] = w;
        s_spawn[wave] = sp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0, wsum = 0, ssum = 0;
        for (uint32_t k = 0; k < n_waves; ++k) {
            total += s_count[k];
            wsum += s_weight[k];
            ssum += s_spawn[k];
        }
        // The sums go in with returning atomics whose results are consumed before the ticket is taken: a returned atomic
        // has been performed at the level where all XCDs see it, so no fence is needed between sums and ticket (an
        // agent-scope fence writes back the whole L2 of the XCD -- per block, that costs more than the traversal).
        // This leans on the hardware, not on the memory model: device-scope atomics of gfx942 / gfx950 are executed at
        // the memory side of the fabric, in issue order per address, and the empty asm below keeps the compiler from
        // moving the ticket above the sums.  A stale total would show at once: the host sizes the next launches from
        // these numbers and every frame's ray counts are compared with the oracle's in the tests.
        const uint32_t r0 = atomicAdd(&counters[CNT_WORK], total);
        s_base = r0;
        const unsigned long long r1 = atomicAdd((unsigned long long*)&counters[CNT_SHADOW_LO], (unsigned long long)wsum);
        const uint32_t r2 = atomicAdd(&counters[CNT_SPAWN], ssum);
        asm volatile("" ::"v"(r0), "v"(r1), "v"(r2) : "memory");
        if (atomicAdd(&counters[CNT_TICKET], 1u) == gridDim.x - 1u) {  // every other block has added its sums
            LevelSlot* slot = reinterpret_cast<LevelSlot*>((uintptr_t)counters[CNT_SLOT_LO] | ((uintptr_t)counters[CNT_SLOT_HI] << 32));
            const uint32_t seq = ++counters[CNT_SEQ];
            if (slot) {
                const unsigned long long sh = atomicAdd((unsigned long long*)&counters[CNT_SHADOW_LO], 0ull);
                slot->n = n_level;
                slot->n_work = atomicAdd(&counters[CNT_WORK], 0u);
                slot->spawn = atomicAdd(&counters[CNT_SPAWN], 0u);
                slot->shadow_lo = (uint32_t)sh;
                slot->shadow_hi = (uint32_t)(sh >> 32);
                __threadfence_system();
                __hip_atomic_store(&slot->flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    uint32_t at = s_base;
    for (uint32_t k = 0; k < wave; ++k) at += s_count[k];
    const uint64_t below = m & ((1ull << lane) - 1ull);
    return at + (uint32_t)__builtin_popcountll(below);
}

// ---- batched queries for caller-supplied rays --------------------------------------------------------------------

template <bool CULL, int MODE>
__global__ void __launch_bounds__(256) k_trace_closest(DevOctree T, const float* __restrict__ orig, const float* __restrict__ dir,
                                                       uint64_t n, float* __restrict__ out_t, int32_t* __restrict__ out_tri, uint32_t opts) {
    bool lane_ok;
    const uint64_t i = ray_index((uint64_t)xcd_block() * blockDim.x + threadIdx.x, opts_sparse(opts), lane_ok);
    const bool valid = lane_ok && i < n;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (valid) {
        o = mk(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]);
        d = mk(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
    }
    __shared__ TraverseLds<MODE> lds[4];
    HitState h;
    trace<false, CULL, MODE>(T, lds[threadIdx.x >> 6], opts, valid, o, d, 0.0f, h);
    if (valid) {
        out_t[i] = h.t;
        out_tri[i] = (h.t == RT_INF) ? -1 : (int32_t)h.tri;  // renderer.cc:140
    }
}

template <bool CULL, int MODE, bool FAST = false>  // FAST: see trace<> (the directions are normalised here)
__global__ void __launch_bounds__(256) k_trace_any(DevOctree T, const float* __restrict__ point, const float* __restrict__ light,
                                                   uint64_t n, uint8_t* __restrict__ out, uint32_t opts) {
    bool lane_ok;
    const uint64_t i = ray_index((uint64_t)xcd_block() * blockDim.x + threadIdx.x, opts_sparse(opts), lane_ok);
    const bool valid = lane_ok && i < n;
    f3 p = mk(0, 0, 0), l = mk(0, 0, 1);
    if (valid) {
        p = mk(point[i * 3], point[i * 3 + 1], point[i * 3 + 2]);
        l = mk(light[i * 3], light[i * 3 + 1], light[i * 3 + 2]);
    }
    // renderer.cc:289-290
    const f3 pl = l - p;
    const float dist = length(pl);
    const f3 d = normalize(pl);
    __shared__ TraverseLds<MODE> lds[4];
    HitState h;
    trace<true, CULL, MODE, FAST>(T, lds[threadIdx.x >> 6], opts, valid, p, d, dist, h);
    if (valid) out[i] = lane_of(h.found) ? 1 : 0;
}

// The instrumented pass over the same caller-supplied rays (rt_stats.h): ANY -> second = light points.
template <bool ANY, bool CULL>
__global__ void __launch_bounds__(256) k_stats_rays(DevOctree T, const float* __restrict__ first, const float* __restrict__ second, uint64_t n,
                                                    unsigned long long* __restrict__ totals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    float t_limit = 0.0f;
    if (valid) {
        o = mk(first[i * 3], first[i * 3 + 1], first[i * 3 + 2]);
        d = mk(second[i * 3], second[i * 3 + 1], second[i * 3 + 2]);
    }
    if (ANY) {  // renderer.cc:289-290
        const f3 pl = d - o;
        t_limit = length(pl);
        d = normalize(pl);
    }
    RayCounters c;
    stats_ray<ANY, CULL>(T, valid, o, d, t_limit, c);
    stats_flush<ANY>(c, totals);
}

// Diagnostics: every (ray, triangle) pair through the three forms of the ray/triangle test -- the compiled one
// (ray_triangle<>), the hand-scheduled any-hit form with the triangle in scalar registers (any_hit_exec) and the one with
// the triangle per lane (any_hit_exec_pairs) -- with an unbounded limit, so that each reports every usable hit.
// out[form][ray * n_tris + tri] = the distance the form computed, +inf where it reports no hit.
template <bool CULL>
__global__ void __launch_bounds__(256) k_debug_triangle_tests(const float* __restrict__ orig, const float* __restrict__ dir, uint32_t n_rays,
                                                              const LeafTri* __restrict__ tris, uint32_t n_tris, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n_rays;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (valid) {
        o = mk(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]);
        d = mk(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
    }
    const lanemask all = ballot64(valid);
    const size_t plane = (size_t)n_rays * n_tris;
    for (uint32_t j = 0; j < n_tris; ++j) {
        cu32_p rec = as_const_u(tris) + (size_t)j * 12;  // wave-uniform: scalar loads
        const float ax = __uint_as_float(rec[0]), ay = __uint_as_float(rec[1]), az = __uint_as_float(rec[2]);
        const float e1x = __uint_as_float(rec[3]), e1y = __uint_as_float(rec[4]), e1z = __uint_as_float(rec[5]);
        const float e2x = __uint_as_float(rec[6]), e2y = __uint_as_float(rec[7]), e2z = __uint_as_float(rec[8]);
        float t0 = RT_INF;
        const lanemask h0 = ray_triangle<CULL>(all, o, d, mk(ax, ay, az), mk(e1x, e1y, e1z), mk(e2x, e2y, e2z), t0);
        lanemask h1 = 0;
        float t1 = RT_INF;
        any_hit_exec<CULL>(all, o, d, ax, ay, az, e1x, e1y, e1z, e2x, e2y, e2z, RT_INF, h1, 0, 0, &t1);
        const LeafTri g = tris[j];  // per lane: vector registers
        float t2 = RT_INF;
        const lanemask h2 = any_hit_exec_pairs<CULL>(all, o, d, mk(g.ax, g.ay, g.az), mk(g.e1x, g.e1y, g.e1z), mk(g.e2x, g.e2y, g.e2z), RT_INF, &t2);
        if (valid) {
            const size_t at = (size_t)i * n_tris + j;
            out[at] = lane_of(h0) ? t0 : RT_INF;
            out[plane + at] = lane_of(h1) ? t1 : RT_INF;
            out[2 * plane + at] = lane_of(h2) ? t2 : RT_INF;
        }
    }
}

// ---- renderer level kernels ------------------------------------------------------------------------------------

__device__ __forceinline__ void load_ray(const FrameParams& F, const RayRec* __restrict__ rays, uint32_t level, uint32_t i, uint32_t n,
                                         bool& valid, f3& o, f3& d, uint32_t& pixel, uint32_t& node_id) {
    valid = i < n;
    o = mk(0, 0, 0);
    d = mk(0, 0, 1);
    pixel = 0;
    node_id = 1;
    if (level == 0) {
        int x = 0, y = 0;
        valid = valid && node0_pixel(F, i, x, y);
        if (valid) {
            pixel = (uint32_t)y * (uint32_t)F.width + (uint32_t)x;
            o = F.cam_pos;
            d = primary_dir(F, x, y, pixel);
        }
    } else if (valid) {
        const RayRec r = rays[i];
        o = mk(r.o[0], r.o[1], r.o[2]);
        d = mk(r.d[0], r.d[1], r.d[2]);
        pixel = r.pixel;
        node_id = r.node_id;
    }
}

// Search + closest-hit loop (renderer.cc:116-142) for every node of one level.
// BIG: launched with 1024-thread blocks (a quarter of the compaction atomics; only for the wave-together form on small
// trees, see closest_block()); otherwise 256-thread blocks.
template <bool CULL, int MODE, bool BIG>
__global__ void __launch_bounds__(BIG ? 1024 : 256) RT_CLOSEST_OCCUPANCY(MODE) k_closest(SceneDev S, FrameParams F, uint32_t level, const RayRec* __restrict__ rays, uint32_t n,
                                                 float* __restrict__ hit_t, int32_t* __restrict__ hit_tri, NodeOut* __restrict__ out,
                                                 uint32_t* __restrict__ work, uint32_t* __restrict__ counters, uint32_t opts) {
    // n: the rays of level 0; for deeper levels an upper bound, the number itself being the previous level's CNT_NEXT
    // (`counters` is the level's own block; the blocks of the levels are consecutive).  Threads beyond it carry no ray.
    if (level > 0) n = counters[CNT_NEXT - CNT_WORDS];
    bool lane_ok;
    const uint32_t i = (uint32_t)ray_index((uint64_t)xcd_block() * blockDim.x + threadIdx.x, opts_sparse(opts), lane_ok);
    bool valid;
    f3 o, d;
    uint32_t pixel, node_id;
    load_ray(F, rays, level, i, lane_ok ? n : 0u, valid, o, d, pixel, node_id);
    __shared__ TraverseLds<MODE> lds[BIG ? 16 : 4];
    HitState h;
    trace<false, CULL, MODE>(S.tree, lds[threadIdx.x >> 6], opts, valid, o, d, 0.0f, h);
    const bool hit = valid && !(h.t == RT_INF);
    if (valid) {
        hit_t[i] = h.t;
        hit_tri[i] = hit ? (int32_t)h.tri : -1;
        if (!hit) {  // renderer.cc:140-142: the background colour is this node's value
            NodeOut r;
            r.c[0] = F.bg.x; r.c[1] = F.bg.y; r.c[2] = F.bg.z;
            r.meta = 0;
            out[i] = r;
        }
    }
    // The shadow / area-light rays this hit will cost (renderer.cc:206-219): samples_per_triangle for every light
    // before the one the hit triangle itself is (the light loop returns there), all lights otherwise.
    // ... and the secondary rays it can spawn at most (renderer.cc:177-192: one per non-zero specular / transmittance)
    uint32_t shadow_rays = 0, spawn = 0;
    if (hit) {
        const TriShade ts = S.tri_shade[h.tri];
        if (F.render_type >= 2 && F.spt > 0) {
            const uint32_t lights = (ts.light_eq >= 0 && (uint32_t)ts.light_eq < S.n_lights) ? (uint32_t)ts.light_eq : S.n_lights;
            shadow_rays = lights * (uint32_t)F.spt;
        }
        spawn = ts.spawn;  // (the host ignores the bound at the last level)
    }
    const uint32_t at = block_append(counters, hit, shadow_rays, spawn, n);
    if (hit) work[at] = i;
}

// src/triangle.h:54-77
__device__ __forceinline__ f3 interpolate_normal(const LeafTri& g, const float* __restrict__ vn, f3 P) {
    const f3 v0 = mk(g.e1x, g.e1y, g.e1z), v1 = mk(g.e2x, g.e2y, g.e2z);
    const f3 v2 = P - mk(g.ax, g.ay, g.az);
    const float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    const float v = (d11 * d20 - d01 * d21) / denom;
    const float w = (d00 * d21 - d01 * d20) / denom;
    const float u = 1.0f - v - w;
    const f3 n0 = mk(vn[0], vn[1], vn[2]), n1 = mk(vn[3], vn[4], vn[5]), n2 = mk(vn[6], vn[7], vn[8]);
    return normalize(((n0 * u) + (n1 * v)) + (n2 * w));
}

// Area-light sample point (src/triangle.h:42-52) for (light l, sample s) of the path node whose generator
// prefix is `pre`.
__device__ __forceinline__ f3 light_sample(const LightDev& L, uint32_t pre, uint32_t l, uint32_t s) {
    const float r1 = u31_to_float(rng_finish(pre, l, s, 0));
    const float r2 = u31_to_float(rng_finish(pre, l, s, 1));
    const float uu = (r1 + r2 > 1.0f) ? 1.0f - r1 : r1;
    const float vv = (r1 + r2 > 1.0f) ? 1.0f - r2 : r2;
    const f3 la = mk(L.a[0], L.a[1], L.a[2]), le1 = mk(L.e1[0], L.e1[1], L.e1[2]), le2 = mk(L.e2[0], L.e2[1], L.e2[2]);
    return (la + (le1 * uu)) + (le2 * vv);
}

// Shadow rays that need no answer.  A visible light sample adds c = (I_a + I_d) + I_s (+ the material's emission with
// Phong, renderer.cc:264-266) to the hit's colour, I_l = emission * w with w = area * max(0, dot(n_l, -d_l)) / (...)
// (renderer.cc:220-240).  Where every term of c is a zero whatever the sample's visibility, the sum -- which starts at
// +0 -- is the same bits with or without the sample, so its shadow ray is not traced and the sample counts as shadowed:
//   A. the light faces away from the hit (max(0, dot(n_l, -d_l)) is 0): w = 0, so I_l, I_a, I_d and I_s are zeros;
//   B. the surface faces away from the light (max(0, dot(N, d_l)) is 0) and its specular colour is zero: I_d and I_s
//      are zeros (Phong's specular term has no such factor, hence the condition on the colour).
// "Zero times x is zero" needs x finite: the host sets the bits only for scenes whose lights and materials are finite
// and moderate (rt_scene_set_shading), the shininess in [0, 1e6] (a finite power), and here the ray must have a finite
// length.  N is the face normal, so rule B is for flat shading only; rule A also holds with smooth shading, whose
// interpolated normal is a normalised vector -- components of magnitude <= 1 or NaN, and max(0, NaN) is 0.  Both tests
// use the very expressions of the shading code, so a sample is skipped exactly when that code would multiply by its zero.  On a closed mesh the skipped rays are the
// expensive ones: with back faces culled a ray that starts on the far side of a sphere runs through its whole interior.
constexpr uint32_t kDarkFinite = 1u;        // rule A may be used (and rule B with kDarkSpecZero)
constexpr uint32_t kDarkEmissionZero = 2u;  // the material's emission is zero: needed with Phong
constexpr uint32_t kDarkSpecZero = 4u;      // the material's specular colour is zero
__device__ __forceinline__ bool shadow_ray_idle(const FrameParams& F, uint32_t dark, f3 N, f3 n_l, f3 d_l, float dist) {
    if (!(dark & kDarkFinite)) return false;
    if (F.render_type == 2 && !(dark & kDarkEmissionZero)) return false;
    if (!(dist < RT_INF)) return false;
    const bool light_faces_away = !(0.0f < dot(n_l, -d_l));
    const bool surface_faces_away = F.shading_type == 0 && !(0.0f < dot(N, d_l)) && (dark & kDarkSpecZero) != 0u;
    return light_faces_away || surface_faces_away;
}

// The any-hit pass: every shadow / area-light ray of every hit of one level (renderer.cc:217-219, 288-313).
// A hit owns 2^gshift neighbouring lanes.  The (light, sample) pairs of the hit are numbered q = light * 2^pshift +
// sample (2^pshift = the samples-per-light count rounded up to a power of two) and lane `sub` of the group traces
// q = sub, sub + 2^gshift, ...: the 64 rays a wave traverses together leave from 64 >> gshift surface points towards
// the lights, rays of one origin side by side, and a launch has enough waves to fill the chip even for a few thousand
// hits.  The result is one visibility bit per (light, sample): vis[(l * words + s / 32) * n_work + j] bit s % 32 = 1
// if not shadowed (hit j innermost); the bits of a word come from one ballot, so no two lanes write the same word.
// WAVES: waves per block.  The kernel has no block-level step, so the per-lane and pool forms run one-wave blocks:
// a wave that finishes frees its slot at once instead of waiting for the slowest of four.
// FAST: see trace<>.
// SIDE (per-lane form with per-light trees only): all lights of a hit side by side, every lane walking the tree of its
// own light (two more vector registers: 6 instead of 7 waves per SIMD, so it is an instantiation of its own).
template <bool CULL, int MODE, int WAVES, bool FAST = false, bool SIDE = false>
__global__ void __launch_bounds__(WAVES * 64) RT_SHADOW_OCCUPANCY(MODE) k_shadow(SceneDev S, FrameParams F, uint32_t level, const RayRec* __restrict__ rays, uint32_t n,
                                                const float* __restrict__ hit_t, const int32_t* __restrict__ hit_tri,
                                                const uint32_t* __restrict__ work, uint32_t* __restrict__ vis, uint32_t words,
                                                uint32_t pshift, uint32_t gshift, uint32_t* __restrict__ counters, uint32_t opts) {
    __shared__ TraverseLds<MODE> lds[WAVES];
    const uint32_t group = 1u << gshift;  // sparse launches: group <= 64 >> sparse, so a group stays inside the lanes of a wave
    bool lane_ok;
    const uint64_t ray_slot = ray_index((uint64_t)xcd_block() * blockDim.x + threadIdx.x, opts_sparse(opts), lane_ok);
    const uint32_t j = (uint32_t)(ray_slot >> gshift);
    const uint32_t sub = (uint32_t)ray_slot & (group - 1u);
    const uint32_t n_work = counters[CNT_WORK];
    const bool active = lane_ok && j < n_work;
    uint32_t i = 0;
    if (active) i = work[j];
    bool valid;
    f3 o, d;
    uint32_t pixel, node_id;
    load_ray(F, rays, level, i, active ? n : 0, valid, o, d, pixel, node_id);
    valid = valid && active;
    // Rays that need no answer (shadow_ray_idle) are left out where every lane walks its own ray: the lanes they free
    // take over other rays' subtrees or tasks, and waves without a ray end at once.  A wave that walks its rays together
    // only gains if all 64 are left out, which does not pay for the test (configs with the reference's integer aspect
    // ratio, whose rays all lie in one horizontal plane, have no such hits at all).
    constexpr bool kSkipIdle = MODE == MODE_LANE || MODE == MODE_POOL;
    float t = 0.0f;
    uint32_t tri = 0;
    if (valid) {
        t = hit_t[i];
        tri = (uint32_t)hit_tri[i];
    }
    const f3 P = o + (d * t);  // renderer.cc:146
    const uint32_t pre = rng_prefix(F.seed, pixel, F.spp_sample, node_id);
    // renderer.cc:212-214: the light loop ends at the light the hit triangle itself is
    uint32_t lights_traced = 0;
    if constexpr (!kSkipIdle) {
        const int32_t light_eq = valid ? S.tri_shade[tri].light_eq : -1;
        lights_traced = !valid ? 0u : ((light_eq >= 0 && (uint32_t)light_eq < S.n_lights) ? (uint32_t)light_eq : S.n_lights);
    }
    // the lanes whose ballot bits form one visibility word: the samples of one light, or 32 of them
    const uint32_t per_light = 1u << pshift;
    const uint32_t word_lanes = per_light < 32u ? per_light : 32u;
    const uint32_t word_first_lane = (uint32_t)__lane_id() & ~(word_lanes - 1u);
    const uint32_t word_mask = word_lanes == 32u ? 0xffffffffu : ((1u << word_lanes) - 1u);
    const bool writer = (sub & (word_lanes - 1u)) == 0u;
    const uint32_t q_end = S.n_lights << pshift;
    for (uint32_t q0 = 0; q0 < q_end; q0 += group) {
        const uint32_t q = q0 + sub;
        const uint32_t l = q >> pshift, smp = q & (per_light - 1u);
        uint32_t dark = 0;
        f3 N = mk(0, 0, 0);
        if constexpr (kSkipIdle) {
            // The hit triangle's record is read again in every pass (the empty statement keeps the read inside the
            // loop): the triangle index is one register to carry through the walks below, the record's fields five.
            asm volatile("" : "+v"(tri));
            const TriShade ts = S.tri_shade[tri];
            lights_traced = !valid ? 0u : ((ts.light_eq >= 0 && (uint32_t)ts.light_eq < S.n_lights) ? (uint32_t)ts.light_eq : S.n_lights);
            dark = ts.dark;
            N = mk(ts.nx, ts.ny, ts.nz);
        }
        const bool lit = l < lights_traced;
        bool on = lit && smp < (uint32_t)F.spt;
        const lanemask on_mask = ballot64(on);
        if (on_mask == 0) continue;
        // one light per pass (group == samples per light): its record is wave-uniform and comes through scalar loads
        const uint32_t l_load = gshift <= pshift ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(q0 >> pshift)) : (lit ? l : 0u);
        const LightDev& L = S.lights[l_load];
        const f3 p_l = light_sample(L, pre, l, smp);
        // renderer.cc:289-290
        const f3 to_l = p_l - P;
        const float dist = length(to_l);
        const f3 d_l = normalize(to_l);
        // samples that add nothing to the hit's colour whether visible or not: not traced, reported as shadowed
        lanemask idle_mask = 0;
        if constexpr (kSkipIdle) {
            const bool idle = on && shadow_ray_idle(F, dark, N, mk(L.n[0], L.n[1], L.n[2]), d_l, dist);
            idle_mask = ballot64(idle);
            on = on && !idle;
            if (idle_mask != 0 && __lane_id() == 0)
                atomicAdd(&counters[CNT_IDLE + 32u * (blockIdx.x % kCntIdleLines)], (uint32_t)__builtin_popcountll(idle_mask));
        }
        HitState h;
        if constexpr (CULL && (MODE == MODE_WAVE || MODE == MODE_CLUSTER || MODE == MODE_LANE)) {
            // one light per pass: the wave walks that light's own lists (wave-uniform record, scalar loads)
            DevOctree T = S.tree;
            // (the wave-together forms are always launched with one light per pass, group <= samples per light; the
            // per-lane form is when the scene has per-light forms)
            // (lights whose forms were merged share one set of arrays: then the lights of a hit may stay side by side)
            if (S.light_forms != nullptr && (S.lights_merged != 0u || gshift <= pshift)) {
                // (read through the constant address space: scalar loads, the pointers stay in scalar registers)
                const uint32_t l_form = S.lights_merged != 0u ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(q0 >> pshift));
                const RT_CONST LightForm& lf = *(const RT_CONST LightForm*)(uintptr_t)(S.light_forms + l_form);
                if (MODE == MODE_LANE && lf.lane_nodes != nullptr) {
                    T.nodes = lf.lane_nodes;
                    T.slot_geom = lf.lane_slot_geom;
                    T.root_info1 = lf.lane_root_info1;
                }
                if (MODE == MODE_WAVE && T.n_flat_tris != 0 && lf.flat_tris != nullptr) {
                    T.flat_tris = lf.flat_tris;
                    T.n_flat_tris = lf.n_flat_tris;
                    // a wave whose hits lie on one triangle: that triangle's (shorter) list
                    if ((opts & kOptHitLists) != 0u && lf.hit_lists != nullptr) {
                        const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)tri, __builtin_ctzll(on_mask));
                        if (ballot64(on && tri != h0) == 0) {
                            const RT_CONST uint32_t* e = (const RT_CONST uint32_t*)(uintptr_t)lf.hit_lists + 2u * h0;
                            T.flat_tris = lf.flat_tris + e[0];
                            T.n_flat_tris = e[1];
                        }
                    }
                }
                if (MODE == MODE_CLUSTER && lf.cl_nodes != nullptr) {
                    T.cl_nodes = lf.cl_nodes;
                    T.cl_tris = lf.cl_tris;
                }
            }
            if constexpr (MODE == MODE_LANE && SIDE) {
                // all lights of a hit side by side (gshift > pshift): every lane walks the tree of its own light
                uint32_t rec_base = 0, slot_base = 0;
                bool walk = on;
                if (S.light_forms != nullptr && gshift > pshift && S.light_forms[0].lane_nodes != nullptr) {
                    const LightForm& lf = S.light_forms[lit ? l : 0u];
                    T.nodes = S.light_forms[0].lane_nodes;
                    T.slot_geom = S.light_forms[0].lane_slot_geom;
                    rec_base = lf.lane_rec_base;
                    slot_base = lf.lane_slot_base;
                    walk = on && lf.lane_root_info1 != 0u;
                }
                trace<true, CULL, MODE, FAST, true>(T, lds[threadIdx.x >> 6], opts, walk, P, d_l, dist, h, rec_base, slot_base);
            } else {
                trace<true, CULL, MODE, FAST>(T, lds[threadIdx.x >> 6], opts, on, P, d_l, dist, h);
            }
        } else {
            trace<true, CULL, MODE, FAST>(S.tree, lds[threadIdx.x >> 6], opts, on, P, d_l, dist, h);
        }
        const lanemask unshadowed = on_mask & ~h.found & ~idle_mask;
        const uint32_t word = smp >> 5;
        if (lit && writer && word < words)
            vis[((size_t)l * words + word) * n_work + j] = (uint32_t)(unshadowed >> word_first_lane) & word_mask;
    }
}

// The instrumented passes of a level (rt_stats.h): the same rays k_closest / k_shadow trace, one ray per thread.
template <bool CULL>
__global__ void __launch_bounds__(256) k_stats_closest(SceneDev S, FrameParams F, uint32_t level, const RayRec* __restrict__ rays, uint32_t n,
                                                       const uint32_t* __restrict__ n_dev, unsigned long long* __restrict__ totals) {
    if (n_dev) n = *n_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid;
    f3 o, d;
    uint32_t pixel, node_id;
    load_ray(F, rays, level, i, n, valid, o, d, pixel, node_id);
    RayCounters c;
    stats_ray<false, CULL>(S.tree, valid, o, d, 0.0f, c);
    stats_flush(c, totals);
}

// Thread g handles shadow ray q = g % per_hit of hit j = g / per_hit, q = light * spt + sample.
template <bool CULL>
__global__ void __launch_bounds__(256) k_stats_shadow(SceneDev S, FrameParams F, uint32_t level, const RayRec* __restrict__ rays, uint32_t n,
                                                      const float* __restrict__ hit_t, const int32_t* __restrict__ hit_tri,
                                                      const uint32_t* __restrict__ work, const uint32_t* __restrict__ counters,
                                                      unsigned long long* __restrict__ totals) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_hit = S.n_lights * (uint32_t)F.spt;
    const uint32_t n_work = counters[CNT_WORK];
    const uint64_t j64 = g / per_hit;
    const bool active = j64 < n_work;
    const uint32_t q = (uint32_t)(g % per_hit);
    const uint32_t l = q / (uint32_t)F.spt, smp = q % (uint32_t)F.spt;
    uint32_t i = 0;
    if (active) i = work[j64];
    bool valid;
    f3 o, d;
    uint32_t pixel, node_id;
    load_ray(F, rays, level, i, active ? n : 0, valid, o, d, pixel, node_id);
    valid = valid && active;
    float t = 0.0f;
    int32_t light_eq = -1;
    if (valid) {
        t = hit_t[i];
        light_eq = S.tri_shade[hit_tri[i]].light_eq;
    }
    const f3 P = o + (d * t);  // renderer.cc:146
    const uint32_t pre = rng_prefix(F.seed, pixel, F.spp_sample, node_id);
    // renderer.cc:212-214: the light loop ends at the light the hit triangle itself is
    const uint32_t lights_traced = !valid ? 0u : ((light_eq >= 0 && (uint32_t)light_eq < S.n_lights) ? (uint32_t)light_eq : S.n_lights);
    const bool on = l < lights_traced;
    const LightDev& L = S.lights[on ? l : 0u];
    const f3 p_l = light_sample(L, pre, l, smp);
    const f3 to_l = p_l - P;  // renderer.cc:289-290
    const float dist = length(to_l);
    const f3 d_l = normalize(to_l);
    RayCounters c;
    stats_ray<true, CULL>(S.tree, on, P, d_l, dist, c);
    stats_flush<true>(c, totals);
}

// pow(x, y) of the specular terms (renderer.cc:263,283) for x >= 0 (a clamped cosine) or NaN: exp2(y * log2(x)) on the
// transcendental unit -- 4 instructions where the library's correctly rounded powf is some 80.  The device's powf is not
// the reference's libm either; shaded colours are compared within 1/255, and this form stays within about
// |y| * 2^-21 relative of the exact power.  The special cases that the formula would get wrong are pow(x, 0) = 1 for
// every x (0 * -inf and NaN included) and pow(1, y) = 1; pow(0, y > 0) = 0 and pow(NaN, y != 0) = NaN come out of it.
__device__ __forceinline__ float shade_pow(float x, float y) {
    const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    return (y == 0.0f || x == 1.0f) ? 1.0f : r;
}

// What ray_color (renderer.cc:144-192) makes of one hit: the clamped local colour, the material's factors for the
// children's colours, and the secondary rays the hit spawns.
struct ShadeResult {
    f3 col;                // clamped local colour (renderer.cc:170)
    f3 fR, fT;             // material.specular / material.transmittance
    uint32_t mat;          // the hit triangle's material
    f3 P;                  // the hit point (origin of the secondary rays)
    f3 dir_R, dir_T;
    bool want_R, want_T;
};

// Local illumination + recursion gate + secondary ray construction for one hit per lane (renderer.cc:144-192, 202-335).
// valid: the lane holds a hit (ray o, d; distance t; triangle tri).  vis_word(l, w): the visibility bits of samples
// 32 w .. 32 w + 31 of light l for this hit (bit = 1: not shadowed).  Every lane of the wave must call it (one wave-level
// test decides whether the specular power is needed at all).
template <class VisWord>
__device__ __forceinline__ void shade_hit(const SceneDev& S, const FrameParams& F, uint32_t level, bool valid, f3 o, f3 d, float t, uint32_t tri,
                                          uint32_t pixel, uint32_t node_id, VisWord vis_word, ShadeResult& r) {
    const TriShade ts = S.tri_shade[valid ? tri : 0u];
    const MatDev m = S.mats[valid ? ts.mat : 0];
    const f3 P = o + (d * t);  // renderer.cc:146
    f3 N = mk(ts.nx, ts.ny, ts.nz);
    if (F.shading_type == 1 && valid) N = interpolate_normal(S.tri_geom[tri], S.tri_vnorm + (size_t)tri * 9, P);
    const f3 diffuse = mk(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    const f3 specular = mk(m.specular[0], m.specular[1], m.specular[2]);

    // The specular term is (I_l * specular) * pow(base, shininess).  With an all-zero specular colour and a
    // non-negative shininess the power is finite, so the product is the same zero (or the NaN that I_l already is)
    // whatever the power's value: a wave none of whose lanes has a specular material skips the power and the two
    // normalisations that only feed it.
    const bool spec_is_zero = specular.x == 0.0f && specular.y == 0.0f && specular.z == 0.0f && m.shininess >= 0.0f;
    const bool wave_needs_pow = ballot64(valid && !spec_is_zero) != 0;

    f3 col = mk(0, 0, 0);
    if (F.render_type == 0) {  // renderer.cc:197-200
        const float g = 1.0f - (std_min(t, 1.5f * F.max_distance) / (1.5f * F.max_distance));
        col = col + mk(g, g, g);
    } else if (F.render_type == 1) {
        col = col + diffuse;
    } else if (valid) {  // renderer.cc:202-247
        const uint32_t pre = rng_prefix(F.seed, pixel, F.spp_sample, node_id);
        f3 final_c = mk(0, 0, 0);
        // the direction to the viewer (renderer.cc:257,277) depends on the hit alone: once per hit, not once per sample
        // (the guarded division in normalize() tests its operands wave-wide, which keeps the compiler from moving it)
        f3 d_v = mk(0, 0, 0);
        if (wave_needs_pow) d_v = normalize(F.cam_pos - P);
        for (uint32_t l = 0; l < S.n_lights; ++l) {
            const LightDev L = S.lights[l];
            const f3 emission = mk(L.emission[0], L.emission[1], L.emission[2]);
            if (ts.light_eq == (int32_t)l) {  // renderer.cc:212-214
                final_c = emission;
                break;
            }
            const f3 n_l = mk(L.n[0], L.n[1], L.n[2]);
            f3 acc = mk(0, 0, 0);
            // Every lane steps through ITS visible samples (ascending, so the sum keeps the reference's order): the wave
            // takes as many turns as its lane with the most visible samples, not one per sample that any lane sees.
            for (uint32_t wd = 0; wd * 32u < (uint32_t)F.spt; ++wd) {
              uint32_t bits = vis_word(l, wd);
              const uint32_t left = (uint32_t)F.spt - wd * 32u;
              if (left < 32u) bits &= (1u << left) - 1u;
              while (bits != 0) {
                const uint32_t s = wd * 32u + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                const f3 p_l = light_sample(L, pre, l, s);
                // renderer.cc:220-240
                const f3 to_l = p_l - P;
                const float dist = length(to_l);
                const f3 d_l = normalize(to_l);
                const float w = div_scalar(L.area * std_max(0.0f, dot(n_l, -d_l)), (float)F.spt * dist * dist + kEps, true);
                const f3 I_l = emission * w;
                const f3 I_a = I_l * mk(0, 0, 0);
                const f3 I_d = (I_l * diffuse) * std_max(0.0f, dot(N, d_l));
                float pw = 1.0f;  // the power of the specular term; irrelevant on lanes whose specular colour is zero
                if (wave_needs_pow) {
                    if (F.render_type == 2) {  // renderer.cc:256-263
                        const f3 d_r = ((N * 2.0f) * dot(N, d_l)) - d_l;
                        pw = shade_pow(std_max(0.0f, dot(d_v, d_r)), m.shininess);
                    } else {  // renderer.cc:276-283
                        const f3 d_h = normalize(d_l + d_v);
                        pw = shade_pow(std_max(0.0f, dot(N, d_h)), m.shininess);
                    }
                }
                const f3 I_s = (I_l * specular) * pw;
                f3 c = (I_a + I_d) + I_s;
                if (F.render_type == 2) c = c + mk(m.emission[0], m.emission[1], m.emission[2]);  // renderer.cc:264-266
                acc = acc + c;
              }
            }
            final_c = final_c + acc;
        }
        col = col + final_c;
    }
    col = clamp_color(col);  // renderer.cc:170

    // renderer.cc:172-192
    bool want_R = false, want_T = false;
    f3 dir_R = mk(0, 0, 0), dir_T = mk(0, 0, 0);
    if (valid && !eq_eps(col, mk(1.0f, 1.0f, 1.0f)) && (int)level < F.max_depth) {
        const f3 d_v = -d;
        if (m.has_specular) {  // renderer.cc:315-321
            const f3 d_r = ((N * 2.0f) * dot(N, d_v)) - d_v;
            dir_R = normalize(d_r);
            want_R = true;
        }
        if (m.has_transmittance) {  // renderer.cc:323-335; the sqrt and the subtraction run in double there
            const float nn = 1.0f / m.ior;
            const float ndotv = dot(N, d_v);
            const double root = sqrt((double)(1.0f - nn * nn * (1.0f - ndotv * ndotv)));
            const float coef = (float)((double)(nn * ndotv) - root);
            const f3 tt = (d_v * -nn) + (N * coef);
            if (length(tt) > kEps) dir_T = normalize(tt);
            want_T = !eq_eps(dir_T, mk(0, 0, 0));  // renderer.cc:187
        }
    }
    r.col = col;
    r.fR = specular;
    r.fT = mk(m.transmittance[0], m.transmittance[1], m.transmittance[2]);
    r.mat = valid ? (uint32_t)ts.mat : 0u;
    r.P = P;
    r.dir_R = dir_R;
    r.dir_T = dir_T;
    r.want_R = want_R;
    r.want_T = want_T;
}

// shade_hit for every hit of one level; the visibility of each light sample comes from k_shadow.
__global__ void __launch_bounds__(256) k_shade(SceneDev S, FrameParams F, uint32_t level, const RayRec* __restrict__ rays, uint32_t n,
                                               const float* __restrict__ hit_t, const int32_t* __restrict__ hit_tri,
                                               const uint32_t* __restrict__ work, const uint32_t* __restrict__ vis, uint32_t words,
                                               NodeOut* __restrict__ out, RayRec* __restrict__ next, uint32_t next_cap,
                                               uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_work = counters[CNT_WORK];
    const bool active = j < n_work;
    uint32_t i = 0;
    if (active) i = work[j];
    bool valid;
    f3 o, d;
    uint32_t pixel, node_id;
    load_ray(F, rays, level, i, active ? n : 0, valid, o, d, pixel, node_id);
    valid = valid && active;
    float t = 0.0f;
    uint32_t tri = 0;
    if (valid) {
        t = hit_t[i];
        tri = (uint32_t)hit_tri[i];
    }
    ShadeResult r;
    shade_hit(S, F, level, valid, o, d, t, tri, pixel, node_id,
              [&](uint32_t l, uint32_t w) { return vis[((size_t)l * words + w) * n_work + j]; }, r);
    if (valid) {
        NodeOut q;
        q.c[0] = r.col.x; q.c[1] = r.col.y; q.c[2] = r.col.z;
        q.meta = (r.want_R ? 1u : 0u) | (r.want_T ? 2u : 0u) | (r.mat << 2);
        out[i] = q;
    }
    const uint32_t sR = wave_append(&counters[CNT_NEXT], r.want_R);
    const uint32_t sT = wave_append(&counters[CNT_NEXT], r.want_T);
    if (r.want_R && sR < next_cap) {
        RayRec q;
        q.o[0] = r.P.x; q.o[1] = r.P.y; q.o[2] = r.P.z;
        q.d[0] = r.dir_R.x; q.d[1] = r.dir_R.y; q.d[2] = r.dir_R.z;
        q.parent = i; q.pixel = pixel; q.node_id = node_id * 2;
        q.pad[0] = q.pad[1] = q.pad[2] = 0;
        next[sR] = q;
    }
    if (r.want_T && sT < next_cap) {
        RayRec q;
        q.o[0] = r.P.x; q.o[1] = r.P.y; q.o[2] = r.P.z;
        q.d[0] = r.dir_T.x; q.d[1] = r.dir_T.y; q.d[2] = r.dir_T.z;
        q.parent = i; q.pixel = pixel; q.node_id = node_id * 2 + 1;
        q.pad[0] = q.pad[1] = q.pad[2] = 0;
        next[sT] = q;
    }
}

// The rest of the recursion in ONE launch, for scenes whose materials spawn at most one secondary ray per hit (reflect
// or refract, never both): from level `first` on every path is a chain, and a lane follows its chain to the end --
// closest hit, the hit's shadow rays, shading, the next ray -- instead of the host launching three kernels per level.
// Deep levels are small (tens of thousands of rays) and each of their launches lasts as long as one wave's chain of
// dependent steps plus a launch boundary; here the boundaries are gone and nothing waits for the slowest wave of a level.
// A path owns 2^gshift neighbouring lanes: the first carries the path (its rays, its closest hits, its shading), and after
// a hit lane `sub` traces shadow ray q = sub = light << pshift | sample (all lights of a hit side by side; the host uses
// this kernel only if they fit 64 lanes).  The walks are the per-lane form's (work stealing included), the arithmetic is
// that of k_closest / k_shadow / k_shade, so a frame has the same bits whichever way its levels are run.
// Values: ray_color returns c + I_child * factor (renderer.cc:181-182,188-189); a path stores (c, factor) per level in
// `stack` ([level - first][path][6]) and folds them from the deepest level up, then hands the result to its parent node
// of level first - 1 like k_resolve does.  counters: the block of level `first` (CNT_NEXT of the block before it = the
// number of paths); the secondary and shadow rays of all fused levels are added to it.
// (118 vector registers, 4 waves per SIMD; forcing 5 to 7 waves with spills measured the same)
template <bool CULL, bool FAST>
__global__ void __launch_bounds__(64) k_tail(SceneDev S, FrameParams F, uint32_t first, const RayRec* __restrict__ rays, uint32_t n_bound,
                                             float* __restrict__ parent_child, float* __restrict__ stack, uint32_t pshift, uint32_t gshift,
                                             uint32_t* __restrict__ counters, uint32_t opts) {
    __shared__ TraverseLds<MODE_LANE> lds[1];
    const uint32_t n = min(counters[CNT_NEXT - CNT_WORDS], n_bound);
    const uint32_t group = 1u << gshift;
    bool lane_ok;
    const uint64_t slot = ray_index((uint64_t)xcd_block() * blockDim.x + threadIdx.x, opts_sparse(opts), lane_ok);
    const uint32_t j = (uint32_t)(slot >> gshift), sub = (uint32_t)slot & (group - 1u);
    const uint32_t lane = __lane_id(), lead = lane - sub;  // (a group never straddles waves: group <= 64 >> sparse)
    const bool carrier = lane_ok && sub == 0u && j < n;
    const lanemask group_mask = group >= 64u ? ~0ull : ((1ull << group) - 1ull);
    const bool lit = F.render_type >= 2 && F.spt > 0 && S.n_lights > 0;
    const uint32_t per_light = 1u << pshift;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    uint32_t pixel = 0, node_id = 1, parent = 0;
    if (carrier) {
        const RayRec r = rays[j];
        o = mk(r.o[0], r.o[1], r.o[2]);
        d = mk(r.d[0], r.d[1], r.d[2]);
        pixel = r.pixel;
        node_id = r.node_id;
        parent = r.parent;
    }
    const uint32_t first_node_id = node_id;
    bool live = carrier;
    uint32_t depth_n = 0, n_next = 0, n_idle = 0;
    unsigned long long n_shadow = 0;
    for (uint32_t level = first;; ++level) {
        // ---- closest hit (renderer.cc:116-142)
        HitState h;
        trace<false, CULL, MODE_LANE>(S.tree, lds[0], opts, live, o, d, 0.0f, h);
        const bool hit = live && !(h.t == RT_INF);
        const f3 P = o + (d * h.t);  // renderer.cc:146
        const uint32_t pre = rng_prefix(F.seed, pixel, F.spp_sample, node_id);
        uint32_t lights_traced = 0, dark = 0;
        f3 N = mk(0, 0, 0);
        if (hit && lit) {  // renderer.cc:212-214: the light loop ends at the light the hit triangle itself is
            const TriShade ts = S.tri_shade[h.tri];
            lights_traced = (ts.light_eq >= 0 && (uint32_t)ts.light_eq < S.n_lights) ? (uint32_t)ts.light_eq : S.n_lights;
            dark = ts.dark;
            N = mk(ts.nx, ts.ny, ts.nz);
        }
        // ---- the hit's shadow rays (renderer.cc:217-219, 288-313), one per lane of the group
        lanemask vis_bits = 0;
        if (lit && ballot64(lights_traced != 0u) != 0) {
            const f3 gP = mk(__shfl(P.x, (int)lead, 64), __shfl(P.y, (int)lead, 64), __shfl(P.z, (int)lead, 64));
            const uint32_t g_pre = __shfl(pre, (int)lead, 64), g_traced = __shfl(lights_traced, (int)lead, 64);
            const uint32_t g_dark = __shfl(dark, (int)lead, 64);
            const f3 gN = mk(__shfl(N.x, (int)lead, 64), __shfl(N.y, (int)lead, 64), __shfl(N.z, (int)lead, 64));
            const uint32_t l = sub >> pshift, smp = sub & (per_light - 1u);
            const bool on = lane_ok && l < g_traced && smp < (uint32_t)F.spt;
            const LightDev& L = S.lights[on ? l : 0u];
            const f3 p_l = light_sample(L, g_pre, l, smp);
            const f3 to_l = p_l - gP;  // renderer.cc:289-290
            const float dist = length(to_l);
            const f3 d_l = normalize(to_l);
            HitState hs;
            // every lane walks the tree of its own light where the scene has per-light trees (LightForm)
            DevOctree T = S.tree;
            uint32_t rec_base = 0, slot_base = 0;
            // (samples that add nothing to the hit's colour whether visible or not are not traced: shadow_ray_idle)
            const bool idle = on && shadow_ray_idle(F, g_dark, gN, mk(L.n[0], L.n[1], L.n[2]), d_l, dist);
            n_idle += idle ? 1u : 0u;
            bool walk = on && !idle;
            if (CULL && S.light_forms != nullptr && S.light_forms[0].lane_nodes != nullptr) {
                const LightForm& lf = S.light_forms[on ? l : 0u];
                T.nodes = S.light_forms[0].lane_nodes;
                T.slot_geom = S.light_forms[0].lane_slot_geom;
                rec_base = lf.lane_rec_base;
                slot_base = lf.lane_slot_base;
                walk = walk && lf.lane_root_info1 != 0u;  // (0: nothing can occlude that light)
            }
            trace<true, CULL, MODE_LANE, FAST, true>(T, lds[0], opts, walk, gP, d_l, dist, hs, rec_base, slot_base);
            const lanemask unshadowed = ballot64(on && !idle) & ~hs.found;
            vis_bits = (unshadowed >> lead) & group_mask;
        }
        // ---- shading, recursion gate, the next ray (renderer.cc:144-192)
        ShadeResult r;
        shade_hit(S, F, level, hit, o, d, h.t, h.tri, pixel, node_id,
                  [&](uint32_t l, uint32_t w) {
                      const uint32_t bits = (uint32_t)(vis_bits >> ((l << pshift) + 32u * w));
                      return per_light >= 32u ? bits : (bits & ((1u << per_light) - 1u));
                  },
                  r);
        if (live) {
            const bool child = r.want_R || r.want_T;
            const f3 c = hit ? r.col : F.bg;  // renderer.cc:140-142: a miss is the background colour
            const f3 f = r.want_R ? r.fR : r.fT;
            float* e = stack + ((size_t)depth_n * n_bound + j) * 6;
            e[0] = c.x; e[1] = c.y; e[2] = c.z;
            e[3] = child ? f.x : 0.0f; e[4] = child ? f.y : 0.0f; e[5] = child ? f.z : 0.0f;
            ++depth_n;
            n_shadow += (unsigned long long)lights_traced * (unsigned long long)(uint32_t)F.spt;
            if (child) {
                ++n_next;
                o = r.P;
                d = r.want_R ? r.dir_R : r.dir_T;
                node_id = node_id * 2u + (r.want_R ? 0u : 1u);
            }
            live = child;
        }
        if (ballot64(live) == 0) break;
    }
    if (carrier) {
        // from the deepest node up: value = c + child value * factor (node_value)
        const float* e = stack + ((size_t)(depth_n - 1u) * n_bound + j) * 6;
        f3 v = mk(e[0], e[1], e[2]);
        for (uint32_t k = depth_n - 1u; k-- > 0u;) {
            e = stack + ((size_t)k * n_bound + j) * 6;
            v = mk(e[0], e[1], e[2]) + (v * mk(e[3], e[4], e[5]));
        }
        float* dst = parent_child + (size_t)parent * kNodeChildFloats + ((first_node_id & 1u) ? 3 : 0);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
    }
    // the rays of the fused levels, for the frame's totals
    uint32_t deepest = carrier ? first + depth_n : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        n_next += __shfl_xor(n_next, s, 64);
        n_idle += __shfl_xor(n_idle, s, 64);
        n_shadow += __shfl_xor(n_shadow, s, 64);
        const uint32_t other = __shfl_xor(deepest, s, 64);
        deepest = deepest > other ? deepest : other;
    }
    if (lane == 0) {
        if (n_next) atomicAdd(&counters[CNT_NEXT], n_next);
        if (n_shadow) atomicAdd((unsigned long long*)&counters[CNT_SHADOW_LO], n_shadow);
        if (n_idle) atomicAdd(&counters[CNT_IDLE + 32u * (blockIdx.x % kCntIdleLines)], n_idle);
        if (deepest) atomicMax(&counters[CNT_DEEPEST], deepest);
    }
}

// renderer.cc:181-182,188-189: final += I_R * specular; final += I_T * transmittance -- evaluated bottom-up,
// one level per launch, each node handing its value to its parent.
__device__ __forceinline__ f3 node_value(const NodeOut& r, const float* __restrict__ child, const MatDev* __restrict__ mats) {
    f3 v = mk(r.c[0], r.c[1], r.c[2]);
    if (r.meta & 3u) {
        const MatDev& m = mats[r.meta >> 2];
        if (r.meta & 1u) v = v + (mk(child[0], child[1], child[2]) * mk(m.specular[0], m.specular[1], m.specular[2]));
        if (r.meta & 2u) v = v + (mk(child[3], child[4], child[5]) * mk(m.transmittance[0], m.transmittance[1], m.transmittance[2]));
    }
    return v;
}

__global__ void __launch_bounds__(256) k_resolve(const NodeOut* __restrict__ cur, const float* __restrict__ cur_child, const RayRec* __restrict__ rays,
                                                 const uint32_t* __restrict__ n_dev, const MatDev* __restrict__ mats, float* __restrict__ parent_child) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_dev) return;
    const f3 v = node_value(cur[i], cur_child + (size_t)i * kNodeChildFloats, mats);
    const RayRec r = rays[i];
    float* dst = parent_child + (size_t)r.parent * kNodeChildFloats + ((r.node_id & 1u) ? 3 : 0);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
}

// renderer.cc:42-51: sum over the samples of a pixel, divide, clamp.
__global__ void __launch_bounds__(256) k_resolve_pixels(FrameParams F, const NodeOut* __restrict__ level0, const float* __restrict__ level0_child,
                                                        const MatDev* __restrict__ mats, uint32_t n, float* __restrict__ accum, float* __restrict__ out_rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y;
    if (!node0_pixel(F, i, x, y)) return;
    const size_t pix = (size_t)y * F.width + x;
    f3 v = node_value(level0[i], level0_child + (size_t)i * kNodeChildFloats, mats);
    if (F.spp_sample > 0) v = mk(accum[i * 3], accum[i * 3 + 1], accum[i * 3 + 2]) + v;
    if ((int)F.spp_sample + 1 < F.spp) {
        accum[i * 3] = v.x; accum[i * 3 + 1] = v.y; accum[i * 3 + 2] = v.z;
    } else {
        v = clamp_color(div_guard(v, (float)F.spp));
        out_rgb[pix * 3] = v.x; out_rgb[pix * 3 + 1] = v.y; out_rgb[pix * 3 + 2] = v.z;
    }
}

// End of one pixel sample: adds the sample's per-level counts to the frame totals (mirrored into pinned host memory for
// the host to read after the frame) and zeroes the per-level counters for the next sample.  One block.
__global__ void __launch_bounds__(256) k_roll_counters(uint32_t* __restrict__ counters, uint32_t levels, FrameTotals* __restrict__ totals,
                                                       FrameTotals* __restrict__ host_totals) {
    __shared__ unsigned long long s_next, s_shadow, s_idle;
    __shared__ uint32_t s_deepest;
    if (threadIdx.x == 0) {
        s_next = s_shadow = s_idle = 0ull;
        s_deepest = 0u;
    }
    __syncthreads();
    for (uint32_t L = threadIdx.x; L < levels; L += blockDim.x) {
        const uint32_t* c = counters + (size_t)L * CNT_WORDS;
        atomicAdd(&s_next, (unsigned long long)c[CNT_NEXT]);
        atomicAdd(&s_shadow, (unsigned long long)c[CNT_SHADOW_LO] | ((unsigned long long)c[CNT_SHADOW_HI] << 32));
        atomicMax(&s_deepest, c[CNT_DEEPEST]);
        for (uint32_t k = 0; k < kCntIdleLines; ++k) atomicAdd(&s_idle, (unsigned long long)c[CNT_IDLE + 32u * k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        FrameTotals t = *totals;
        t.secondary_rays += s_next;
        t.shadow_rays += s_shadow;
        t.idle_shadow_rays += s_idle;
        t.deepest_level = t.deepest_level > s_deepest ? t.deepest_level : (unsigned long long)s_deepest;
        *totals = t;
        *host_totals = t;
        __threadfence_system();
    }
    __syncthreads();
    for (uint32_t L = threadIdx.x; L < levels; L += blockDim.x) {
        uint32_t* c = counters + (size_t)L * CNT_WORDS;
        c[CNT_WORK] = c[CNT_NEXT] = c[CNT_SHADOW_LO] = c[CNT_SHADOW_HI] = c[CNT_SPAWN] = c[CNT_TICKET] = c[CNT_DEEPEST] = 0u;
        for (uint32_t k = 0; k < kCntIdleLines; ++k) c[CNT_IDLE + 32u * k] = 0u;
    }
}

// Start of a frame: clears the counters of every level and the frame totals, and tells each level where the host
// listens (slots + L, pinned host memory; nullptr = nowhere) and from which sequence number its launches count.
__global__ void __launch_bounds__(256) k_frame_begin(uint32_t* __restrict__ counters, uint32_t levels, LevelSlot* slots, uint32_t seq_base,
                                                     FrameTotals* __restrict__ totals) {
    for (uint32_t k = threadIdx.x; k < levels * CNT_WORDS; k += blockDim.x) counters[k] = 0u;
    __syncthreads();
    for (uint32_t L = threadIdx.x; L < levels; L += blockDim.x) {
        uint32_t* c = counters + (size_t)L * CNT_WORDS;
        const uintptr_t p = slots ? (uintptr_t)(slots + L) : (uintptr_t)0;
        c[CNT_SEQ] = seq_base;
        c[CNT_SLOT_LO] = (uint32_t)p;
        c[CNT_SLOT_HI] = (uint32_t)((uint64_t)p >> 32);
    }
    if (threadIdx.x == 0 && totals) totals->secondary_rays = totals->shadow_rays = totals->deepest_level = totals->idle_shadow_rays = 0ull;
}

__global__ void __launch_bounds__(256) k_primary_hits(FrameParams F, const float* __restrict__ hit_t, const int32_t* __restrict__ hit_tri,
                                                      uint32_t n, float* __restrict__ out_t, int32_t* __restrict__ out_tri) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y;
    if (!node0_pixel(F, i, x, y)) return;
    const size_t pix = (size_t)y * F.width + x;
    out_t[pix] = hit_t[i];
    out_tri[pix] = hit_tri[i];
}

// ---- Ads::Search restated (octree.cc:173-207): breadth-first, one ray per thread, queue in a workspace ---------------
//
// FILL = false: counts[i] = the length of ray i's candidate list (octree.cc:204).  FILL = true: the list itself, written
// to out_tri from position offsets[i] (stride == 0) or i * stride (stride != 0: every ray owns a region of `stride`
// entries and counts[i] is written as well, so that one launch serves callers that accept a retry when a list is
// longer than its region).  Entries at or beyond `limit` (stride == 0) / beyond the ray's region are not written.
// `stats`, if not null, receives the sums over the launch of nodes taken from the queue and of leaves among them
// (search_node_count_ and search_leaves_visited_, octree.cc:185-187), one 64-bit atomic each per wave.

template <bool FILL>
__global__ void __launch_bounds__(64) k_search(DevOctree T, const float* __restrict__ rays, uint32_t n, uint32_t* __restrict__ queue,
                                               uint32_t queue_cap, uint64_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
                                               uint32_t* __restrict__ out_tri, uint64_t limit, uint64_t stride,
                                               unsigned long long* __restrict__ stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nodes_taken = 0, leaves_taken = 0;
    if (i < n) {
        const f3 o = mk(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
        const f3 d = mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        const f3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        uint32_t* q = queue + (size_t)i * queue_cap * 2;
        uint64_t total = 0;
        uint64_t w = 0, lim = 0;
        if (FILL) {
            w = stride ? (uint64_t)i * stride : offsets[i];
            lim = stride ? w + stride : limit;
        }
        uint32_t qh = 0, qt = 0;
        if (T.root_info1 != 0) {
            q[0] = T.root_info0;
            q[1] = T.root_info1;
            qt = 1;
        }
        while (qh < qt && qh < queue_cap) {
            const uint32_t i0 = q[qh * 2], i1 = q[qh * 2 + 1];
            ++qh;
            ++nodes_taken;
            if (!(i1 & kChildInternal)) {
                ++leaves_taken;
                if (FILL)
                    for (uint32_t k = 0; k < i1; ++k)
                        if (w + k < lim) out_tri[w + k] = T.slots[i0 + k].tri;
                w += i1;
                total += i1;
                continue;
            }
            const NodeRec& r = T.nodes[i0];
            for (int c = 0; c < 8; ++c) {
                if (r.info[c][1] == 0) continue;
                const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
                if (!ray_box(o, inv, r.lo[0][bx], r.lo[1][by], r.lo[2][bz], r.hi[0][bx], r.hi[1][by], r.hi[2][bz])) continue;
                if (qt < queue_cap) {
                    q[qt * 2] = r.info[c][0];
                    q[qt * 2 + 1] = r.info[c][1];
                }
                ++qt;
            }
        }
        if (!FILL || stride) counts[i] = (qt > queue_cap) ? ~0ull : total;
    }
    if (stats) {
        uint32_t a = nodes_taken, b = leaves_taken;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            a += __shfl_xor(a, s, 64);
            b += __shfl_xor(b, s, 64);
        }
        if (__lane_id() == 0 && a) {
            atomicAdd(&stats[0], (unsigned long long)a);
            atomicAdd(&stats[1], (unsigned long long)b);
        }
    }
}

// Exclusive prefix sum of n <= 256 * 16 candidate counts: offsets[0..n] and *total (one block of 256 threads).  Used by
// the entry point that allocates nothing, where the whole Search batch stays on the device.
constexpr uint32_t kSearchScanMax = 256 * 16;
__global__ void __launch_bounds__(256) k_search_offsets(const uint64_t* __restrict__ counts, uint32_t n, uint64_t* __restrict__ offsets,
                                                        uint64_t* __restrict__ total) {
    __shared__ uint64_t s_sum[256];
    const uint32_t t = threadIdx.x, first = t * 16u;
    uint64_t local = 0;
    for (uint32_t k = 0; k < 16u; ++k)
        if (first + k < n) local += counts[first + k];
    s_sum[t] = local;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint64_t v = t >= d ? s_sum[t - d] : 0ull;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint64_t run = s_sum[t] - local;
    for (uint32_t k = 0; k < 16u; ++k)
        if (first + k < n) {
            offsets[first + k] = run;
            run += counts[first + k];
        }
    if (t == 255) {
        offsets[n] = s_sum[255];
        *total = s_sum[255];
    }
}

}  // namespace rt
