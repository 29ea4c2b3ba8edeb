// rt_traverse.h -- wave64 octree traversal + ray/triangle test for gfx950 (device only).
//
// One ray per lane, 64 rays per wavefront, traversed TOGETHER: the wave walks the union of the
// nodes its rays reach, and each stack entry carries the 64-bit mask of the lanes that reached
// it.  Because the node being worked on is the same for every lane, node records and triangle
// records are fetched with scalar loads (s_load through the constant cache) and feed the VALU as
// SGPR operands; no vector memory instruction and no LDS is needed inside the loop.  The stack
// itself lives in registers: entry i of the wave-uniform stack is stored in lane i of a VGPR
// (written with a lane-select, read with v_readlane), two VGPRs per field for up to 128 entries.
//
// Lane predicates are kept as explicit 64-bit masks in scalar registers: every float comparison is
// turned into a mask by a ballot (which is just the v_cmp result), masks are combined with scalar
// logic, and the only branches are wave-uniform "no lane is left" tests.  All lanes execute every
// arithmetic stage; lanes that are not in the mask compute garbage that is never selected.
//
// The set of (ray, triangle) tests performed is exactly the candidate set of the reference's
// Octree::Search (src/ADS/Octree/octree.cc:173-207 of the reference): a triangle reference is
// tested iff the ray passes the slab test of every box on the path from the root's children down
// to its leaf; the root box is never tested; empty octants are skipped.  The traversal order is
// depth-first instead of the reference's breadth-first; the closest-hit reduction therefore keys
// ties on the global leaf-slot index, which is the reference's candidate-list order (see
// host_octree.h).  For any-hit queries a lane retires as soon as it has found an occluder (the
// result is a boolean, src/Renderer/renderer.cc:302-306).
#pragma once

#include <hip/hip_runtime.h>

#include "host_octree.h"
#include "rt_math.h"

namespace rt {

#define RT_CONST __attribute__((address_space(4)))

struct DevOctree {
    const NodeRec* nodes;
    const LeafTri* slots;
    uint32_t root_info0, root_info1;
    uint32_t n_nodes, n_slots;
    // Small-tree form (n_flat_tris > 0): at most 32 leaves.  flat_nodes = nodes with each leaf octant's info[0]
    // replaced by the leaf's ordinal (breadth-first leaf order); flat_tris[j] = triangle j's geometry with the
    // set of leaves that list it as a bit mask in pad0.
    const NodeRec* flat_nodes;
    const LeafTri* flat_tris;
    uint32_t n_flat_tris;
    // Clustered form (cl_nodes != nullptr; any-hit walks of mid-size trees): see ClusterForm in host_octree.h.
    const NodeRec* cl_nodes;
    const LeafTri* cl_tris;
    uint32_t cl_root_info1;
    // The geometry of the leaf slots alone, 9 floats (a, e1, e2) per slot: what the per-lane and task-pool forms read.
    // Their pair tests fetch every record from beyond the L2 (a big tree's slots are read with next to no reuse there),
    // so the kernel's time follows the bytes per record: 36 instead of 48.
    const float* slot_geom;
};

typedef const RT_CONST float* cfloat_p;
typedef const RT_CONST uint32_t* cu32_p;
typedef uint64_t lanemask;

__device__ __forceinline__ cfloat_p as_const_f(const void* p) { return (cfloat_p)(uintptr_t)p; }
__device__ __forceinline__ cu32_p as_const_u(const void* p) { return (cu32_p)(uintptr_t)p; }

__device__ __forceinline__ lanemask ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_of(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// Writes the wave-uniform value v into lane `idx` of the per-lane register `old` (a v_writelane; this
// compiler exposes no builtin for it, and a compare + select is what it lowers to anyway).
__device__ __forceinline__ int write_lane(int v, int idx, int old) { return ((int)__lane_id() == idx) ? v : old; }

// Moeller-Trumbore exactly as src/collision_detection.h:12-68: same operation order, the same
// comparisons (so NaNs fall through the same way), 1/d as a correctly rounded division.
// Returns the lanes of `mine` whose ray hits the triangle with a usable distance, i.e. that pass
// the reference's `t == infinity || t < epsilon` filter too (renderer.cc:126 / :298); t is valid for them.
template <bool CULL>
__device__ __forceinline__ lanemask ray_triangle(lanemask mine, f3 o, f3 s, f3 a, f3 e1, f3 e2, float& t) {
    const f3 p = cross(s, e2);
    const float d = dot(e1, p);
    lanemask ok = mine & ~ballot64(CULL ? (d < kEps) : (fabsf(d) < kEps));
    if (ok == 0) return 0;
    // 1 / d: the lanes of `ok` have |d| >= kEps; if none of them exceeds kRcpFastMax either, the refinement chain of
    // rt_math.h gives the bits of the division at half its cost (the other lanes' quotients are never used)
    float d_inv;
    if ((ok & ballot64(fabsf(d) > kRcpFastMax)) == 0) d_inv = rcp_refined(d);
    else d_inv = 1.0f / d;
    const f3 q = o - a;
    const float u = d_inv * dot(q, p);
    ok &= ~(ballot64(u < 0.0f) | ballot64(u > 1.0f));
    if (ok == 0) return 0;
    const f3 w = cross(q, e1);
    const float v = d_inv * dot(w, s);
    ok &= ~(ballot64(v < 0.0f) | ballot64((u + v) > 1.0f));
    if (ok == 0) return 0;
    t = d_inv * dot(e2, w);
    ok &= ~(ballot64(t == RT_INF) | ballot64(t < kEps));
    return ok;
}

// The any-hit form of the same test, written in ISA: found |= lanes of `need` whose ray hits the triangle at a usable
// distance below t_limit (renderer.cc:298-302).  The compiled form above keeps every lane predicate in a scalar mask and
// pays four or five scalar instructions per stage for it (and-not, compare with zero, branch, ...); on gfx950 a scalar
// instruction costs a SIMD 1.7 times the issue time of a vector one (measured, tools/salu_rate.hip), and the wave-together
// any-hit kernels were bound by exactly that.  Here the predicate IS the EXEC mask: v_cmpx narrows it stage by stage
// for free and one s_cbranch_execz per stage leaves when no lane is left -- 8 scalar instructions per test instead of 22.
// Same operations in the same order as ray_triangle<> (no fused multiply-add outside the reciprocal's refinement
// chain, which is rcp_refined's), same comparisons (a NaN falls through the same way).
// Only for callers that guarantee |d| <= kRcpFastMax for every lane that passes the first stage: unit-length (or zero,
// or NaN) ray directions and a scene whose triangles have |e1| * |e2| below that bound -- then 1 / d never needs the
// full division.  rec: a, e1, e2 in wave-uniform registers.
// The stages of the test, entered with EXEC = the lanes to test; leaves through label 1 with EXEC = the lanes that found
// an occluder already or-ed into %[found].
#define RT_ANY_HIT_STAGES(FIRST_STAGE) \
    /* p = cross(s, e2) */ \
    "v_mul_f32 %[p0], %[e2z], %[sy]\n" \
    "v_mul_f32 %[t0], %[e2y], %[sz]\n" \
    "v_sub_f32 %[p0], %[p0], %[t0]\n" \
    "v_mul_f32 %[p1], %[e2x], %[sz]\n" \
    "v_mul_f32 %[t0], %[e2z], %[sx]\n" \
    "v_sub_f32 %[p1], %[p1], %[t0]\n" \
    "v_mul_f32 %[p2], %[e2y], %[sx]\n" \
    "v_mul_f32 %[t0], %[e2x], %[sy]\n" \
    "v_sub_f32 %[p2], %[p2], %[t0]\n" \
    /* d = dot(e1, p) */ \
    "v_mul_f32 %[d], %[e1x], %[p0]\n" \
    "v_mul_f32 %[t0], %[e1y], %[p1]\n" \
    "v_add_f32 %[d], %[d], %[t0]\n" \
    "v_mul_f32 %[t0], %[e1z], %[p2]\n" \
    "v_add_f32 %[d], %[t0], %[d]\n" \
    /* collision_detection.h:28-35: leave unless !(d < eps) / !(|d| < eps) */ \
    FIRST_STAGE \
    "s_cbranch_execz 1f\n" \
    /* 1 / d (rcp_refined), q = o - a in the shadow of the reciprocal */ \
    "v_rcp_f32 %[r], %[d]\n" \
    "v_subrev_f32 %[q0], %[ax], %[ox]\n" \
    "v_subrev_f32 %[q1], %[ay], %[oy]\n" \
    "v_subrev_f32 %[q2], %[az], %[oz]\n" \
    "v_fma_f32 %[e], -%[d], %[r], 1.0\n" \
    "v_fmac_f32 %[r], %[e], %[r]\n" \
    "v_fma_f32 %[e], -%[d], %[r], 1.0\n" \
    "v_fma_f32 %[e], %[e], %[r], %[r]\n" \
    "v_fma_f32 %[d], -%[d], %[e], 1.0\n" \
    "v_fmac_f32 %[e], %[d], %[r]\n" \
    /* u = d_inv * dot(q, p); leave unless !(u < 0) && !(u > 1) */ \
    "v_mul_f32 %[u], %[q0], %[p0]\n" \
    "v_mul_f32 %[t0], %[q1], %[p1]\n" \
    "v_add_f32 %[u], %[u], %[t0]\n" \
    "v_mul_f32 %[t0], %[q2], %[p2]\n" \
    "v_add_f32 %[u], %[t0], %[u]\n" \
    "v_mul_f32 %[u], %[u], %[e]\n" \
    "v_cmpx_ngt_f32 vcc, 0, %[u]\n" \
    "v_cmpx_nlt_f32 vcc, 1.0, %[u]\n" \
    "s_cbranch_execz 1f\n" \
    /* w = cross(q, e1) */ \
    "v_mul_f32 %[p0], %[e1z], %[q1]\n" \
    "v_mul_f32 %[t0], %[e1y], %[q2]\n" \
    "v_sub_f32 %[p0], %[p0], %[t0]\n" \
    "v_mul_f32 %[p1], %[e1x], %[q2]\n" \
    "v_mul_f32 %[t0], %[e1z], %[q0]\n" \
    "v_sub_f32 %[p1], %[p1], %[t0]\n" \
    "v_mul_f32 %[p2], %[e1y], %[q0]\n" \
    "v_mul_f32 %[t0], %[e1x], %[q1]\n" \
    "v_sub_f32 %[p2], %[p2], %[t0]\n" \
    /* v = d_inv * dot(w, s); leave unless !(v < 0) && !(u + v > 1) */ \
    "v_mul_f32 %[d], %[p0], %[sx]\n" \
    "v_mul_f32 %[t0], %[p1], %[sy]\n" \
    "v_add_f32 %[d], %[d], %[t0]\n" \
    "v_mul_f32 %[t0], %[p2], %[sz]\n" \
    "v_add_f32 %[d], %[t0], %[d]\n" \
    "v_mul_f32 %[d], %[d], %[e]\n" \
    "v_add_f32 %[u], %[u], %[d]\n" \
    "v_cmpx_ngt_f32 vcc, 0, %[d]\n" \
    "v_cmpx_nlt_f32 vcc, 1.0, %[u]\n" \
    "s_cbranch_execz 1f\n" \
    /* t = d_inv * dot(e2, w); an occluder unless t == inf or t < eps (renderer.cc:298), and only below the limit (:302) */ \
    "v_mul_f32 %[d], %[e2x], %[p0]\n" \
    "v_mul_f32 %[t0], %[e2y], %[p1]\n" \
    "v_add_f32 %[d], %[d], %[t0]\n" \
    "v_mul_f32 %[t0], %[e2z], %[p2]\n" \
    "v_add_f32 %[d], %[t0], %[d]\n" \
    "v_mul_f32 %[d], %[d], %[e]\n" \
    "v_cmpx_neq_f32 vcc, 0x7f800000, %[d]\n" \
    "v_cmpx_ngt_f32 vcc, %[eps], %[d]\n" \
    "v_cmpx_lt_f32 vcc, %[d], %[tl]\n" \
    "s_or_b64 %[found], %[found], exec\n" \
    "1:\n" \
    "s_mov_b64 exec, %[save]\n"
#define RT_ANY_HIT_OUTPUTS                                                                                                      \
    [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [d] "=&v"(d), [r] "=&v"(r), [e] "=&v"(e), [q0] "=&v"(q0), [q1] "=&v"(q1),     \
        [q2] "=&v"(q2), [u] "=&v"(u), [t0] "=&v"(t0), [save] "=&s"(save), [found] "+s"(found)
#define RT_ANY_HIT_INPUTS(REC)                                                                                                  \
    [need] "s"(need), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [sx] "v"(s.x), [sy] "v"(s.y), [sz] "v"(s.z), [ax] REC(ax),  \
        [ay] REC(ay), [az] REC(az), [e1x] REC(e1x), [e1y] REC(e1y), [e1z] REC(e1z), [e2x] REC(e2x), [e2y] REC(e2y),            \
        [e2z] REC(e2z), [tl] "v"(t_limit), [eps] "s"(eps)

#define RT_ANY_HIT_CULL "v_cmpx_ngt_f32 vcc, %[eps], %[d]\n"                                         /* !(d < eps) */
#define RT_ANY_HIT_NOCULL "v_and_b32 %[t0], 0x7fffffff, %[d]\n v_cmpx_ngt_f32 vcc, %[eps], %[t0]\n" /* !(|d| < eps) */
// One asm statement: save EXEC, select the lanes (PROLOGUE), run the stages, restore EXEC.
#define RT_ANY_HIT_ASM(PROLOGUE, FIRST_STAGE, REC, EXTRA_INPUTS)                                \
    asm volatile("s_mov_b64 %[save], exec\n" PROLOGUE RT_ANY_HIT_STAGES(FIRST_STAGE)            \
                 : RT_ANY_HIT_OUTPUTS                                                            \
                 : RT_ANY_HIT_INPUTS(REC) EXTRA_INPUTS                                           \
                 : "vcc", "scc")
#define RT_ANY_HIT_MASKED_PROLOGUE                \
    "s_andn2_b64 exec, %[need], %[found]\n"       \
    "v_and_b32 %[t0], %[lm], %[reach]\n"          \
    "v_cmpx_ne_u32 vcc, 0, %[t0]\n"               \
    "s_cbranch_execz 1f\n"
#define RT_ANY_HIT_MASKED_INPUTS , [reach] "v"(reach), [lm] "s"(lm)
// MASKED = false: tests the lanes of `need`.  MASKED = true: `need` is the set of lanes that carry a ray; the lanes tested
// are those of them that have no occluder yet and whose `reach` (the leaves the lane reaches) intersects lm (the leaves
// that list the triangle) -- the selection of the small-tree and clustered passes, folded into the EXEC mask as well.
// t_seen (diagnostics): receives, for the lanes added to `found`, the distance the last stage computed.
template <bool CULL, bool MASKED = false>
__device__ __forceinline__ void any_hit_exec(lanemask need, f3 o, f3 s, float ax, float ay, float az, float e1x, float e1y, float e1z,
                                             float e2x, float e2y, float e2z, float t_limit, lanemask& found, uint32_t reach = 0,
                                             uint32_t lm = 0, float* t_seen = nullptr) {
    float p0, p1, p2, d, r, e, q0, q1, q2, u, t0;
    lanemask save;
    const float eps = kEps;
    if constexpr (MASKED && CULL) RT_ANY_HIT_ASM(RT_ANY_HIT_MASKED_PROLOGUE, RT_ANY_HIT_CULL, "s", RT_ANY_HIT_MASKED_INPUTS);
    else if constexpr (MASKED) RT_ANY_HIT_ASM(RT_ANY_HIT_MASKED_PROLOGUE, RT_ANY_HIT_NOCULL, "s", RT_ANY_HIT_MASKED_INPUTS);
    else if constexpr (CULL) RT_ANY_HIT_ASM("s_mov_b64 exec, %[need]\n", RT_ANY_HIT_CULL, "s", );
    else RT_ANY_HIT_ASM("s_mov_b64 exec, %[need]\n", RT_ANY_HIT_NOCULL, "s", );
    if (t_seen) *t_seen = d;
}

// The same test for one (ray, triangle) pair per lane: every lane brings its own ray, triangle and limit in vector
// registers (the pair phases of the per-lane and task-pool forms).  Returns the lanes of `need` whose pair is an
// occluder.  Same caller's guarantee as above.
template <bool CULL>
__device__ __forceinline__ lanemask any_hit_exec_pairs(lanemask need, f3 o, f3 s, f3 a, f3 e1, f3 e2, float t_limit, float* t_seen = nullptr) {
    float p0, p1, p2, d, r, e, q0, q1, q2, u, t0;
    lanemask save, found = 0;
    const float eps = kEps;
    const float ax = a.x, ay = a.y, az = a.z, e1x = e1.x, e1y = e1.y, e1z = e1.z, e2x = e2.x, e2y = e2.y, e2z = e2.z;
    if constexpr (CULL) RT_ANY_HIT_ASM("s_mov_b64 exec, %[need]\n", RT_ANY_HIT_CULL, "v", );
    else RT_ANY_HIT_ASM("s_mov_b64 exec, %[need]\n", RT_ANY_HIT_NOCULL, "v", );
    if (t_seen) *t_seen = d;
    return found;
}

// Slab test exactly as src/collision_detection.h:71-86 with the per-ray reciprocals hoisted (the
// reference recomputes the same three quotients for every box).
__device__ __forceinline__ bool ray_box(f3 o, f3 inv, float bx0, float by0, float bz0, float bx1, float by1, float bz1) {
    float t_min = -RT_INF, t_max = RT_INF;
    {
        float t0 = (bx0 - o.x) * inv.x, t1 = (bx1 - o.x) * inv.x;
        if (inv.x < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = std_max(t_min, t0);
        t_max = std_min(t_max, t1);
    }
    {
        float t0 = (by0 - o.y) * inv.y, t1 = (by1 - o.y) * inv.y;
        if (inv.y < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = std_max(t_min, t0);
        t_max = std_min(t_max, t1);
    }
    {
        float t0 = (bz0 - o.z) * inv.z, t1 = (bz1 - o.z) * inv.z;
        if (inv.z < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        t_min = std_max(t_min, t0);
        t_max = std_min(t_max, t1);
    }
    return t_max >= t_min;
}

#ifdef RT_LANE_PROFILE
// Any-hit walks of the wave-together forms.  Plain walk: 0 waves, 1 node visits, 2 leaf visits, 3 triangle tests started.
// Clustered walk: 8 waves, 9 node visits, 10 subtree clusters entered, 11 leaves reached, 12 list records scanned, 13 of
// them listed by no reached leaf, 14 of them needed by no live lane, 15 triangle tests started.
__device__ unsigned long long g_wave_prof[16];
#define RT_WPROF(k, v) do { if (__lane_id() == 0) atomicAdd(&g_wave_prof[k], (unsigned long long)(v)); } while (0)
constexpr bool kHandScheduledTests = false;  // (the counters' lane branches and the ISA block's scalar operands do not mix)
#else
#define RT_WPROF(k, v) do { } while (0)
constexpr bool kHandScheduledTests = true;
#endif

struct HitState {
    float t;         // closest: best t so far (+inf = none)
    uint32_t slot;   // closest: global leaf slot of the winner (small-tree form: ordinal of its first reached leaf)
    uint32_t tri;    // closest: its triangle index
    lanemask found;  // any-hit: lanes that have found an occluder (wave-uniform)
    lanemask tie;    // closest, pruned walk: lanes whose best distance was met by two distinct triangles (wave-uniform)
};

// ---- the part of a ray that is walked -----------------------------------------------------------------------------
// The reference's slab test (collision_detection.h:71-86) starts its running bounds at -inf / +inf, so Octree::Search
// returns the leaves along the whole LINE of the ray: behind its origin and, for a shadow ray, beyond the light.  The
// loops over the candidates then discard what those leaves contribute: a hit needs t >= epsilon (renderer.cc:126,298),
// a shadow hit t < distance to the light (:302), a closest hit t < the best so far (:130).  A triangle that is hit at
// distance t contains the hit point, so it is listed by the leaf that contains that point (the triangle/box test of
// Octree::Build assigns a triangle to every octant it touches, octree.cc:100-105), and every box on the way down to
// that leaf contains the point too: their slab intervals contain t.  Starting the running bounds at [lo, hi] instead of
// [-inf, +inf] therefore keeps every leaf that can contribute a hit with lo <= t <= hi, and the candidates it drops
// could only have produced hits the loops discard.  The bounds are padded (kSpanPad of the length, plus kSpanPadAbs of
// the origin's magnitude) far beyond the rounding of the slab products and of t itself; boxes overlap their neighbours by
// 2e-6 of their coordinates on top (octree.cc:82-97).  Span{-inf, +inf} is the reference's walk.
struct Span {
    float lo, hi;
};
constexpr float kSpanPad = 0x1p-7f, kSpanPadAbs = 0x1p-10f;
__device__ __forceinline__ Span span_all() { return Span{-RT_INF, RT_INF}; }
__device__ __forceinline__ float span_pad_abs(f3 o) {
    return (__builtin_fmaxf(__builtin_fmaxf(fabsf(o.x), fabsf(o.y)), __builtin_fmaxf(fabsf(o.z), 1.0f))) * kSpanPadAbs;
}
// Any-hit ray of length dist (renderer.cc:289-302): occluders lie in [epsilon, dist).  A ray whose length or origin is
// not finite is walked like the reference walks it.
__device__ __forceinline__ Span span_segment(f3 o, float dist, bool clip) {
    const float pad = dist * kSpanPad + span_pad_abs(o);
    const bool bounded = clip && pad < RT_INF;  // (false for a NaN or infinite length or origin as well)
    return Span{bounded ? -pad : -RT_INF, bounded ? dist + pad : RT_INF};
}
// Closest-hit ray: hits lie in [epsilon, best so far]; the upper end follows the best distance during the walk.
__device__ __forceinline__ float span_lo_closest(f3 o, bool clip) { return clip ? -span_pad_abs(o) : -RT_INF; }
__device__ __forceinline__ float span_hi_closest(float best, float pad_abs) { return best + (best * kSpanPad + pad_abs); }

// ---- wide scalar loads -------------------------------------------------------------------------------------------
// Records are fetched in blocks of 16 dwords (s_load_dwordx16) so that a node visit or a group of 4 triangles
// costs one round trip to the scalar cache instead of one per field.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef u32x16 u32x16_a4 __attribute__((aligned(4)));
struct Block48 {  // 48 dwords: one node's 8 boxes, or 4 triangle records
    u32x16 v[3];
};
__device__ __forceinline__ u32x16 sload16(cu32_p p) { return *(const RT_CONST u32x16_a4*)p; }
__device__ __forceinline__ Block48 sload48(cu32_p p) {
    Block48 b;
    b.v[0] = sload16(p);
    b.v[1] = sload16(p + 16);
    b.v[2] = sload16(p + 32);
    return b;
}
#define RT_W(b, i) ((b).v[(i) >> 4][(i) & 15])
#define RT_F(b, i) __uint_as_float(RT_W(b, i))

constexpr int kTriChunk = 4;  // triangle records per scalar-load group (arrays are padded by this many records)

// One record of a leaf's 4-record block (record j, global slot `slot`) for the lanes in `mine`.
// TIE (closest hit, pruned walks): also note the lanes whose best distance is met by two distinct triangles.
template <bool ANY, bool CULL, bool FAST = false, bool TIE = false>
__device__ __forceinline__ void leaf_one(const Block48& blk, const int j, uint32_t slot, lanemask mine, f3 o, f3 d, float t_limit,
                                         HitState& h) {
    if (mine == 0) return;
    if (ANY) RT_WPROF(3, 1);
    if constexpr (ANY && FAST && kHandScheduledTests) {
        any_hit_exec<CULL>(mine, o, d, RT_F(blk, 12 * j + 0), RT_F(blk, 12 * j + 1), RT_F(blk, 12 * j + 2), RT_F(blk, 12 * j + 3),
                           RT_F(blk, 12 * j + 4), RT_F(blk, 12 * j + 5), RT_F(blk, 12 * j + 6), RT_F(blk, 12 * j + 7), RT_F(blk, 12 * j + 8),
                           t_limit, h.found);
        return;
    }
    const f3 a = mk(RT_F(blk, 12 * j + 0), RT_F(blk, 12 * j + 1), RT_F(blk, 12 * j + 2));
    const f3 e1 = mk(RT_F(blk, 12 * j + 3), RT_F(blk, 12 * j + 4), RT_F(blk, 12 * j + 5));
    const f3 e2 = mk(RT_F(blk, 12 * j + 6), RT_F(blk, 12 * j + 7), RT_F(blk, 12 * j + 8));
    float t = RT_INF;
    const lanemask hit = ray_triangle<CULL>(mine, o, d, a, e1, e2, t);
    if (hit == 0) return;
    if (ANY) {
        h.found |= hit & ballot64(t < t_limit);  // renderer.cc:302
    } else {
        // renderer.cc:130: strictly smaller t wins; equal t keeps the earlier candidate, which is the smaller global slot.
        const lanemask less = hit & ballot64(t < h.t), same_t = hit & ballot64(t == h.t);
        const lanemask better = less | (same_t & ballot64(slot < h.slot));
        const uint32_t tri = RT_W(blk, 12 * j + 9);
        // Two distinct triangles at the best distance: which one the reference takes depends on every leaf that lists
        // them, also on leaves a pruned walk leaves out -- such rays are walked again without pruning (traverse<>).
        if (TIE) h.tie = (h.tie & ~less) | (same_t & ballot64(tri != h.tri));
        if (better == 0) return;
        const bool b = lane_of(better);
        h.t = b ? t : h.t;
        h.slot = b ? slot : h.slot;
        h.tri = b ? tri : h.tri;
    }
}

// Tests the triangles of leaf slots [begin, begin+count) for the lanes in `mine`.  Records are read four at a time;
// those past the leaf's end (the next leaf's, or the array's padding) get an empty lane mask.
template <bool ANY, bool CULL, bool FAST = false, bool TIE = false>
__device__ __forceinline__ void leaf_tests(const DevOctree& T, uint32_t begin, uint32_t count, lanemask mine, f3 o, f3 d, float t_limit,
                                           HitState& h) {
    cu32_p base = as_const_u(T.slots);
    for (uint32_t k0 = 0; k0 < count; k0 += kTriChunk) {
        const Block48 blk = sload48(base + (size_t)(begin + k0) * 12);
        const lanemask none = 0;
        leaf_one<ANY, CULL, FAST, TIE>(blk, 0, begin + k0 + 0, ANY ? (mine & ~h.found) : mine, o, d, t_limit, h);
        leaf_one<ANY, CULL, FAST, TIE>(blk, 1, begin + k0 + 1, k0 + 1 < count ? (ANY ? (mine & ~h.found) : mine) : none, o, d, t_limit, h);
        leaf_one<ANY, CULL, FAST, TIE>(blk, 2, begin + k0 + 2, k0 + 2 < count ? (ANY ? (mine & ~h.found) : mine) : none, o, d, t_limit, h);
        leaf_one<ANY, CULL, FAST, TIE>(blk, 3, begin + k0 + 3, k0 + 3 < count ? (ANY ? (mine & ~h.found) : mine) : none, o, d, t_limit, h);
        if (ANY && (mine & ~h.found) == 0) return;  // every ray that reached this leaf is decided
    }
}

// Small-tree form of the leaf work.  `reach` holds, per lane, the set of leaves the lane's ray reaches (one bit
// per leaf ordinal).  Instead of walking leaf after leaf -- which tests a triangle once per leaf that lists it and,
// for a wave, once per leaf ANY lane reaches -- the wave walks the triangle array once: triangle j is a candidate
// of a lane iff it is listed by a leaf the lane reaches.  The candidate SET per ray is unchanged, so any-hit
// answers and closest distances are unchanged.  The reference's tie rule (first in candidate-list order) becomes:
// smaller ordinal of the first reached leaf that lists the triangle, then smaller triangle index (leaf lists are
// ascending); the loop runs in ascending j, so an equal-t later triangle wins only with a smaller first leaf.
// One triangle record of the small-tree pass (record jj of the 4-record block, triangle index j).
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void flat_one(const Block48& blk, const int jj, uint32_t j, lanemask alive, uint32_t reach, f3 o, f3 d,
                                         float t_limit, HitState& h) {
    const uint32_t lm = RT_W(blk, 12 * jj + 10);
    if constexpr (ANY && FAST && kHandScheduledTests) {  // (alive: every lane with a ray; the test leaves out those that have an occluder)
        any_hit_exec<CULL, true>(alive, o, d, RT_F(blk, 12 * jj + 0), RT_F(blk, 12 * jj + 1), RT_F(blk, 12 * jj + 2), RT_F(blk, 12 * jj + 3),
                                 RT_F(blk, 12 * jj + 4), RT_F(blk, 12 * jj + 5), RT_F(blk, 12 * jj + 6), RT_F(blk, 12 * jj + 7),
                                 RT_F(blk, 12 * jj + 8), t_limit, h.found, reach, lm);
        return;
    }
    const lanemask need = alive & ballot64((reach & lm) != 0);
    if (need == 0) return;
    const f3 a = mk(RT_F(blk, 12 * jj + 0), RT_F(blk, 12 * jj + 1), RT_F(blk, 12 * jj + 2));
    const f3 e1 = mk(RT_F(blk, 12 * jj + 3), RT_F(blk, 12 * jj + 4), RT_F(blk, 12 * jj + 5));
    const f3 e2 = mk(RT_F(blk, 12 * jj + 6), RT_F(blk, 12 * jj + 7), RT_F(blk, 12 * jj + 8));
    float t = RT_INF;
    const lanemask hit = ray_triangle<CULL>(need, o, d, a, e1, e2, t);
    if (hit == 0) return;
    if (ANY) {
        h.found |= hit & ballot64(t < t_limit);  // renderer.cc:302
    } else {
        const uint32_t first_leaf = (uint32_t)__builtin_ctz(reach & lm);  // garbage on lanes outside `need`
        const lanemask better = hit & (ballot64(t < h.t) | (ballot64(t == h.t) & ballot64(first_leaf < h.slot)));
        if (better == 0) return;
        const bool b = lane_of(better);
        h.t = b ? t : h.t;
        h.slot = b ? first_leaf : h.slot;
        h.tri = b ? j : h.tri;
    }
}

#ifdef RT_LDS_STAGED_TRIS
// Measurement variant (tools/ab.sh): the same pass with the triangle batches staged through LDS -- 48 lanes fetch the 48
// dwords of a 4-record block with one coalesced vector load, write them to the wave's LDS buffer, and every operand of
// the tests is read back from LDS (same address in all lanes: a broadcast) into vector registers -- instead of scalar
// loads that feed the VALU from scalar registers.
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void flat_tests(const DevOctree& T, lanemask all, uint32_t reach, f3 o, f3 d, float t_limit, HitState& h) {
    __shared__ uint32_t stage_all[16][48];
    typedef __attribute__((address_space(3))) volatile uint32_t lds_word;
    lds_word* stage = (lds_word*)stage_all[threadIdx.x >> 6];
    const uint32_t* base = reinterpret_cast<const uint32_t*>(T.flat_tris);
    const uint32_t lane = __lane_id();
    const uint32_t n = T.n_flat_tris;
    for (uint32_t j0 = 0; j0 < n; j0 += kTriChunk) {
        uint32_t w = 0;
        if (lane < 48) w = base[(size_t)j0 * 12 + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) stage[lane] = w;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const lanemask alive = ANY ? (all & ~h.found) : all;
            const uint32_t lm = stage[12 * jj + 10];
            const lanemask need = alive & ballot64((reach & lm) != 0);
            if (need == 0) continue;
            const f3 a = mk(__uint_as_float(stage[12 * jj + 0]), __uint_as_float(stage[12 * jj + 1]), __uint_as_float(stage[12 * jj + 2]));
            const f3 e1 = mk(__uint_as_float(stage[12 * jj + 3]), __uint_as_float(stage[12 * jj + 4]), __uint_as_float(stage[12 * jj + 5]));
            const f3 e2 = mk(__uint_as_float(stage[12 * jj + 6]), __uint_as_float(stage[12 * jj + 7]), __uint_as_float(stage[12 * jj + 8]));
            float t = RT_INF;
            const lanemask hit = ray_triangle<CULL>(need, o, d, a, e1, e2, t);
            if (hit == 0) continue;
            if (ANY) {
                h.found |= hit & ballot64(t < t_limit);
            } else {
                const uint32_t first_leaf = (uint32_t)__builtin_ctz(reach & lm);
                const lanemask better = hit & (ballot64(t < h.t) | (ballot64(t == h.t) & ballot64(first_leaf < h.slot)));
                if (better == 0) continue;
                const bool b = lane_of(better);
                h.t = b ? t : h.t;
                h.slot = b ? first_leaf : h.slot;
                h.tri = b ? (j0 + jj) : h.tri;
            }
        }
        if (ANY && (all & ~h.found) == 0) return;
    }
}
#else
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void flat_tests(const DevOctree& T, lanemask all, uint32_t reach, f3 o, f3 d, float t_limit, HitState& h) {
    cu32_p base = as_const_u(T.flat_tris);
    const uint32_t n = T.n_flat_tris;
    // The array is padded to whole blocks with records whose leaf mask is 0: no lane ever needs them.
    for (uint32_t j0 = 0; j0 < n; j0 += kTriChunk) {
        const Block48 blk = sload48(base + (size_t)j0 * 12);
        flat_one<ANY, CULL, FAST>(blk, 0, j0 + 0, (ANY && !(FAST && kHandScheduledTests)) ? (all & ~h.found) : all, reach, o, d, t_limit, h);
        flat_one<ANY, CULL, FAST>(blk, 1, j0 + 1, (ANY && !(FAST && kHandScheduledTests)) ? (all & ~h.found) : all, reach, o, d, t_limit, h);
        flat_one<ANY, CULL, FAST>(blk, 2, j0 + 2, (ANY && !(FAST && kHandScheduledTests)) ? (all & ~h.found) : all, reach, o, d, t_limit, h);
        flat_one<ANY, CULL, FAST>(blk, 3, j0 + 3, (ANY && !(FAST && kHandScheduledTests)) ? (all & ~h.found) : all, reach, o, d, t_limit, h);
        if (ANY && (all & ~h.found) == 0) return;  // every ray of the wave is decided
    }
}

#endif

// The wave-uniform traversal stack: entry i lives in lane (i & 63) of register (i >> 6) of each field.
// An entry is one octant that some lanes reach: its NodeRec::info pair and the mask of those lanes.
struct WaveStack {
    // named registers on purpose: an array indexed by (sp >= 64) would be placed in scratch memory
    int f0a = 0, f0b = 0, f1a = 0, f1b = 0, loa = 0, lob = 0, hia = 0, hib = 0;
    int sp = 0;
    __device__ __forceinline__ void push(uint32_t info0, uint32_t info1, lanemask m) {
        // branch-free: at most one of the two compares holds for a lane
        const int lane = (int)__lane_id();
        const bool ha = lane == sp, hb = lane + 64 == sp;
        f0a = ha ? (int)info0 : f0a;
        f1a = ha ? (int)info1 : f1a;
        loa = ha ? (int)(uint32_t)m : loa;
        hia = ha ? (int)(uint32_t)(m >> 32) : hia;
        f0b = hb ? (int)info0 : f0b;
        f1b = hb ? (int)info1 : f1b;
        lob = hb ? (int)(uint32_t)m : lob;
        hib = hb ? (int)(uint32_t)(m >> 32) : hib;
        ++sp;
    }
    __device__ __forceinline__ void pop(uint32_t& info0, uint32_t& info1, lanemask& m) {
        --sp;
        const int l = sp & 63;
        const bool first = sp < 64;
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane(f0a, l), b0 = (uint32_t)__builtin_amdgcn_readlane(f0b, l);
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane(f1a, l), b1 = (uint32_t)__builtin_amdgcn_readlane(f1b, l);
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane(loa, l), b2 = (uint32_t)__builtin_amdgcn_readlane(lob, l);
        const uint32_t a3 = (uint32_t)__builtin_amdgcn_readlane(hia, l), b3 = (uint32_t)__builtin_amdgcn_readlane(hib, l);
        info0 = first ? a0 : b0;
        info1 = first ? a1 : b1;
        m = ((lanemask)(first ? a3 : b3) << 32) | (first ? a2 : b2);
    }
};

// The slab products of one node for one ray, per axis and per half (octant bit 0 / 1): collision_detection.h:77-81,
// already ordered by the sign of inv.  pl = the 12 planes of a NodeRec (lo[3][2], hi[3][2]).
struct SlabSides {
    float tn[3][2], tf[3][2];
};
__device__ __forceinline__ void slab_sides(const u32x16& pl, f3 o, f3 inv, SlabSides& s) {
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float oo = comp(o, ax), ii = comp(inv, ax);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float t0 = (__uint_as_float(pl[ax * 2 + b]) - oo) * ii;
            float t1 = (__uint_as_float(pl[6 + ax * 2 + b]) - oo) * ii;
            if (ii < 0.0f) { const float sw = t0; t0 = t1; t1 = sw; }
            s.tn[ax][b] = t0;
            s.tf[ax][b] = t1;
        }
    }
}
// collision_detection.h:72-85 for octant c in its order: x, then y, then z, folded into the running max / min from
// -inf / +inf (the reference), or from the bounds of the part of the ray that is walked (Span).
__device__ __forceinline__ bool octant_passes(const SlabSides& s, const int c, float lo = -RT_INF, float hi = RT_INF) {
    const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
    float t_min = slab_max(lo, s.tn[0][bx]), t_max = slab_min(hi, s.tf[0][bx]);
    t_min = slab_max(t_min, s.tn[1][by]);
    t_max = slab_min(t_max, s.tf[1][by]);
    t_min = slab_max(t_min, s.tn[2][bz]);
    t_max = slab_min(t_max, s.tf[2][bz]);
    return t_max >= t_min;
}

// valid: this lane carries a ray.  For ANY, t_limit is the distance to the light.
// sp: the part of the ray whose boxes are walked (closest-hit walks that follow their best hit: traverse_packet_pruned).
template <bool ANY, bool CULL, bool FLAT, bool FAST = false>
__device__ __forceinline__ void traverse_impl(const DevOctree& T, lanemask all, f3 o, f3 d, float t_limit, Span sp, HitState& h) {
    const f3 inv = reciprocal3(d, lane_of(all));  // collision_detection.h:76
    WaveStack st;
    uint32_t reach = 0;
    cu32_p nodes = as_const_u(FLAT ? T.flat_nodes : T.nodes);
    uint32_t i0 = T.root_info0, i1 = T.root_info1;
    lanemask mine = all;
    if (ANY && !FLAT) RT_WPROF(0, 1);
    for (;;) {
        if (i1 & kChildInternal) {
            if (ANY && !FLAT) RT_WPROF(1, 1);
            // one round trip for the node: 12 planes (+ the non-empty mask) and the 8 info pairs (NodeRec, 32 dwords)
            cu32_p rec = nodes + (size_t)i0 * 32;
            const u32x16 pl = sload16(rec);
            const u32x16 info = sload16(rec + 16);  // dwords 16..31 = info[8][2]
            SlabSides sl;
            slab_sides(pl, o, inv, sl);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t c1 = info[2 * c + 1];
                if (c1 == 0) continue;  // null or empty octant (octree.cc:195)
                const lanemask pm = mine & ballot64(octant_passes(sl, c, sp.lo, sp.hi));
                if (pm == 0) continue;
                if (FLAT && !(c1 & kChildInternal)) reach |= lane_of(pm) ? (1u << info[2 * c]) : 0u;
                else st.push(info[2 * c], c1, pm);
            }
        } else {
            if (ANY && !FLAT) RT_WPROF(2, 1);
            leaf_tests<ANY, CULL, FAST>(T, i0, i1, mine, o, d, t_limit, h);
        }
        // next entry that still has a live lane
        bool got = false;
        while (st.sp > 0) {
            st.pop(i0, i1, mine);
            if (ANY) mine &= ~h.found;
            if (mine != 0) {
                got = true;
                break;
            }
        }
        if (!got) break;
    }
    if (FLAT) flat_tests<ANY, CULL, FAST>(T, all, reach, o, d, t_limit, h);
}

// ---- ordered, pruned closest-hit walk of the wave ---------------------------------------------------------------------
// The wave-together closest-hit walk with the upper end of the walked part following each lane's best hit (Span).  Two
// things make the pruning bite.  Order: the octants of a node are pushed far to near in the octant order of the wave's
// first ray (k -> octant k ^ flip, see node_ordered_nonempty), so the nearest is popped first -- any order gives the same
// result, a good one finds the hits before the boxes behind them are opened.  And every stack entry carries the smallest
// entry distance of the lanes it was pushed for: when it is popped, lanes whose best hit is nearer than that are dropped
// without a load, and an entry no lane still wants costs two scalar instructions instead of a node visit or a leaf's
// triangle tests.
struct WaveStackT : WaveStack {
    int ta = 0, tb = 0;  // per entry: bits of the smallest entry distance
    __device__ __forceinline__ void push_t(uint32_t info0, uint32_t info1, lanemask m, float t_enter) {
        const int lane = (int)__lane_id();
        ta = (lane == sp) ? (int)__float_as_uint(t_enter) : ta;
        tb = (lane + 64 == sp) ? (int)__float_as_uint(t_enter) : tb;
        push(info0, info1, m);
    }
    __device__ __forceinline__ void pop_t(uint32_t& info0, uint32_t& info1, lanemask& m, float& t_enter) {
        pop(info0, info1, m);  // (sp now indexes the popped entry)
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(ta, sp & 63), b = (uint32_t)__builtin_amdgcn_readlane(tb, sp & 63);
        t_enter = __uint_as_float(sp < 64 ? a : b);
    }
};

// Minimum over the wave of x (lanes that do not take part pass +inf), as a wave-uniform value.
__device__ __forceinline__ float wave_min_f32(float x) {
#define RT_MIN_STEP(CTRL, ROW_MASK)                                                                                         \
    x = __builtin_fminf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false)))
    RT_MIN_STEP(0x111, 0xf);  // row_shr:1 (lanes without a source keep their own value)
    RT_MIN_STEP(0x112, 0xf);  // row_shr:2
    RT_MIN_STEP(0x114, 0xf);  // row_shr:4
    RT_MIN_STEP(0x118, 0xf);  // row_shr:8
    RT_MIN_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    RT_MIN_STEP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef RT_MIN_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// collision_detection.h:72-85 for octant c with the running bounds started at [lo, hi]; t_enter = the running maximum.
__device__ __forceinline__ bool octant_interval(const SlabSides& s, const int c, float lo, float hi, float& t_enter) {
    const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
    float t_min = slab_max(lo, s.tn[0][bx]), t_max = slab_min(hi, s.tf[0][bx]);
    t_min = slab_max(t_min, s.tn[1][by]);
    t_max = slab_min(t_max, s.tf[1][by]);
    t_min = slab_max(t_min, s.tn[2][bz]);
    t_max = slab_min(t_max, s.tf[2][bz]);
    t_enter = t_min;
    return t_max >= t_min;
}

template <int FLIP>
__device__ __forceinline__ void push_octants_ordered(const SlabSides& sl, const u32x16& info, lanemask mine, float lo, float hi, WaveStackT& st) {
#pragma unroll
    for (int k = 7; k >= 0; --k) {  // far to near
        const int c = k ^ FLIP;
        const uint32_t c1 = info[2 * c + 1];
        if (c1 == 0) continue;  // null or empty octant (octree.cc:195)
        float t_enter;
        const bool pass = octant_interval(sl, c, lo, hi, t_enter);
        const lanemask pm = mine & ballot64(pass);
        if (pm == 0) continue;
        st.push_t(info[2 * c], c1, pm, wave_min_f32(lane_of(pm) ? t_enter : RT_INF));
    }
}

template <bool CULL>
__device__ __forceinline__ void traverse_packet_pruned(const DevOctree& T, lanemask all, f3 o, f3 d, float lo, HitState& h) {
    const f3 inv = reciprocal3(d, lane_of(all));  // collision_detection.h:76
    const float pad_abs = span_pad_abs(o);
    // the octant order of the wave: that of its first ray
    const uint32_t flip_lane = (inv.x < 0.0f ? 4u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 1u : 0u);
    const uint32_t flip = (uint32_t)__builtin_amdgcn_readlane((int)flip_lane, __builtin_ctzll(all));
    WaveStackT st;
    cu32_p nodes = as_const_u(T.nodes);
    uint32_t i0 = T.root_info0, i1 = T.root_info1;
    lanemask mine = all;
    for (;;) {
        if (i1 & kChildInternal) {
            cu32_p rec = nodes + (size_t)i0 * 32;
            const u32x16 pl = sload16(rec);
            const u32x16 info = sload16(rec + 16);
            SlabSides sl;
            slab_sides(pl, o, inv, sl);
            const float hi = span_hi_closest(h.t, pad_abs);
            switch (flip) {
                case 0: push_octants_ordered<0>(sl, info, mine, lo, hi, st); break;
                case 1: push_octants_ordered<1>(sl, info, mine, lo, hi, st); break;
                case 2: push_octants_ordered<2>(sl, info, mine, lo, hi, st); break;
                case 3: push_octants_ordered<3>(sl, info, mine, lo, hi, st); break;
                case 4: push_octants_ordered<4>(sl, info, mine, lo, hi, st); break;
                case 5: push_octants_ordered<5>(sl, info, mine, lo, hi, st); break;
                case 6: push_octants_ordered<6>(sl, info, mine, lo, hi, st); break;
                default: push_octants_ordered<7>(sl, info, mine, lo, hi, st); break;
            }
        } else {
            leaf_tests<false, CULL, false, true>(T, i0, i1, mine, o, d, 0.0f, h);
        }
        // next entry that some lane can still gain from: a lane whose best hit is nearer than the entry's nearest point is out
        bool got = false;
        while (st.sp > 0) {
            float t_enter;
            st.pop_t(i0, i1, mine, t_enter);
            mine &= ballot64(span_hi_closest(h.t, pad_abs) >= t_enter);
            if (mine != 0) {
                got = true;
                break;
            }
        }
        if (!got) break;
    }
}

// One record of a cluster's triangle list (record jj of the 4-record block).  lm = the cluster's leaves that list the
// triangle, wreach = the leaves some lane of the wave reaches, reach = the leaves this lane reaches.
template <bool CULL, bool FAST = false>
__device__ __forceinline__ void cluster_one(const Block48& blk, const int jj, uint32_t wreach, lanemask alive, uint32_t reach, f3 o, f3 d,
                                            float t_limit, HitState& h) {
    const uint32_t lm = RT_W(blk, 12 * jj + 10);
    RT_WPROF(12, 1);
    if ((lm & wreach) == 0) {
        RT_WPROF(13, 1);
        return;
    }
    if constexpr (FAST && kHandScheduledTests) {  // (alive: every lane with a ray; the test leaves out those that have an occluder)
        RT_WPROF(15, 1);
        any_hit_exec<CULL, true>(alive, o, d, RT_F(blk, 12 * jj + 0), RT_F(blk, 12 * jj + 1), RT_F(blk, 12 * jj + 2), RT_F(blk, 12 * jj + 3),
                                 RT_F(blk, 12 * jj + 4), RT_F(blk, 12 * jj + 5), RT_F(blk, 12 * jj + 6), RT_F(blk, 12 * jj + 7),
                                 RT_F(blk, 12 * jj + 8), t_limit, h.found, reach, lm);
        return;
    }
    const lanemask need = alive & ballot64((reach & lm) != 0);
    if (need == 0) {
        RT_WPROF(14, 1);
        return;
    }
    RT_WPROF(15, 1);
    const f3 a = mk(RT_F(blk, 12 * jj + 0), RT_F(blk, 12 * jj + 1), RT_F(blk, 12 * jj + 2));
    const f3 e1 = mk(RT_F(blk, 12 * jj + 3), RT_F(blk, 12 * jj + 4), RT_F(blk, 12 * jj + 5));
    const f3 e2 = mk(RT_F(blk, 12 * jj + 6), RT_F(blk, 12 * jj + 7), RT_F(blk, 12 * jj + 8));
    float t = RT_INF;
    const lanemask hit = ray_triangle<CULL>(need, o, d, a, e1, e2, t);
    if (hit == 0) return;
    h.found |= hit & ballot64(t < t_limit);  // renderer.cc:302
}

// Any-hit walk over the clustered form.  Leaves are never pushed: the lanes that reach leaf k of the current cluster
// set bit k of `reach` and the wave sets bit k of `wreach`.  A cluster made of a node's leaf children is complete after
// that node's visit; a subtree cluster once the stack is back at its height of the moment its root was visited.  Then
// the cluster's list of distinct triangles is walked once: a triangle that a reached leaf lists is tested for all lanes
// that reach any leaf listing it.  The candidate set of every ray is the one of the plain walk -- only the number of
// times the wave tests a triangle changes -- so the answers are the same.
template <bool CULL, bool FAST = false>
__device__ __forceinline__ void traverse_clustered(const DevOctree& T, lanemask all, f3 o, f3 d, float t_limit, Span sp, HitState& h) {
    const f3 inv = reciprocal3(d, lane_of(all));  // collision_detection.h:76
    WaveStack st;
    cu32_p nodes = as_const_u(T.cl_nodes);
    cu32_p tris = as_const_u(T.cl_tris);
    uint32_t i0 = T.root_info0, i1 = T.cl_root_info1;
    lanemask mine = all;
    int cl_base = -1;  // stack height at the visit of a subtree cluster's root; -1 outside those
    uint32_t reach = 0, wreach = 0;
    uint32_t list_begin = 0, list_count = 0;  // the current cluster's records in T.cl_tris
    RT_WPROF(8, 1);
    for (;;) {  // (i0, i1) is an internal node: only those are pushed
        RT_WPROF(9, 1);
        cu32_p rec = nodes + (size_t)i0 * 32;
        const u32x16 pl = sload16(rec);
        const u32x16 info = sload16(rec + 16);
        if (cl_base < 0) {  // this node owns the cluster that its leaf children (or, a subtree root, all leaves below it) belong to
            reach = 0;
            wreach = 0;
            list_begin = pl[13];
            list_count = pl[14];
            if (i1 & kChildCluster) {
                RT_WPROF(10, 1);
                cl_base = st.sp;
            }
        }
        SlabSides sl;
        slab_sides(pl, o, inv, sl);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t c1 = info[2 * c + 1];
            if (c1 == 0) continue;  // null or empty octant (octree.cc:195)
            const lanemask pm = mine & ballot64(octant_passes(sl, c, sp.lo, sp.hi));
            if (pm == 0) continue;
            if (c1 & kChildInternal) {
                st.push(info[2 * c], c1, pm);
            } else {
                const uint32_t bit = 1u << (info[2 * c] >> kClusterOrdShift);
                reach |= lane_of(pm) ? bit : 0u;
                wreach |= bit;
                RT_WPROF(11, 1);
            }
        }
        // next node that still has a live lane; a cluster whose walk is complete is tested first
        bool got = false;
        for (;;) {
            if (cl_base < 0 || st.sp == cl_base) {
                cl_base = -1;
                if (wreach != 0) {
                    for (uint32_t k0 = 0; k0 < list_count && (all & ~h.found) != 0; k0 += kTriChunk) {
                        const Block48 blk = sload48(tris + (size_t)(list_begin + k0) * 12);
                        cluster_one<CULL, FAST>(blk, 0, wreach, (FAST && kHandScheduledTests) ? all : (all & ~h.found), reach, o, d, t_limit, h);
                        cluster_one<CULL, FAST>(blk, 1, wreach, (FAST && kHandScheduledTests) ? all : (all & ~h.found), reach, o, d, t_limit, h);
                        cluster_one<CULL, FAST>(blk, 2, wreach, (FAST && kHandScheduledTests) ? all : (all & ~h.found), reach, o, d, t_limit, h);
                        cluster_one<CULL, FAST>(blk, 3, wreach, (FAST && kHandScheduledTests) ? all : (all & ~h.found), reach, o, d, t_limit, h);
                    }
                    wreach = 0;
                }
            }
            if (st.sp == 0) break;
            st.pop(i0, i1, mine);
            mine &= ~h.found;
            if (mine != 0) {
                got = true;
                break;
            }
        }
        if (!got) break;
    }
}

__device__ __forceinline__ void hit_state_reset(HitState& h) {
    h.t = RT_INF;
    h.slot = 0xffffffffu;
    h.tri = 0xffffffffu;
    h.found = 0;
    h.tie = 0;
}

// FAST (any-hit only): the caller guarantees unit-length directions and a scene within the bound of any_hit_exec.
// clip: walk only the part of the ray that can contribute a result (Span) instead of the reference's whole line.  Any
// hit: the segment up to the light.  Closest hit: from the origin to the best hit so far; a ray whose best distance is
// shared by two distinct triangles is then walked a second time like the reference walks it, so that the reference's
// rule for equal distances (first in ITS candidate list, renderer.cc:130) sees every listing leaf.  The small-tree form
// walks whole lines: its tie rule is stated on the reached leaves themselves, and its walk is a node or a few.
template <bool ANY, bool CULL, bool FAST = false>
__device__ __forceinline__ void traverse(const DevOctree& T, bool valid, f3 o, f3 d, float t_limit, bool clip, HitState& h) {
    hit_state_reset(h);
    const lanemask all = ballot64(valid);
    if (T.root_info1 == 0 || all == 0) return;  // empty tree / no rays
    // A root that is a leaf makes every triangle a candidate without any box test (octree.cc:56-63,186-191);
    // traverse_impl handles it as a leaf entry.
    if (T.n_flat_tris) {  // (a tree of at most 32 leaves: its walk is one node or a few, nothing worth clipping)
        traverse_impl<ANY, CULL, true, FAST>(T, all, o, d, t_limit, span_all(), h);
        return;
    }
    if (ANY) {
        traverse_impl<ANY, CULL, false, FAST>(T, all, o, d, t_limit, span_segment(o, t_limit, clip), h);
        return;
    }
    if (!clip) {
        traverse_impl<ANY, CULL, false, FAST>(T, all, o, d, t_limit, span_all(), h);
        return;
    }
    traverse_packet_pruned<CULL>(T, all, o, d, span_lo_closest(o, true), h);
    if (h.tie == 0) return;
    // the rays whose best distance two distinct triangles share: once more, like the reference walks them
    const lanemask again = h.tie;
    HitState w;
    hit_state_reset(w);
    traverse_impl<ANY, CULL, false, FAST>(T, again, o, d, t_limit, span_all(), w);
    const bool redo = lane_of(again);
    h.t = redo ? w.t : h.t;
    h.slot = redo ? w.slot : h.slot;
    h.tri = redo ? w.tri : h.tri;
}

// Any-hit query over the clustered form (T.cl_nodes must be set; the host launches this only for such trees).
template <bool CULL, bool FAST = false>
__device__ __forceinline__ void traverse_any_clustered(const DevOctree& T, bool valid, f3 o, f3 d, float t_limit, bool clip, HitState& h) {
    hit_state_reset(h);
    const lanemask all = ballot64(valid);
    if (T.root_info1 == 0 || all == 0) return;  // empty tree / no rays
    traverse_clustered<CULL, FAST>(T, all, o, d, t_limit, span_segment(o, t_limit, clip), h);
}

}  // namespace rt
