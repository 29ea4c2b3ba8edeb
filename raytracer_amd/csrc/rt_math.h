// rt_math.h -- float3 arithmetic shared by the host octree builder and the gfx950 kernels.
//
// Every operation is written in the association order of the reference's Vec3
// (src/vec3.h:23-64 of the reference) and the whole library is compiled with
// -ffp-contract=off, so that host and device round exactly like the reference's
// x86-64 build (which has no FMA): bit-identical boxes, hit distances and tie-breaks
// depend on it.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

constexpr float kEps = 1e-9f;  // src/util.h:9
#define RT_INF (__builtin_inff())

struct f3 {
    float x, y, z;
};

RT_HD f3 mk(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
#if defined(__HIPCC__)
// Division on the device.  The compiler expands an IEEE f32 division into two v_div_scale, v_rcp, a refinement chain
// of fused multiply-adds, v_div_fmas and v_div_fixup -- 43 cycles per wave (tools/valu_ops.hip), as much as 18
// multiplications.  The scale and fixup steps only act on operands near the ends of the exponent range (zero,
// denormal, huge, or far apart); elsewhere the quotient is the refinement chain alone, which is what the functions
// below compute.  Callers use them only where every lane that matters has both operands within
// [kDivFastMin, kDivFastMax] in magnitude, and the true division otherwise.  tools/rcp_check.hip compares
// rcp_refined with 1.0f / d for EVERY float in the range and div_refined with n / d for billions of operand pairs
// (random and extreme mantissas): identical bits.
constexpr float kDivFastMin = 0x1p-40f, kDivFastMax = 0x1p40f;
constexpr float kRcpFastMax = kDivFastMax;
__device__ __forceinline__ float rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(e1, r1, r1);
    const float e2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e2, r1, q1);
}
// 1 / d per component (collision_detection.h:76) for the lanes with `need`; the other lanes get unspecified values
__device__ __forceinline__ f3 reciprocal3(f3 d, bool need);
// r1: the once-refined reciprocal of d, shared by all numerators of the same denominator
__device__ __forceinline__ float div_refined_r1(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_refined(float n, float d, float r1) {
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}
#endif

#if defined(__HIPCC__)
// n / d for the lanes with `need`: the refinement chain when every such lane has both operands inside the range it is
// checked on (a zero numerator as well: the chain then carries zeros of the numerator's sign through every step, which is
// the quotient), the true division otherwise.
__device__ __forceinline__ float div_scalar(float n, float d, bool need) {
    const float an = fabsf(n), ad = fabsf(d);
    const bool in_range = (an == 0.0f || (an >= kDivFastMin && an <= kDivFastMax)) && ad >= kDivFastMin && ad <= kDivFastMax;
    if (__builtin_amdgcn_ballot_w64(need && !in_range) == 0) return div_refined(n, d, div_refined_r1(d));
    return n / d;
}

__device__ __forceinline__ f3 reciprocal3(f3 d, bool need) {
    // The range rcp_refined is checked on: every component of every lane that matters is either inside it or an exact
    // zero, whose reciprocal is written down directly (+-inf with the zero's sign, what 1.0f / d gives) -- the primary
    // rays of an image wider than high all have dir.y == 0 (the reference's integer aspect ratio, renderer.cc:93), and a
    // zero component must not send the other two down the full division.  fminf / fmaxf drop a NaN operand, so a NaN
    // component does not by itself force the true division; it does not need to: both forms return a NaN for that
    // component, and no caller lets a reciprocal reach an output -- it only feeds comparisons, which a NaN fails whatever
    // its sign and payload.
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const float hi = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float lo_all = __builtin_fminf(__builtin_fminf(ax, ay), az);
    if (__builtin_amdgcn_ballot_w64(need && lo_all == 0.0f) == 0) {  // no zero component in the wave: nothing to patch
        if (__builtin_amdgcn_ballot_w64(need && !(lo_all >= kEps && hi <= kDivFastMax)) == 0) return f3{rcp_refined(d.x), rcp_refined(d.y), rcp_refined(d.z)};
        return f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    }
    const float lo = __builtin_fminf(__builtin_fminf(ax == 0.0f ? 1.0f : ax, ay == 0.0f ? 1.0f : ay), az == 0.0f ? 1.0f : az);
    const bool in_range = lo >= kEps && hi <= kDivFastMax;
    if (__builtin_amdgcn_ballot_w64(need && !in_range) == 0) {
        const float ix = rcp_refined(d.x), iy = rcp_refined(d.y), iz = rcp_refined(d.z);
        return f3{ax == 0.0f ? __builtin_copysignf(RT_INF, d.x) : ix, ay == 0.0f ? __builtin_copysignf(RT_INF, d.y) : iy,
                  az == 0.0f ? __builtin_copysignf(RT_INF, d.z) : iz};
    }
    return f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
}
#endif

// src/vec3.h:25 -- guarded division: |s| <= eps gives the zero vector.
RT_HD f3 div_guard(f3 a, float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the refinement chain with one shared reciprocal when the operands of every active lane that divides are in range
    const bool divides = fabsf(s) > kEps;
    const float lo = __builtin_fminf(__builtin_fminf(fabsf(a.x), fabsf(a.y)), fabsf(a.z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)), fabsf(s));
    // fminf / fmaxf drop NaN operands (see reciprocal3); a NaN component stays a NaN through either form, and a NaN
    // divisor fails `divides`
    const bool in_range = lo >= kDivFastMin && hi <= kDivFastMax;
    if (__builtin_amdgcn_ballot_w64(divides && !in_range) == 0) {
        const float r1 = div_refined_r1(s);
        return divides ? f3{div_refined(a.x, s, r1), div_refined(a.y, s, r1), div_refined(a.z, s, r1)} : f3{0.f, 0.f, 0.f};
    }
#endif
    return fabsf(s) > kEps ? f3{a.x / s, a.y / s, a.z / s} : f3{0.f, 0.f, 0.f};
}
// src/vec3.h:32 -- epsilon equality.
RT_HD bool eq_eps(f3 a, f3 b) { return fabsf(a.x - b.x) < kEps && fabsf(a.y - b.y) < kEps && fabsf(a.z - b.z) < kEps; }
RT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_HD f3 vabs(f3 a) { return f3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// src/vec3.h:43 -- the reference calls ::sqrt(double) on the float sum and narrows the result, which
// equals the correctly rounded float square root.
RT_HD float length(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_HD f3 normalize(f3 a) { return div_guard(a, length(a)); }
RT_HD float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// std::max(a,b) = (a<b)?b:a and std::min(a,b) = (b<a)?b:a: a NaN second argument is ignored, a NaN
// first argument is kept.  The slab test relies on this (src/collision_detection.h:82-83).
RT_HD float std_max(float a, float b) { return (a < b) ? b : a; }
RT_HD float std_min(float a, float b) { return (b < a) ? b : a; }
RT_HD float clamp01(float v) { return std_max(std_min(v, 1.0f), 0.0f); }  // src/util.h:13-16, src/color.h:17-19
RT_HD f3 clamp_color(f3 c) { return f3{clamp01(c.x), clamp01(c.y), clamp01(c.z)}; }

// The shared counter-based generator: a chain of 32-bit avalanche steps over the key fields, reduced to
// 31 bits so that it maps to float exactly like (float)rand() / (float)RAND_MAX (src/triangle.h:43-44,
// src/Renderer/renderer.cc:44-45): k / 2^31, which can round to 1.0.
RT_HD uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7feb352dU;
    h ^= h >> 15;
    h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}
RT_HD uint32_t rng_prefix(uint32_t seed, uint32_t pixel, uint32_t spp_sample, uint32_t node_id) {
    uint32_t h = mix32(seed + 0x9e3779b9U);
    h = mix32(h ^ pixel);
    h = mix32(h ^ (spp_sample + 0x85ebca6bU));
    h = mix32(h ^ (node_id + 0xc2b2ae35U));
    return h;
}
RT_HD uint32_t rng_finish(uint32_t prefix, uint32_t light, uint32_t sample, uint32_t dim) {
    uint32_t h = mix32(prefix ^ (light + 0x27d4eb2fU));
    h = mix32(h ^ (sample + 0x165667b1U));
    h = mix32(h ^ (dim + 0xd3a2646cU));
    return h >> 1;
}
RT_HD float u31_to_float(uint32_t k) { return (float)(int32_t)k / 2147483648.0f; }  // (float)RAND_MAX == 2^31

#if defined(__HIPCC__)
// Running maximum / minimum of the slab test (collision_detection.h:82-83 uses std::max / std::min: a NaN second
// argument is ignored, a NaN first argument would be kept).  The first argument is the running value: it starts at
// -inf / +inf and is only ever replaced by a second argument that compared greater / smaller, so it is never NaN; and
// with a non-NaN first argument the hardware maximum (which also ignores a NaN second argument -- a product, never a
// signaling NaN) returns the same value up to the sign of a zero.  The only use of the results is the comparison
// t_max >= t_min, which the sign of a zero does not change.  v_max_f32 / v_min_f32 are single full-rate instructions
// where compare-and-select is two half-rate ones plus a wait state (tools/valu_ops.hip).
__device__ __forceinline__ float slab_max(float running, float b) { return __builtin_fmaxf(running, b); }
__device__ __forceinline__ float slab_min(float running, float b) { return __builtin_fminf(running, b); }

#endif

}  // namespace rt
