// Checks of the division fast paths of rt_math.h on the GPU:
//  * rcp_refined(d) against 1.0f / d for EVERY float d with kEps <= |d| <= kDivFastMax;
//  * div_refined(n, d, div_refined_r1(d)) against n / d for 2^35 operand pairs with both magnitudes in
//    [kDivFastMin, kDivFastMax]: random exponents and signs, mantissas random or from a set of extreme patterns
//    (all zeros, all ones, single bits at either end), plus n == d and n == nextafter(d).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I raytracer_amd/csrc tools/rcp_check.hip -o tools/rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "rt_math.h"

__global__ void k_rcp(unsigned long long* bad, unsigned long long* tested) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long nb = 0, nt = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float d = __uint_as_float((uint32_t)b);
        if (!(fabsf(d) >= rt::kEps && fabsf(d) <= rt::kDivFastMax)) continue;
        ++nt;
        nb += __float_as_uint(rt::rcp_refined(d)) != __float_as_uint(1.0f / d);
    }
    atomicAdd(bad, nb);
    atomicAdd(tested, nt);
}

__device__ __forceinline__ float make_operand(uint32_t r) {
    // exponent in [-40, 39] (so the magnitude is in [2^-40, 2^40)), sign and mantissa from the remaining bits
    const uint32_t e = 127u - 40u + (r >> 8) % 80u;
    const uint32_t kind = (r >> 4) & 7u;
    uint32_t m = rt::mix32(r ^ 0x9e3779b9u) & 0x7fffffu;
    if (kind == 1) m = 0;
    else if (kind == 2) m = 0x7fffffu;
    else if (kind == 3) m = 1u;
    else if (kind == 4) m = 0x400000u;
    else if (kind == 5) m = 0x7ffffeu;
    return __uint_as_float(((r & 1u) << 31) | (e << 23) | m);
}

__global__ void k_div(unsigned long long* bad, unsigned long long* tested, uint32_t rounds) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0, nt = 0;
    uint32_t h = rt::mix32(tid + 0x85ebca6bu);
    for (uint32_t i = 0; i < rounds; ++i) {
        h = rt::mix32(h + i);
        const float d = make_operand(h);
        h = rt::mix32(h ^ 0xc2b2ae35u);
        float n = make_operand(h);
        if ((i & 63u) == 1) n = d;
        if ((i & 63u) == 2) n = __uint_as_float(__float_as_uint(d) + 1u);
        if (!(fabsf(n) >= rt::kDivFastMin && fabsf(n) <= rt::kDivFastMax && fabsf(d) >= rt::kDivFastMin && fabsf(d) <= rt::kDivFastMax)) continue;
        ++nt;
        nb += __float_as_uint(rt::div_refined(n, d, rt::div_refined_r1(d))) != __float_as_uint(n / d);
    }
    atomicAdd(bad, nb);
    atomicAdd(tested, nt);
}

int main() {
    unsigned long long *d_c, h_c[4] = {0, 0, 0, 0};
    if (hipMalloc(&d_c, sizeof h_c) != hipSuccess) return 2;
    (void)hipMemset(d_c, 0, sizeof h_c);
    k_rcp<<<4096, 256>>>(d_c, d_c + 1);
    k_div<<<8192, 256>>>(d_c + 2, d_c + 3, 1u << 14);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(h_c, d_c, sizeof h_c, hipMemcpyDeviceToHost);
    printf("reciprocal: %llu values, %llu differ from 1.0f / d\n", h_c[1], h_c[0]);
    printf("division:   %llu pairs, %llu differ from n / d\n", h_c[3], h_c[2]);
    return (h_c[0] == 0 && h_c[2] == 0) ? 0 : 1;
}
