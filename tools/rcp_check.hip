// Exhaustive check of rt::rcp_exact's fast path (rt_math.h): for every float d with kEps <= |d| <= 2^40 the
// refinement sequence must give the bits of the IEEE division 1.0f / d.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I raytracer_amd/csrc tools/rcp_check.hip -o tools/rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "rt_math.h"

__global__ void k(unsigned long long* bad, unsigned long long* tested, uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long nb = 0, nt = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float d = __uint_as_float((uint32_t)b);
        if (!(fabsf(d) >= rt::kEps && fabsf(d) <= rt::kRcpFastMax)) continue;
        ++nt;
        const float fast = rt::rcp_refined(d);
        const float ref = 1.0f / d;
        if (__float_as_uint(fast) != __float_as_uint(ref)) {
            ++nb;
            atomicMin(first_bad, (uint32_t)b);
        }
    }
    atomicAdd(bad, nb);
    atomicAdd(tested, nt);
}

int main() {
    unsigned long long *d_bad, *d_tested, h_bad = 0, h_tested = 0;
    uint32_t *d_first, h_first = 0xffffffffu;
    if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_tested, 8) != hipSuccess || hipMalloc(&d_first, 4) != hipSuccess) return 2;
    (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_tested, 0, 8); (void)hipMemcpy(d_first, &h_first, 4, hipMemcpyHostToDevice);
    k<<<4096, 256>>>(d_bad, d_tested, d_first);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&h_tested, d_tested, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h_first, d_first, 4, hipMemcpyDeviceToHost);
    printf("tested %llu values, %llu differ from 1.0f / d (first: 0x%08x)\n", h_tested, h_bad, h_first);
    return h_bad == 0 ? 0 : 1;
}
