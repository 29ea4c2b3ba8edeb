// Microbenchmark: sustained wave64 VALU issue rate on gfx950 for plain (non-fused) f32 mul/add streams,
// the instruction mix that bit-parity with an FMA-less CPU build forces on the traversal kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // mul, add alternating (independent chains)
                x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
            } else if (MODE == 1) {  // fma
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else {  // v_cmp + s_and style mask work mixed with valu
                x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b;
                unsigned long long m = __builtin_amdgcn_ballot_w64(x0 < x1) & __builtin_amdgcn_ballot_w64(x2 < x3);
                if (m == 0x123456789ull) x4 += 1.0f;
                x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int valu_per_iter) {
    int dev_cus = 256;
    int blocks = dev_cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * valu_per_iter;  // wave-instructions
    double per_simd_per_s = insts / (ms * 1e-3) / (dev_cus * 4);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f G wave-inst/s/SIMD  (=%.2f cycles/inst at 2.4 GHz)  %.1f T lane-ops/s\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, insts * 64 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) run<0>("mul/add independent", w, 64);
    for (int w : {1, 2, 4, 8}) run<1>("fma independent", w, 64);
    for (int w : {2, 8}) run<2>("mul/add + 2 v_cmp + salu", w, 64 + 16);
    return 0;
}
