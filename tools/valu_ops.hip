// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD at 8 waves/SIMD) of the VALU instruction kinds the
// kernels use besides plain f32 mul/add: selects, compares, 32-bit integer ops, 32-bit integer multiply, rcp, sqrt.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_ops.hip -o tools/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, unsigned ua, unsigned ub) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // f32 mul/add
                x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
            } else if (MODE == 1) {  // integer xor with a shifted copy (2 instructions each; chains the compiler cannot fold)
                u0 ^= u0 >> 15; u1 ^= u1 >> 13; u2 ^= u2 << 7; u3 ^= u3 >> 11; u4 ^= u4 << 5; u5 ^= u5 >> 9; u6 ^= u6 << 3; u7 ^= u7 >> 17;
            } else if (MODE == 2) {  // 32-bit integer multiply by an odd constant + xor-shift (3 instructions each)
                u0 = u0 * ua; u0 ^= u0 >> 15; u1 = u1 * ub; u1 ^= u1 >> 15; u2 = u2 * ua; u2 ^= u2 >> 15; u3 = u3 * ub; u3 ^= u3 >> 15;
                u4 = u4 * ua; u4 ^= u4 >> 15; u5 = u5 * ub; u5 ^= u5 >> 15; u6 = u6 * ua; u6 ^= u6 >> 15; u7 = u7 * ub; u7 ^= u7 >> 15;
            } else if (MODE == 3) {  // v_max / v_min against a moving operand
                x0 = fmaxf(x0, x1 * a); x1 = fminf(x1, x2 * b); x2 = fmaxf(x2, x3 * a); x3 = fminf(x3, x4 * b);
                x4 = fmaxf(x4, x5 * a); x5 = fminf(x5, x6 * b); x6 = fmaxf(x6, x7 * a); x7 = fminf(x7, x0 * b);
            } else if (MODE == 4) {  // rcp
                x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
            } else if (MODE == 5) {  // compare + select (2 instructions each)
                x0 = x0 < a ? x1 : x0; x1 = x1 < b ? x2 : x1; x2 = x2 < a ? x3 : x2; x3 = x3 < b ? x4 : x3;
                x4 = x4 < a ? x5 : x4; x5 = x5 < b ? x6 : x5; x6 = x6 < a ? x7 : x6; x7 = x7 < b ? x0 : x7;
            } else if (MODE == 6) {  // IEEE division
                x0 = x0 / a; x1 = x1 / b; x2 = x2 / a; x3 = x3 / b; x4 = x4 / a; x5 = x5 / b; x6 = x6 / a; x7 = x7 / b;
            } else if (MODE == 7) {  // IEEE sqrt
                x0 = __builtin_sqrtf(x0); x1 = __builtin_sqrtf(x1); x2 = __builtin_sqrtf(x2); x3 = __builtin_sqrtf(x3);
                x4 = __builtin_sqrtf(x4); x5 = __builtin_sqrtf(x5); x6 = __builtin_sqrtf(x6); x7 = __builtin_sqrtf(x7);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int MODE>
void run(const char* name, int ops_per_iter) {
    const int dev_cus = 256, waves_per_simd = 8;
    const int blocks = dev_cus * waves_per_simd;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 50, 1.0001f, 0.5f, 0x7feb352du, 0x846ca68bu);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f, 0x7feb352du, 0x846ca68bu);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * ops_per_iter;  // source-level operations per wave
    const double per_simd_per_s = ops / (ms * 1e-3) / (dev_cus * 4);
    printf("%-32s %.3f ms  %.2f cycles per operation per SIMD at 2.4 GHz\n", name, ms, 2.4e9 / per_simd_per_s);
    hipFree(out);
}

int main() {
    run<0>("f32 mul/add", 64);
    run<1>("int shift + xor (2 inst)", 64);
    run<2>("int mul32 + shift + xor (3 inst)", 64);
    run<3>("f32 mul + max/min (2 inst)", 64);
    run<4>("rcp", 64);
    run<5>("compare+select (2 inst)", 64);
    run<6>("IEEE f32 division", 64);
    run<7>("IEEE f32 sqrt", 64);
    return 0;
}
