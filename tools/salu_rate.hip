// Microbenchmark: scalar-ALU issue rate of gfx950 next to a VALU stream, as mask-style wave code mixes them
// (every lane predicate is a 64-bit scalar mask handled by s_and / s_andn2 / s_or / s_cmp / s_cbranch).
// Build: hipcc --offload-arch=gfx950 -O3 tools/salu_rate.hip -o /tmp/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 64 SALU per iteration; 1: 64 VALU; 2: 64 VALU + 32 SALU interleaved; 3: 64 VALU + 64 SALU interleaved
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned c) {
    unsigned s0 = c, s1 = c + 1, s2 = c + 2, s3 = c + 3;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const float a = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0 || MODE == 3) {
                asm volatile("s_xor_b32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_xor_b32 %2, %2, %4\n s_add_u32 %3, %3, %4"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(c) : "scc");
            }
            if (MODE == 2) {
                asm volatile("s_xor_b32 %0, %0, %2\n s_add_u32 %1, %1, %2" : "+s"(s0), "+s"(s1) : "s"(c) : "scc");
            }
            if (MODE != 0) {
                asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + (unsigned)(x0 + x1 + x2 + x3);
}

template <int MODE>
void run(const char* name, int waves_per_simd, int valu, int salu) {
    const int cus = 256, blocks = cus * waves_per_simd, iters = 20000;
    unsigned* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 3u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cycles_per_simd = ms * 1e-3 * 2.4e9;
    const double per_simd_iters = (double)waves_per_simd * iters;
    printf("%-26s waves/SIMD=%d  %.3f ms  cycles per SIMD per (wave-iteration of %d VALU + %d SALU): %.1f\n", name, waves_per_simd, ms, valu, salu,
           cycles_per_simd / per_simd_iters);
    hipFree(out);
}

int main() {
    for (int w : {1, 4, 8}) run<0>("salu only", w, 0, 64);
    for (int w : {1, 4, 8}) run<1>("valu only", w, 64, 0);
    for (int w : {1, 4, 8}) run<2>("valu + salu/2", w, 64, 32);
    for (int w : {1, 4, 8}) run<3>("valu + salu", w, 64, 64);
    return 0;
}
