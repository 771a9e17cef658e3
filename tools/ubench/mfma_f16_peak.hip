// Microbenchmark: the SUSTAINED rate of v_mfma_f32_32x32x16_f16 (the instruction of the split-fp16 kernels) on the
// whole chip, registers only -- no LDS, no global loads -- with pseudo-random operands (zeros toggle no wires and
// flatter the power draw).  Long launches (tens of ms) so the clock settles where the power limit puts it.
// hipcc --offload-arch=gfx950 -O3 mfma_f16_peak.hip -o mfma_f16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed, int zero) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[4], b[4];
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 8; ++e) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            a[q][e] = zero ? (_Float16)0.f : (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 32768.f));
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            b[q][e] = zero ? (_Float16)0.f : (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 32768.f));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + j) & 3], b[j], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same with v_mfma_f32_16x16x32_f16 (16 384 flop each, 16 cycles): MI355X_MICROARCH.md reports it sustains a higher
// clock under the power limit.  16 accumulators of 4 registers = the same 64 accumulator registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned seed, int zero) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    f16x8 a[4], b[4];
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 8; ++e) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            a[q][e] = zero ? (_Float16)0.f : (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 32768.f));
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            b[q][e] = zero ? (_Float16)0.f : (_Float16)(((int)(x & 0xffff) - 32768) * (1.f / 32768.f));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + j) & 3], b[(i >> 2) ^ j], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero <= 1; ++zero)
        for (int wpe = 1; wpe <= 2; ++wpe) {
            const int grid = 256 * wpe, iters = 40000;  // 4 waves per workgroup = one per SIMD
            float ms = 0.f;
            for (int rep = 0; rep < 4; ++rep) {  // the last of four back-to-back launches is the settled one
                hipEventRecord(e0);
                hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u + rep, zero);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                const double mf = (double)grid * 4 * iters * 16;  // MFMAs (4 j x 4 acc per iteration)
                printf("%s operands, %d wave(s)/SIMD, launch %d: %.1f ms, %.0f TFLOP/s (32x32x16 f16: 32768 flop each), "
                       "%.1f cycles/MFMA/SIMD at 2.4 GHz\n", zero ? "zero" : "random", wpe, rep, ms,
                       mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
            }
        }
    for (int wpe = 1; wpe <= 2; ++wpe) {
        const int grid = 256 * wpe, iters = 40000;
        float ms = 0.f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, iters, 777u + rep, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)grid * 4 * iters * 32;  // MFMAs (2 j x 16 acc per iteration)
            printf("random operands, %d wave(s)/SIMD, launch %d: %.1f ms, %.0f TFLOP/s (16x16x32 f16: 16384 flop each), "
                   "%.1f cycles/MFMA/SIMD at 2.4 GHz\n", wpe, rep, ms, mf * 16384 / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / (mf / 1024));
        }
    }
    return 0;
}
