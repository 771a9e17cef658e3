// Microbenchmark: cycles per 16-element softmax tile update (the VALU part of k_gen's projection loop),
// one wave per SIMD, no memory traffic.  Variants isolate exp / fma / add chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../textgan-pytorch_amd/csrc/tg_common.h"
void tg_set_error(const char*, ...) {}
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed * (r + 1) + threadIdx.x * 1e-3f;
    float m_run = -1e30f, c = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) {  // full softmax tile
            float tmax = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = __builtin_fmaxf(tmax, acc[r]);
            const float m_new = __builtin_fmaxf(m_run, tmax);
            c = c * tg_expf_neg(m_run - m_new);
            float ev[16];
            tg_exp16_neg(acc, m_new, ev);
            float ts = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ts += ev[r];
            c += ts; m_run = m_new;
        } else if (VAR == 1) {  // 16 bare v_exp
            float ts = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ts += __builtin_amdgcn_exp2f(acc[r]);
            c += ts;
        } else if (VAR == 2) {  // 96 independent fmas
#pragma unroll
            for (int k2 = 0; k2 < 6; ++k2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(acc[r], 1.0001f, 1e-7f);
        } else {  // 96 dependent fmas
#pragma unroll
            for (int k2 = 0; k2 < 96; ++k2) c = __builtin_fmaf(c, 1.0001f, seed);
        }
        // keep the inputs changing so nothing is hoisted
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += c * 1e-9f;
    }
    long long t1 = clock64();
    float s = c + m_run;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[VAR] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.37f);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.37f);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.37f);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.37f);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    const char* nm[4] = {"softmax tile (16 exp, ~105 VALU)", "16 v_exp + 16 add (+16 fma)", "96 indep fma (+16 fma)", "96 dep fma (+16 fma)"};
    for (int i = 0; i < 4; ++i) printf("%-36s %.0f cycles/iter\n", nm[i], (double)h[i] / iters);
    return 0;
}
