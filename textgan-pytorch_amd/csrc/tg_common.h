// Shared device helpers for the textgan HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/textgan_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A lane l -> A[i=l&31][k=l>>5]; B lane l -> B[k=l>>5][j=l&31];
// D lane l, reg r -> D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].  Exact fp32 fma chain, k=0 then k=1.
#define TG_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define TG_NEG_INF (-__builtin_inff())

// expf with ~1.5 ulp error: exp2 on the split product x*log2(e) (hi + lo), the lo part
// applied as a first-order correction.  v_exp_f32 alone would carry |x|*6e-8 relative error.
__device__ __forceinline__ float tg_expf(float x) {
    const float L2E_HI = 1.44269502162933349609375f;   // fp32(log2 e)
    const float L2E_LO = 1.92596299112661746e-8f;      // log2 e - L2E_HI
    const float xc = __builtin_fminf(x, 88.0f);  // keeps exp finite (sigmoid of very negative inputs)
    const float hi = xc * L2E_HI;
    const float lo = __builtin_fmaf(xc, L2E_HI, -hi) + xc * L2E_LO;
    const float r = __builtin_amdgcn_exp2f(hi);
    const float v = __builtin_fmaf(r, lo * 0.693147182464599609375f, r);
    return (x < -104.0f) ? 0.0f : v;  // also maps -inf (masked logits) to exactly 0
}

__device__ __forceinline__ float tg_sigmoid(float x) { return 1.0f / (1.0f + tg_expf(-x)); }

__device__ __forceinline__ float tg_tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = tg_expf(-2.0f * ax);
    const float r = (1.0f - t) / (1.0f + t);
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ int tg_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int tg_max(int a, int b) { return a > b ? a : b; }

// host-side error plumbing and per-kernel event timing (tg_api.hip)
void tg_set_error(const char* fmt, ...);
void tg_prof_begin(int kernel_id, hipStream_t s);
void tg_prof_end(hipStream_t s);
