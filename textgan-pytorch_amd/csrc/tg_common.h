// Shared device helpers for the textgan HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

// Same for x <= 0 with no range guards (7 VALU per call incl. the caller's subtract): the inner
// softmax loop.  Very negative x (down to TG_NEG_BIG - anything finite) gives exactly 0, no NaN.
#define TG_NEG_BIG (-1.0e30f)
__device__ __forceinline__ float tg_expf_neg(float x) {
    const float L2E_HI = 1.44269502162933349609375f;
    const float L2E_LO = 1.92596299112661746e-8f;
    const float hi = x * L2E_HI;
    float lo = __builtin_fmaf(x, L2E_HI, -hi);
    lo = __builtin_fmaf(x, L2E_LO, lo);
    const float r = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(r, lo * 0.693147182464599609375f, r);
}

// 16 of them at once, staged so that the element chains are independent (ILP) and adjacent pairs map
// onto v_pk_mul_f32 / v_pk_fma_f32: out[r] = exp(x[r] - m), x[r] <= m.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tg_exp16_neg(const f32x16& x, float m, float (&out)[16]) {
    const f32x2 L2E_HI = {1.44269502162933349609375f, 1.44269502162933349609375f};
    const f32x2 L2E_LO = {1.92596299112661746e-8f, 1.92596299112661746e-8f};
    const f32x2 LN2 = {0.693147182464599609375f, 0.693147182464599609375f};
    const f32x2 m2 = {m, m};
    f32x2 d[8], hi[8], lo[8], e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const f32x2 v = {x[2 * i], x[2 * i + 1]}; d[i] = v - m2; }
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = d[i] * L2E_HI;
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = __builtin_elementwise_fma(d[i], L2E_HI, -hi[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = __builtin_elementwise_fma(d[i], L2E_LO, lo[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i].x = __builtin_amdgcn_exp2f(hi[i].x); e[i].y = __builtin_amdgcn_exp2f(hi[i].y); }
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_elementwise_fma(e[i], lo[i] * LN2, e[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) { out[2 * i] = e[i].x; out[2 * i + 1] = e[i].y; }
}

__device__ __forceinline__ float tg_sigmoid(float x) { return 1.0f / (1.0f + tg_expf(-x)); }

__device__ __forceinline__ float tg_tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = tg_expf(-2.0f * ax);
    const float r = (1.0f - t) / (1.0f + t);
    return __builtin_copysignf(r, x);
}

// Counter-based dropout mask (TgDisWeights.dropout_seed): element idx keeps iff tg_hash32(idx + seed*golden) < keep*2^32.
// "lowbias32" integer finaliser: 5 ALU ops, no state, any element in any order on any rank.
__host__ __device__ __forceinline__ uint32_t tg_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ bool tg_drop_keep(uint32_t idx, uint32_t seed, uint32_t keep_u32) {
    return tg_hash32(idx + seed * 0x9E3779B9U) < keep_u32;
}

// Phase-cycle diagnostics (tools/ubench/stamps.py, conv_stamps.py) exist only in a `make DIAG=1` build: the
// shipped library reads no clock and takes no pointer from the environment.
#ifndef TG_DIAG
#define TG_DIAG 0
#endif
__device__ __forceinline__ long long tg_clock() {
#if TG_DIAG
    return clock64();
#else
    return 0;
#endif
}
inline long long* tg_diag_pointer(const char* env_name) {
#if TG_DIAG
    const char* d = getenv(env_name);
    return d ? (long long*)strtoull(d, nullptr, 0) : nullptr;
#else
    (void)env_name;
    return nullptr;
#endif
}

__device__ __forceinline__ int tg_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int tg_max(int a, int b) { return a > b ? a : b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the device that is current when it is called: one
// record per kernel instantiation (a function-local static TgLdsAttr) remembers, per device, the largest size set,
// so a second GPU used from the same thread gets its attribute too.  Setting it twice from two threads is harmless.
void tg_set_error(const char* fmt, ...);
struct TgLdsAttr { std::atomic<size_t> set[64]; };
inline int tg_lds_attr(TgLdsAttr& st, const void* fn, size_t bytes, const char* what) {
    if (bytes <= 48 * 1024) return TG_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        tg_set_error("%s: hipGetDevice failed or device index out of range", what);
        return TG_ERR_LAUNCH;
    }
    if (bytes <= st.set[dev].load(std::memory_order_relaxed)) return TG_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        tg_set_error("hipFuncSetAttribute(%s, %zu bytes of LDS): %s", what, bytes, hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    st.set[dev].store(bytes, std::memory_order_relaxed);
    return TG_OK;
}

// host-side error plumbing and per-kernel event timing (tg_api.hip)
void tg_prof_begin(int kernel_id, hipStream_t s);
void tg_prof_end(hipStream_t s);
