// Library plumbing (version, per-thread error string) and the reward reduce kernel.
#include <stdarg.h>
#include <stdio.h>

#include "tg_common.h"

static thread_local char g_err[512] = "";

void tg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tg_last_error(void) { return g_err; }
extern "C" const char* tg_version(void) { return "textgan_hip 0.1 gfx950"; }

// torch.mean(rewards.view(B, L, R), dim=-1) on the flat [R*L, B] buffer (utils/rollout.py:153):
// out[bt] = mean_r raw[bt*R + r].  One thread per output, sequential fp32 sum (R <= a few hundred).
__global__ __launch_bounds__(256) void k_reward_reduce(const float* __restrict__ raw, int BL, int R, float* __restrict__ out) {
    const int bt = blockIdx.x * 256 + threadIdx.x;
    if (bt >= BL) return;
    const float* p = raw + (size_t)bt * R;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += p[r];
    out[bt] = s / (float)R;
}

extern "C" int tg_reward_reduce(const float* raw, int B, int L, int R, float* out, void* stream) {
    if (!raw || !out || B < 1 || L < 1 || R < 1) {
        tg_set_error("tg_reward_reduce: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    const int BL = B * L;
    hipLaunchKernelGGL(k_reward_reduce, dim3((BL + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, BL, R, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("k_reward_reduce launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}
