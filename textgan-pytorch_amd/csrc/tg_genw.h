// Host entry points of the wide generator path (tg_genw.hip), called from the C ABI functions in tg_gen.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/textgan_hip.h"

// true when (precision, shape) is served by the step-synchronous split-fp16 GEMM path
bool tg_genw_applies(const TgGenWeights* w);
size_t tg_genw_rollout_workspace_bytes(const TgGenWeights* w, int B, int L, int n_i);
size_t tg_genw_sample_workspace_bytes(const TgGenWeights* w, int N);
// states: [L+1,B,2,H] scratch for the prefix states (tg_gen_teacher_forced's layout); filled by the call itself
int tg_genw_rollout(const TgGenWeights* w, const int64_t* sentences, int B, int L, int g_lo, int g_hi, int i_lo, int i_hi,
                    const float* uniforms, const int64_t* forced, int64_t* completions, float* states,
                    void* workspace, size_t workspace_bytes, hipStream_t stream);
int tg_genw_sample(const TgGenWeights* w, int N, int L, int64_t start_letter, const float* uniforms,
                   const int64_t* forced, int64_t* out, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t tg_genw_nll_workspace_bytes(const TgGenWeights* w, int N);
int tg_genw_nll(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L, float* logp_out,
                float* lse_out, void* workspace, size_t workspace_bytes, hipStream_t stream);
