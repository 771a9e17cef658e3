// Host entry points of the GEMM-formulated generator forward / policy-gradient backward (tg_pgw.hip), called from the
// C ABI functions in tg_gen.hip and tg_pg.hip for shapes beyond what their on-chip kernels cover.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/textgan_hip.h"

bool tg_pgw_shape_ok(const TgGenWeights* w);   // E, H multiples of 32 up to 1024
// split: the vocabulary products (logits, dlogits . W_out, dlogits^T . h) as split-fp16 GEMMs on fragment-ordered operands
// (TG_GEN_SPLIT_F16 of the policy-gradient entry points); false = exact fp32 MFMA throughout
size_t tg_pgw_forward_workspace_bytes(const TgGenWeights* w, int N, int L, bool have_logits, bool have_states,
                                      bool split = false);
int tg_pgw_forward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L, float* logp_out,
                   float* lse_out, float* logits_out, float* states_out, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, const float* states_in = nullptr,   // states_in [N,2,H]: initial (h,c), null = zeros
                   bool split = false);
size_t tg_pgw_backward_workspace_bytes(const TgGenWeights* w, int N, int L, bool split = false);
int tg_pgw_backward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, const float* coeff,
                    const float* lse, const float* states, int N, int L, int64_t padding_idx, const TgGenGrads* grads,
                    void* workspace, size_t workspace_bytes, hipStream_t stream, bool split = false);
