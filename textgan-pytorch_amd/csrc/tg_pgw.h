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

// ---- the split-fp16 fragment GEMM of tg_pgw.hip for other translation units (tg_leak.hip's projection).
// Operands are pieces [row/32][k/16][hi|lo][64 lanes] x 8 halves (tg_pgw.hip, k_to_frag): 32-row blocks padded to whole
// 256-row tiles (tg_fg_rb), 16-wide k blocks in pairs (tg_fg_kb).
int tg_fg_rb(long R);
int tg_fg_kb(long K);
size_t tg_fg_bytes(long R, long K);
// |max| of x as ordered bits, atomically max'ed into *out (zeroed by the caller)
void tg_fg_absmax(hipStream_t s, const float* x, size_t n, unsigned* out);
// X (element (r, k) at X[r*rs + k*cs]) * scal[si] -> pieces at dst (tg_fg_bytes(R, K))
void tg_fg_to_frag(hipStream_t s, const float* X, long rs, long cs, long R, long K, const float* scal, int si, void* dst);
// out[m*ldo + n] = scal[inv_idx] * sum_k A(m,k) B(n,k): A pieces of M rows, B pieces of N rows, K the contraction
int tg_fg_gemm_rows(hipStream_t s, const void* Af, long M, const void* Bf, long N, long K, float* out, long ldo,
                    const float* scal, int inv_idx);

