// Internal entry of the scorer (tg_dis.hip) used by the LeakGAN rollout's position loop (tg_leak.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/textgan_hip.h"

// CNNDiscriminator.get_feature over [N,L] sequences -> feature_out [N,F] (16-byte aligned).
// Positions t >= valid_len are read as token 0 (the padding the reference feeds there) whatever seqs holds: the caller
// may be writing later positions on another stream meanwhile.
// Workspaces: weights_ws (tg_dis_features_weights_bytes) holds the prepared (scaled, split) weights and is shared by any
// number of per-call workspaces (tg_dis_features_call_bytes each, laid out for N_layout >= N sequences, so that calls
// with different N share one layout).  tg_dis_features_prepare fills weights_ws and readies the n_ws call workspaces;
// it is needed once per (w, N_layout), and every tg_dis_features call must be ordered behind it.
size_t tg_dis_features_weights_bytes(const TgDisWeights* w, int L);
size_t tg_dis_features_call_bytes(const TgDisWeights* w, int N_layout, int L);
int tg_dis_features_prepare(const TgDisWeights* w, int N_layout, int L, float* feature_out, void* const* call_ws, int n_ws,
                            size_t call_ws_bytes, void* weights_ws, hipStream_t stream);
int tg_dis_features(const TgDisWeights* w, const int64_t* seqs, int N, int N_layout, int L, int valid_len,
                    float* feature_out, void* call_ws, size_t call_ws_bytes, void* weights_ws, hipStream_t stream);
