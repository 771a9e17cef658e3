// Internal entry of the scorer (tg_dis.hip) used by the LeakGAN rollout's position loop (tg_leak.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/textgan_hip.h"

// CNNDiscriminator.get_feature over [N,L] sequences -> feature_out [N,F]; workspace of tg_dis_workspace_bytes(w,N,L).
// Positions t >= valid_len are read as token 0 (the padding the reference feeds there) whatever seqs holds: the caller
// may be writing later positions on another stream meanwhile.
// The workspace is laid out for N_layout >= N sequences (tg_dis_workspace_bytes(w, N_layout, L)): calls with different
// N share one layout, and with it the prepared weights.
// reuse = the workspace already holds the prepared (scaled, split) weights a previous call wrote for the same w and
// N_layout.
int tg_dis_features(const TgDisWeights* w, const int64_t* seqs, int N, int N_layout, int L, int valid_len,
                    float* feature_out, void* workspace, size_t workspace_bytes, bool reuse, hipStream_t stream);
