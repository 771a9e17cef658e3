// K-DIS: the CNN discriminator scorer (gfx950).
//
//   k_dis_conv     embedding gather -> LDS; all filter widths as implicit GEMMs on
//                  v_mfma_f32_32x32x2_f32 (M = filters, N = (sequence, time), K = f*Ed);
//                  bias + ReLU + max-over-time in registers; pooled [N,F] out.
//   k_dis_highway  h = pooled . W_hw^T as a 128x128-tiled fp32 MFMA GEMM with the sequences
//                  on the lanes, fused sigmoid/relu gate, dropout mask and the partial
//                  feature2out dot; partial logits [N, chunks, n_out] out.
//   k_dis_final    fixed-order sum of the partials + bias, softmax column -> reward.
//
// Replaces models/discriminator.py:40-64 and utils/rollout.py:146-148 of the reference.
// Embedding dim Ed = 64 (config.py:113 of the reference) is compiled in.
#include <stdlib.h>

#include <type_traits>

#include "tg_common.h"

namespace {

constexpr int kEd = 64;
constexpr int kERow = 68;          // LDS floats per embedded token: 16-B aligned, slot = row index mod 16
constexpr int kConvThreads = 512;  // 8 waves
constexpr int kMaxItems = 192;
constexpr int kMaxOut = 8;

struct ConvItem {
    uint8_t k;   // filter width index
    uint8_t mt;  // 32-filter tile of that width
};

// k_dis_conv modes
constexpr int kConvPlain = 0;   // any [N,L] sequences: all windows, pooled out
constexpr int kConvStoreAll = 1;// base sentences: every window's conv+bias value out (cv [N,L,F])
constexpr int kConvPrefix = 2;  // rollout completions [R,L,B,L]: sequence n = (i, g-1, b) shares its first g tokens
                                // with sentence b, so windows inside the prefix come from the running-max table

constexpr int kMaxMerged = 4;
// A mixed-width tile (split path): the banks' remainder rows (n_k mod 32) packed 32 to a tile.  Row i carries filter
// j[i] of bank k[i]; the tile runs fmax window rows with zero weights past each row's own width, over the positions
// of its narrowest member.  Per-row epilogue data lives in device memory (k_split_weights writes it).
struct MergedMeta {
    float bias[32];
    int feat[32];  // global feature index, -1 = unused row
    int fw[32];    // the row's filter width
};
struct MergedTile {
    uint8_t nrow, fmax, fmin, pad;
    uint8_t k[32];
    uint8_t j[32];  // filter index inside its bank (remainder rows start at (n_k/32)*32 <= 224)
};

struct ConvArgs {
    TgDisWeights w;
    const int64_t* seqs;  // [N,L]
    float* pooled;        // [N,F]
    float* cv;            // StoreAll: [N,L,F] out; Prefix: running max over time of the sentences' windows, in
    int B;                // Prefix: sentences per (i, g) block
    int N, L;
    int S, log2S;         // sequences per workgroup (power of two <= 32), TT = 32 / S
    int n_items;
    int foff[TG_MAX_FILTERS];  // feature offset of each width
    // split-fp16 path: weights pre-split into scaled (hi, lo) halves, bank k's hi part at w16 + woff[k],
    // its lo part n_k*f_k*64 halves later; scales = {scale_w, scale_e, 1/(scale_w*scale_e)} on the device
    const _Float16* w16;
    const float* scales;
    unsigned* pooled_absmax;   // nullable: max of the pooled features as ordered bits (they are >= 0), for the highway's scale
    int woff[TG_MAX_FILTERS];
    int n_merged;              // mixed-width remainder tiles (items with k == 0xFF, mt = tile index); split path only
    int pad_rows;              // zero rows kept after the sequence in LDS: max (fmax - fmin) over the merged tiles
    int mwoff[kMaxMerged];     // halves offset of a merged tile's fragments in w16
    uint8_t mfmax[kMaxMerged], mfmin[kMaxMerged];
    const MergedMeta* mmeta;   // [n_merged]
    ConvItem items[kMaxItems]; // sorted by cost, heaviest first
    long long* stamps;         // diagnostic (TG_CONV_STAMPS env = device pointer): per workgroup [16] clock64 values
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kERowH = 2 * kERow;  // LDS halves per embedded token in the split path: [64 hi | 64 lo | 8 pad]
#define TG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// SPLIT = false: exact fp32 on v_mfma_f32_32x32x2_f32.
// SPLIT = true:  x = hi + lo with hi = fp16(x*scale), lo = fp16(x*scale - hi) (22 mantissa bits; scale a power of
//                two that puts the tensor's max near 2^14); w.e ~ w_hi.e_hi + w_hi.e_lo + w_lo.e_hi on
//                v_mfma_f32_32x32x16_f16 with fp32 accumulation: ~5e-7 relative per product at 3/16 of the
//                fp32-MFMA cycles.
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(kConvThreads) void k_dis_conv(const ConvArgs a) {
    const long long st0 = tg_clock();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ebuf = reinterpret_cast<float*>(smem);  // [L][S][68] floats, or [L][S][136] halves (same bytes)
    _Float16* ebuf_h = reinterpret_cast<_Float16*>(smem);
    const int L = a.L, S = a.S, log2S = a.log2S, TT = 32 >> log2S;
    int* ctr = reinterpret_cast<int*>(ebuf + (size_t)(L + a.pad_rows) * S * kERow);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = lane & 31;
    const int hh = lane >> 5;
    const int seq0 = blockIdx.x * S;

    // ---- gather: token embeddings -> LDS, time-major [t][s][68].  Eight 16-B pieces per thread and pass, all token
    // reads first and all embedding reads next: two memory round trips per pass instead of two per piece.
    {
        constexpr int kGB = 8;
        const int total = L * S * (kEd / 4);
        const float sc = SPLIT ? a.scales[1] : 1.f;
        for (int base = tid; base < total; base += kConvThreads * kGB) {
            int tok[kGB];
            f32x4 v[kGB];
#pragma unroll
            for (int j = 0; j < kGB; ++j) {
                const int idx = tg_min(base + kConvThreads * j, total - 1);
                const int ts = idx >> 4;          // t*S + s
                const int sq = ts & (S - 1), t = ts >> log2S;
                const int seq = tg_min(seq0 + sq, a.N - 1);
                tok[j] = (int)a.seqs[(size_t)seq * L + t];
            }
#pragma unroll
            for (int j = 0; j < kGB; ++j) {
                const int c = (base + kConvThreads * j) & 15;  // 16-B chunk of the 256-B row
                const int tk = tg_min(tg_max(tok[j], 0), a.w.V - 1);
                v[j] = *reinterpret_cast<const f32x4*>(a.w.emb + (size_t)tk * kEd + 4 * c);
            }
#pragma unroll
            for (int j = 0; j < kGB; ++j) {
                const int idx = base + kConvThreads * j;
                if (idx >= total) continue;
                const int c = idx & 15, ts = idx >> 4;
                if (!SPLIT) {
                    *reinterpret_cast<f32x4*>(ebuf + (size_t)ts * kERow + 4 * c) = v[j];
                } else {
                    f16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[j][e] * sc;
                        h[e] = (_Float16)x;
                        l[e] = (_Float16)(x - (float)h[e]);
                    }
                    *reinterpret_cast<f16x4*>(ebuf_h + (size_t)ts * kERowH + 4 * c) = h;
                    *reinterpret_cast<f16x4*>(ebuf_h + (size_t)ts * kERowH + kEd + 4 * c) = l;
                }
            }
        }
    }
    // zero rows behind the sequence: a merged tile's wider members read past position L - 1 against zero weights
    for (int idx = tid; idx < a.pad_rows * S * (kERow / 4); idx += kConvThreads)
        *reinterpret_cast<f32x4*>(ebuf + (size_t)L * S * kERow + 4 * idx) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid == 0) *ctr = 0;
    __syncthreads();

    const int s = n & (S - 1);   // this lane's sequence within the workgroup
    const int tsub = n >> log2S; // and its time sub-position within a tile
    // Prefix mode: given_num of this lane's sequence, and the smallest one in the workgroup
    int g_lane = 0, g_min = 0, b_lane = 0;
    if (MODE == kConvPrefix) {
        const int seq = tg_min(seq0 + s, a.N - 1);
        g_lane = (seq / a.B) % L + 1;
        b_lane = seq % a.B;
        g_min = g_lane;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) g_min = tg_min(g_min, __shfl_xor(g_min, off, 64));
        g_min = __builtin_amdgcn_readfirstlane(g_min);
    }

    float pmax = 0.f;  // largest pooled value this lane stored
    const long long st1 = tg_clock();
    long long st_mfma = 0;
    int st_items = 0;
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(ctr, 1);
        // gridDim.y > 1 (few sequences: the sentence pass): the items are dealt round-robin to the y workgroups
        item = __builtin_amdgcn_readfirstlane(item) * (int)gridDim.y + (int)blockIdx.y;
        if (item >= a.n_items) break;
        const long long st_item = tg_clock();
        const int k = a.items[item].k, mt = a.items[item].mt;
        const bool merged = SPLIT && k == 0xFF;  // mt = merged tile index
        const int f = merged ? a.mfmax[mt] : a.w.f[k];
        const int nk = merged ? 32 : a.w.n[k];
        const int K = f * kEd, T = L - (merged ? a.mfmin[mt] : f) + 1;
        const int arow = tg_min(mt * 32 + n, nk - 1);
        const float* wp = merged ? nullptr : a.w.conv_w[k] + (size_t)arow * K + 32 * hh;
        const MergedMeta* mm = merged ? a.mmeta + mt : nullptr;
        auto mrow = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };  // tile row of accumulator register r
        // windows t <= g - f lie inside the shared prefix (Prefix mode): tiles start at the first fresh window of the
        // workgroup's smallest given_num, not at a multiple of the tile's time span
        const int t_org = (MODE == kConvPrefix) ? tg_max(0, g_min - f + 1) : 0;
        const int ntile = tg_max(0, T - t_org + TT - 1) >> (5 - log2S);  // ceil((T - t_org) / TT), 0 if none is fresh

        float runmax[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) runmax[r] = TG_NEG_INF;

        f32x16 acc[4];
        int boff[4];
        auto start_chunk = [&](int nb0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
                const int t = tg_min(t_org + (nb0 + nt) * TT + tsub, T - 1);
                boff[nt] = SPLIT ? (t * S + s) * kERowH + 8 * hh : (t * S + s) * kERow + 32 * hh;
            }
        };
        auto finish_chunk = [&](int nb0, int nbc) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int tw = t_org + (nb0 + nt) * TT + tsub;
                const bool ok = (nt < nbc) && (tw < T);
                if (merged) {  // per-row width: position tw exists for row r iff tw + fw <= L
                    if (MODE == kConvStoreAll) {
                        if (ok && seq0 + s < a.N) {
                            float* po = a.cv + ((size_t)(seq0 + s) * L + tw) * a.w.F;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int ft = mm->feat[mrow(r)];
                                if (ft >= 0 && tw + mm->fw[mrow(r)] <= L) po[ft] = acc[nt][r] + mm->bias[mrow(r)];
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            runmax[r] = __builtin_fmaxf(runmax[r], (ok && tw + mm->fw[mrow(r)] <= L) ? acc[nt][r] : TG_NEG_INF);
                    }
                } else if (MODE == kConvStoreAll) {
                    if (ok && seq0 + s < a.N) {
                        float* po = a.cv + ((size_t)(seq0 + s) * L + tw) * a.w.F + a.foff[k];
                        const float* pb = a.w.conv_b[k];
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int m0 = mt * 32 + 8 * rq + 4 * hh;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (m0 + e < nk) po[m0 + e] = acc[nt][4 * rq + e] + pb[m0 + e];
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) runmax[r] = __builtin_fmaxf(runmax[r], ok ? acc[nt][r] : TG_NEG_INF);
                }
            }
        };

        const int nb_start = 0;
        if constexpr (SPLIT) {
            // A fragments of (k, mt) in load order (k_split_weights): [tau][8 pieces][64 lanes] x 16 B, pieces 0-3 the
            // hi halves of k blocks 0-3, pieces 4-7 their lo halves.  One flat loop over (chunk of 4 N-tiles, tau)
            // with the next stage's fragment in flight in a second register set.
            const f16x8* wfrag = merged ? reinterpret_cast<const f16x8*>(a.w16 + a.mwoff[mt]) + lane
                                        : reinterpret_cast<const f16x8*>(a.w16 + a.woff[k]) + ((size_t)mt * f * 8) * 64 + lane;
            const float inv = a.scales[2];
            const int total = ((ntile - nb_start + 3) >> 2) * f;
            int tau = 0, nb0 = nb_start;
            f16x8 A0[8], A1[8];
            auto load_a = [&](f16x8 (&A)[8], int ta) {
                const f16x8* pa = wfrag + (size_t)ta * 8 * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) A[i] = pa[i * 64];
            };
            // the MFMA block of one stage for a compile-time tile count: straight-line code, so the LDS reads of later
            // tiles are issued under the MFMAs of earlier ones (a run-time "nt < nbc" guard puts every tile in a basic
            // block of its own, each waiting out its LDS latency)
            auto mfma_block = [&](const f16x8 (&cur)[8], int toff, auto nbc_tag) {
                constexpr int NBC = decltype(nbc_tag)::value;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int nt = 0; nt < NBC; ++nt) {
                        const f16x8 bh = *reinterpret_cast<const f16x8*>(ebuf_h + boff[nt] + toff + 16 * ks);
                        const f16x8 bl = *reinterpret_cast<const f16x8*>(ebuf_h + boff[nt] + toff + kEd + 16 * ks);
                        acc[nt] = TG_MFMA16(cur[ks], bh, acc[nt]);
                        acc[nt] = TG_MFMA16(cur[ks], bl, acc[nt]);
                        acc[nt] = TG_MFMA16(cur[4 + ks], bh, acc[nt]);
                    }
                }
            };
            auto stage = [&](const f16x8 (&cur)[8], f16x8 (&nxt)[8], bool more) {
                const int nbc = tg_min(4, ntile - nb0);
                if (tau == 0) start_chunk(nb0);
                const int tau_n = (tau + 1 == f) ? 0 : tau + 1;
                if (more) load_a(nxt, tau_n);
                const int toff = tau * S * kERowH;
                switch (nbc) {
                    case 4: mfma_block(cur, toff, std::integral_constant<int, 4>{}); break;
                    case 3: mfma_block(cur, toff, std::integral_constant<int, 3>{}); break;
                    case 2: mfma_block(cur, toff, std::integral_constant<int, 2>{}); break;
                    default: mfma_block(cur, toff, std::integral_constant<int, 1>{}); break;
                }
                if (tau_n == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nt][r] *= inv;
                    finish_chunk(nb0, nbc);
                    nb0 += 4;
                }
                tau = tau_n;
            };
            if (total > 0) load_a(A0, 0);
            for (int it = 0; it < total; it += 2) {
                stage(A0, A1, it + 1 < total);
                if (it + 1 < total) stage(A1, A0, it + 2 < total);
            }
        } else {
            for (int nb0 = nb_start; nb0 < ntile; nb0 += 4) {
                const int nbc = tg_min(4, ntile - nb0);
                start_chunk(nb0);
                auto mfma_block = [&](const f32x4 (&av)[8], int toff, auto nbc_tag) {
                    constexpr int NBC = decltype(nbc_tag)::value;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                        for (int nt = 0; nt < NBC; ++nt) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(ebuf + boff[nt] + toff + 4 * jj);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[nt] = TG_MFMA32(av[jj][e], bv[e], acc[nt]);
                        }
                    }
                };
                for (int tau = 0; tau < f; ++tau) {
                    f32x4 av[8];
                    const f32x4* pa = reinterpret_cast<const f32x4*>(wp + tau * kEd);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) av[jj] = pa[jj];
                    const int toff = tau * S * kERow;
                    switch (nbc) {
                        case 4: mfma_block(av, toff, std::integral_constant<int, 4>{}); break;
                        case 3: mfma_block(av, toff, std::integral_constant<int, 3>{}); break;
                        case 2: mfma_block(av, toff, std::integral_constant<int, 2>{}); break;
                        default: mfma_block(av, toff, std::integral_constant<int, 1>{}); break;
                    }
                }
                finish_chunk(nb0, nbc);
            }
        }
        if (a.stamps) { st_mfma += tg_clock() - st_item; ++st_items; }
        if (MODE == kConvStoreAll) continue;
        // combine the TT time sub-positions of a sequence (lanes n, n+S, n+2S, ...)
        for (int off = S; off < 32; off <<= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) runmax[r] = __builtin_fmaxf(runmax[r], __shfl_xor(runmax[r], off, 64));
        }
        if (merged) {
            if (tsub == 0 && seq0 + s < a.N) {
                float* po = a.pooled + (size_t)(seq0 + s) * a.w.F;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ft = mm->feat[mrow(r)], fwr = mm->fw[mrow(r)];
                    if (ft < 0) continue;
                    float o = __builtin_fmaxf(runmax[r] + mm->bias[mrow(r)], 0.f);
                    if (MODE == kConvPrefix && g_lane >= fwr)
                        o = __builtin_fmaxf(o, a.cv[((size_t)b_lane * L + (g_lane - fwr)) * a.w.F + ft]);
                    po[ft] = o;
                    pmax = __builtin_fmaxf(pmax, o);
                }
            }
            continue;
        }
        // bias + ReLU, store: D reg r -> filter mt*32 + (r&3) + 8*(r>>2) + 4*hh
        if (tsub == 0 && seq0 + s < a.N) {
            float* po = a.pooled + (size_t)(seq0 + s) * a.w.F + a.foff[k];
            const float* pb = a.w.conv_b[k];
            // Prefix mode: max over the windows t <= g - f (bias included) of the base sentence
            const float* pm = (MODE == kConvPrefix && g_lane >= f)
                                  ? a.cv + ((size_t)b_lane * L + (g_lane - f)) * a.w.F + a.foff[k] : nullptr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int m0 = mt * 32 + 8 * rq + 4 * hh;
                if (m0 + 4 <= nk && (((a.foff[k] + m0) | a.w.F) & 3) == 0) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaxf(runmax[4 * rq + e] + pb[m0 + e], 0.f);
                    if (pm) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pm + m0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaxf(o[e], pv[e]);
                    }
                    *reinterpret_cast<f32x4*>(po + m0) = o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pmax = __builtin_fmaxf(pmax, o[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m0 + e < nk) {
                            float o = __builtin_fmaxf(runmax[4 * rq + e] + pb[m0 + e], 0.f);
                            if (pm) o = __builtin_fmaxf(o, pm[m0 + e]);
                            po[m0 + e] = o;
                            pmax = __builtin_fmaxf(pmax, o);
                        }
                }
            }
        }
    }
    if (a.stamps && lane == 0) {
        const long long st2 = tg_clock();
        long long* o = a.stamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 4;
        o[0] = st1 - st0; o[1] = st2 - st1; o[2] = st_mfma; o[3] = st_items;
    }
    if (MODE != kConvStoreAll && a.pooled_absmax) {  // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pmax = __builtin_fmaxf(pmax, __shfl_xor(pmax, off, 64));
        __syncthreads();  // every wave is out of the item loop: the counter word is free
        if (tid == 0) *ctr = 0;
        __syncthreads();
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(ctr), __float_as_uint(pmax));
        __syncthreads();
        if (tid == 0 && *ctr) atomicMax(a.pooled_absmax, (unsigned)*ctr);
    }
}

// ---- split-fp16 preparation: |max| of a tensor, power-of-two scales, (hi, lo) halves of the conv weights
// One atomic per workgroup: thousands of same-address atomics serialise (~20 ns each) and were most of this kernel.
__device__ __forceinline__ void block_atomic_max(unsigned m, unsigned* out) {
    __shared__ unsigned red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m) atomicMax(out, m);
    }
}

__global__ __launch_bounds__(256) void k_absmax(const float* x, size_t n, unsigned* out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);  // |x| as ordered bits (NaN sorts high: scale 1 below)
    block_atomic_max(m, out);
}

__global__ void k_make_scales(const unsigned* absmax, float* scales) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sc[2];
    for (int i = 0; i < 2; ++i) {
        const float m = __uint_as_float(absmax[i]);
        int ex = 0;
        sc[i] = 1.f;
        if (m > 0.f && m < __builtin_inff()) {
            (void)frexpf(m, &ex);          // m < 2^ex
            sc[i] = ldexpf(1.f, 14 - ex);  // m*scale in [2^13, 2^14)
        }
    }
    scales[0] = sc[0];
    scales[1] = sc[1];
    scales[2] = 1.f / (sc[0] * sc[1]);
}

struct SplitArgs {
    const float* src[TG_MAX_FILTERS];
    const float* bias[TG_MAX_FILTERS];
    int woff[TG_MAX_FILTERS];   // halves
    int nk[TG_MAX_FILTERS];     // filters of the bank
    int nfull[TG_MAX_FILTERS];  // rows written as the bank's own tiles (a multiple of 32 if its remainder was merged)
    int f[TG_MAX_FILTERS];
    int foff[TG_MAX_FILTERS];
    int n_filters;
    _Float16* w16;
    const float* scales;
    int n_merged;
    int mwoff[kMaxMerged];
    MergedTile merged[kMaxMerged];
    MergedMeta* mmeta;
};

// Conv weights -> scaled (hi, lo) fp16 halves in the order k_dis_conv<.., SPLIT> loads them: bank k, 32-filter tile
// mt, window row tau, piece p (0-3 hi / 4-7 lo of the k block p&3), lane: 8 halves
// W[min(mt*32 + (lane&31), n_k-1)][tau*64 + 16*(p&3) + 8*(lane>>5) + 0..7] -- one 16-B store per thread, and every
// load instruction of a wave reads 1 KiB contiguous.
__global__ __launch_bounds__(256) void k_split_weights(const SplitArgs a) {
    const float sc = a.scales[0];
    if ((int)blockIdx.y >= a.n_filters) {  // a merged tile: [tau < fmax][8 pieces][64 lanes], zero past a row's width
        const int m = blockIdx.y - a.n_filters;
        const MergedTile& mt = a.merged[m];
        const int total = mt.fmax * 8 * 64;
        f16x8* dst = reinterpret_cast<f16x8*>(a.w16 + a.mwoff[m]);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int lane = i & 63, p = (i >> 6) & 7, tau = i >> 9;
            const int row = lane & 31;
            f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row < mt.nrow) {
                const int k = mt.k[row], fk = a.f[k];
                if (tau < fk) {
                    const float* src = a.src[k] + (size_t)mt.j[row] * fk * kEd + tau * kEd + 16 * (p & 3) + 8 * (lane >> 5);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = src[e] * sc;
                        const _Float16 h = (_Float16)x;
                        o[e] = (p < 4) ? h : (_Float16)(x - (float)h);
                    }
                }
            }
            dst[i] = o;
        }
        if (blockIdx.x == 0 && threadIdx.x < 32) {
            const int row = threadIdx.x;
            MergedMeta& mm = a.mmeta[m];
            if (row < mt.nrow) {
                const int k = mt.k[row];
                mm.bias[row] = a.bias[k][mt.j[row]];
                mm.feat[row] = a.foff[k] + mt.j[row];
                mm.fw[row] = a.f[k];
            } else {
                mm.bias[row] = 0.f;
                mm.feat[row] = -1;
                mm.fw[row] = mt.fmin;
            }
        }
        return;
    }
    const int k = blockIdx.y;
    const int f = a.f[k], nk = a.nk[k], K = f * kEd;
    const int total = ((a.nfull[k] + 31) / 32) * f * 8 * 64;
    f16x8* dst = reinterpret_cast<f16x8*>(a.w16 + a.woff[k]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, p = (i >> 6) & 7, mtau = i >> 9;
        const int tau = mtau % f, mt = mtau / f;
        const int row = tg_min(mt * 32 + (lane & 31), nk - 1);
        const float* src = a.src[k] + (size_t)row * K + tau * kEd + 16 * (p & 3) + 8 * (lane >> 5);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = src[e] * sc;
            const _Float16 h = (_Float16)x;
            o[e] = (p < 4) ? h : (_Float16)(x - (float)h);
        }
        dst[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_absmax_banks(const SplitArgs a, unsigned* out) {
    const int k = blockIdx.y, cnt = a.nk[k] * a.f[k] * kEd;
    unsigned m = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256)
        m = max(m, __float_as_uint(a.src[k][i]) & 0x7fffffffu);
    block_atomic_max(m, out);
}

__global__ __launch_bounds__(256) void k_split(const float* src, size_t n, _Float16* hi, _Float16* lo,
                                               const float* scale) {
    const float sc = *scale;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[e] * sc;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<f16x4*>(hi + i) = h;
            *reinterpret_cast<f16x4*>(lo + i) = l;
        } else {
            for (size_t j = i; j < n; ++j) {
                const float x = src[j] * sc;
                const _Float16 h = (_Float16)x;
                hi[j] = h;
                lo[j] = (_Float16)(x - (float)h);
            }
        }
    }
}

// Running max over time of the sentences' window values: cv[b][t][f] <- max_{t' <= t} cv[b][t'][f]
// (entries with t >= L - f_k + 1 of a width are never read).
__global__ __launch_bounds__(256) void k_dis_prefix_scan(float* cv, int B, int L, int F) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * F) return;
    const int b = idx / F, f = idx - b * F;
    float* p = cv + (size_t)b * L * F + f;
    float m = TG_NEG_INF;
    for (int t = 0; t < L; ++t) {
        m = __builtin_fmaxf(m, p[(size_t)t * F]);
        p[(size_t)t * F] = m;
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int kHwThreads = 256;
constexpr int kHwTile = 128;   // 128 features x 128 sequences per workgroup
constexpr int kHwBK = 64;
constexpr int kHwStride = kHwBK + 4;  // LDS row stride (floats): 16-B aligned, odd number of 16-B slots
constexpr int kHwCh = kHwBK / 4;       // 16-B chunks per tile row
constexpr int kHwPass = kHwTile * kHwCh / kHwThreads;  // float4 per thread per operand tile

struct HwArgs {
    TgDisWeights w;
    const float* pooled;       // [N,F]
    const float* dropout_u;    // [N,F] or null
    float* feature_out;        // [N,F] or null
    float* partial;            // [N, nchunk, kMaxOut]
    float keep, scale;         // keep = 1-p, scale = 1/(1-p); p == 0 -> keep = 2 (always), scale = 1
    int N, nchunk, nseqtile;
    int drop_gen;              // 1: dropout_u is null and the masks come from the counter-based generator
    uint32_t drop_seed, keep_u32;
};

// One thread's share of a 128-row x 32-float operand tile: 4 float4 (rows r0 + 32*i, 16-B chunk c).
// Rows >= nrows are clamped, k >= F reads as 0.  Split into an early global->register stage and a late
// register->LDS stage so the loads of tile k+1 fly under the MFMAs of tile k.
struct HwStage { f32x4 v[kHwPass]; };

__device__ __forceinline__ void hw_fetch(HwStage& st, const float* src, int row0, int nrows, int k0, int F, int tid) {
    const bool vec = (F & 3) == 0;
    const int c = tid % kHwCh, r0 = tid / kHwCh;  // 256 threads: (256/kHwCh) rows x kHwCh chunks per pass
    const int k = k0 + 4 * c;
#pragma unroll
    for (int i = 0; i < kHwPass; ++i) {
        const int row = tg_min(row0 + r0 + (kHwThreads / kHwCh) * i, nrows - 1);
        const float* p = src + (size_t)row * F + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            if (k < F) v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < F) v[e] = p[e];
        }
        st.v[i] = v;
    }
}

__device__ __forceinline__ void hw_commit(const HwStage& st, float* dst, int tid) {
    const int c = tid % kHwCh, r0 = tid / kHwCh;
#pragma unroll
    for (int i = 0; i < kHwPass; ++i)
        *reinterpret_cast<f32x4*>(dst + (r0 + (kHwThreads / kHwCh) * i) * kHwStride + 4 * c) = st.v[i];
}

// NJ 32-sequence tiles per wave, TS sequences per workgroup; red is [feature wave groups][TS][kMaxOut]
template <int NJ, int TS, class ARGS>
__device__ __forceinline__ void hw_epilogue(const ARGS& a, f32x16 (&acc)[2][NJ], float (*red)[TS][kMaxOut], int nwm,
                                            int f0, int s0, int bn, int wm, int wn, int n, int hh, int tid);

__global__ __launch_bounds__(kHwThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dis_highway(const HwArgs a) {
    __shared__ __attribute__((aligned(16))) float As[kHwTile * kHwStride];  // hw_w rows (output features)
    __shared__ __attribute__((aligned(16))) float Bs[kHwTile * kHwStride];  // pooled rows (sequences)
    __shared__ float red[2][kHwTile][kMaxOut];                              // cross-wave logit partials

    const int F = a.w.F;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;  // wave tile: features wm*64.., sequences wn*64..
    const int lane = tid & 63, n = lane & 31, hh = lane >> 5;
    // consecutive workgroups share the sequence tile (B operand) and walk the feature chunks
    const int bn = blockIdx.x % a.nchunk, bm = blockIdx.x / a.nchunk;
    const int f0 = bn * kHwTile, s0 = bm * kHwTile;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    HwStage sa, sb;
    hw_fetch(sa, a.w.hw_w, f0, F, 0, F, tid);
    hw_fetch(sb, a.pooled, s0, a.N, 0, F, tid);
    for (int k0 = 0; k0 < F; k0 += kHwBK) {
        __syncthreads();  // everyone is done reading the previous tile
        hw_commit(sa, As, tid);
        hw_commit(sb, Bs, tid);
        __syncthreads();
        if (k0 + kHwBK < F) {  // next tile's loads fly under this tile's MFMAs
            hw_fetch(sa, a.w.hw_w, f0, F, k0 + kHwBK, F, tid);
            hw_fetch(sb, a.pooled, s0, a.N, k0 + kHwBK, F, tid);
        }
#pragma unroll
        for (int kc = 0; kc < kHwBK / 32; ++kc) {
        float af[2][16], bf[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* pa = As + (wm * 64 + i * 32 + n) * kHwStride + 32 * kc + 16 * hh;
            const float* pb = Bs + (wn * 64 + i * 32 + n) * kHwStride + 32 * kc + 16 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(pa + 4 * q);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { af[i][4 * q + e] = va[e]; bf[i][4 * q + e] = vb[e]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = TG_MFMA32(af[i][j], bf[jn][j], acc[i][jn]);
        }
    }

    hw_epilogue<2, kHwTile>(a, acc, red, 2, f0, s0, bn, wm, wn, n, hh, tid);
}

// ---- highway GEMM on split-fp16 operands (TG_DIS_SPLIT_F16): W_hw pre-split into (hi, lo) halves by k_split, the
// pooled features split while they are staged into LDS (they are read as fp32 either way: no extra pass, no copy)
constexpr int kHsBK = 64;              // halves per tile row
constexpr int kHsStride = kHsBK + 8;   // 144-B rows: 9 16-B slots (odd) -> conflict-free ds_read_b128
struct HsArgs {
    HwArgs h;
    const _Float16* w_hi;   // [F,F] scaled hw_w, lo part F*F halves later
    const float* scales;    // {scale_w, scale_p, 1/(scale_w*scale_p)}
};

struct HsStage { f16x8 v[4]; };
template <int NT>
__device__ __forceinline__ void hs_fetch(HsStage& st, const _Float16* src, int row0, int nrows, int k0, int F, int tid) {
    const int c = tid & 7, r0 = tid >> 3;  // 8 chunks of 8 halves per row, NT/8 rows per pass
    const int k = k0 + 8 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = tg_min(row0 + r0 + (NT / 8) * i, nrows - 1);
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < F) v = *reinterpret_cast<const f16x8*>(src + (size_t)row * F + k);
        st.v[i] = v;
    }
}
template <int NT>
__device__ __forceinline__ void hs_commit(const HsStage& st, _Float16* dst, int tid) {
    const int c = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f16x8*>(dst + (r0 + (NT / 8) * i) * kHsStride + 8 * c) = st.v[i];
}
// the fp32 operand (pooled): same tile shape, 8 floats per (row, chunk)
struct HsStageF { f32x4 v[4][2]; };
template <int NT>
__device__ __forceinline__ void hs_fetch_f32(HsStageF& st, const float* src, int row0, int nrows, int k0, int F, int tid) {
    const int c = tid & 7, r0 = tid >> 3;
    const int k = k0 + 8 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = tg_min(row0 + r0 + (NT / 8) * i, nrows - 1);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        st.v[i][0] = z;
        st.v[i][1] = z;
        if (k < F) {
            const f32x4* p = reinterpret_cast<const f32x4*>(src + (size_t)row * F + k);
            st.v[i][0] = p[0];
            st.v[i][1] = p[1];
        }
    }
}
template <int NT>
__device__ __forceinline__ void hs_commit_split(const HsStageF& st, _Float16* hi, _Float16* lo, float sc, int tid) {
    const int c = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = st.v[i][e >> 2][e & 3] * sc;
            h[e] = (_Float16)x;
            l[e] = (_Float16)(x - (float)h[e]);
        }
        *reinterpret_cast<f16x8*>(hi + (r0 + (NT / 8) * i) * kHsStride + 8 * c) = h;
        *reinterpret_cast<f16x8*>(lo + (r0 + (NT / 8) * i) * kHsStride + 8 * c) = l;
    }
}

// Epilogue of the split kernel (F % 8 == 0): the accumulators (h without its bias) go through LDS so that the
// element-wise part runs row-major -- 32 consecutive threads own one sequence row of the tile, 4 features each:
// pooled / dropout reads and the feature store are 512-B contiguous per row instead of 16-B pieces of 64 rows per
// instruction, and with the accumulators dead every thread's loads (all its rows) are in flight before the first use.
template <int NJ, int TF, int TS, int NT, class ARGS>
__device__ __forceinline__ void hw_epilogue_rows(const ARGS& a, f32x16 (&acc)[2][NJ], float* ht, int f0, int s0, int bn,
                                                 int wm, int wn, int n, int hh, int tid) {
    constexpr int HS = TF + 4;            // h tile row stride (floats)
    constexpr int RP = NT / (TF / 4);     // rows per pass
    constexpr int NP = TS / RP;           // passes
    static_assert(TF / 4 == 32, "one row = 32 threads (the reduction below is over 32 lanes)");
    const int F = a.w.F, n_out = a.w.n_out;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float* d = ht + (wn * (32 * NJ) + jn * 32 + n) * HS + wm * 64 + i * 32 + 8 * rq + 4 * hh;
                *reinterpret_cast<f32x4*>(d) =
                    f32x4{acc[i][jn][4 * rq], acc[i][jn][4 * rq + 1], acc[i][jn][4 * rq + 2], acc[i][jn][4 * rq + 3]};
            }
    __syncthreads();
    const int c4 = tid & 31, r8 = tid >> 5;
    const int fb = f0 + 4 * c4;           // this thread's 4 features (all inside or all outside F: F % 4 == 0)
    const bool fin = fb < F;
    f32x4 pv[NP], uv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int seqc = tg_min(s0 + p * RP + r8, a.N - 1);
        const size_t o4 = (size_t)seqc * F + fb;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pv[p] = z;
        uv[p] = z;
        if (fin) {
            pv[p] = *reinterpret_cast<const f32x4*>(a.pooled + o4);
            if (a.dropout_u) uv[p] = *reinterpret_cast<const f32x4*>(a.dropout_u + o4);
        }
    }
    float hb4[4], ow[kMaxOut][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hb4[e] = fin ? a.w.hw_b[fb + e] : 0.f;
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) ow[o][e] = (fin && o < n_out) ? a.w.out_w[(size_t)o * F + fb + e] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = p * RP + r8;
        const int seq = s0 + row;
        const f32x4 hv = *reinterpret_cast<const f32x4*>(ht + row * HS + 4 * c4);
        float fv[4], part[kMaxOut];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) part[o] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = hv[e] + hb4[e];
            const float sg = tg_sigmoid(h);
            fv[e] = sg * __builtin_fmaxf(h, 0.f) + (1.f - sg) * pv[p][e];
        }
        if (a.feature_out && fin && seq < a.N)
            *reinterpret_cast<f32x4*>(a.feature_out + (size_t)seq * F + fb) = f32x4{fv[0], fv[1], fv[2], fv[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = fin ? fv[e] : 0.f;
            if (a.dropout_u) d = (uv[p][e] < a.keep) ? d * a.scale : 0.f;
            else if (a.drop_gen)
                d = tg_drop_keep((uint32_t)seq * (uint32_t)F + (uint32_t)(fb + e), a.drop_seed, a.keep_u32) ? d * a.scale : 0.f;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o)
                if (o < n_out) part[o] = __builtin_fmaf(d, ow[o][e], part[o]);
        }
        // fixed tree over the row's 32 threads
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < n_out) {
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) part[o] += __shfl_xor(part[o], off, 64);
            }
        }
        if (c4 == 0 && seq < a.N) {
            float* po = a.partial + ((size_t)seq * a.nchunk + bn) * kMaxOut;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o)
                if (o < n_out) po[o] = part[o];
        }
    }
}

// Workgroup tile = (64*WF features) x (32*NJ*WS sequences), one wave per (feature group, sequence group), 2 x NJ
// accumulators of 32 x 32 per wave.  Used as <2,2,2>: 128 x 128, 256 threads, two workgroups per CU.
template <int WF, int WS, int NJ>
__global__ __launch_bounds__(64 * WF * WS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_dis_highway_split(const HsArgs a) {
    constexpr int TF = 64 * WF, TS = 32 * NJ * WS, NT = 64 * WF * WS;
    static_assert(TF * 8 == NT * 4 && TS * 8 == NT * 4, "fetch helpers: 4 passes of NT/8 rows cover a tile");
    // operand tiles (hi, lo of both); the epilogue's cross-wave partials reuse the space
    extern __shared__ __attribute__((aligned(16))) _Float16 tiles[];
    _Float16* Ah = tiles;
    _Float16* Al = Ah + TF * kHsStride;
    _Float16* Bh = Al + TF * kHsStride;
    _Float16* Bl = Bh + TS * kHsStride;
    static_assert(sizeof(float) * TS * (TF + 4) <= sizeof(_Float16) * 2 * (TF + TS) * kHsStride, "h tile must fit");

    const int F = a.h.w.F;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WS, wn = wv % WS;
    const int lane = tid & 63, n = lane & 31, hh = lane >> 5;
    // Persistent workgroups, dealt to the 8 XCDs round-robin by blockIdx.  XCD x owns the sequence tiles bm = x (mod 8)
    // and its workgroups walk that list -- all feature chunks of a tile, then the next tile -- in strides of the
    // XCD's workgroup count.  Every tile costs the same, so the workgroups of an XCD stay within a k-step or two of
    // each other and the operand slices they share (a pooled tile 14 ways, a W_hw slab ~4 ways) are served by that
    // XCD's L2; dispatched one tile per workgroup they drift apart and the L2 thrashes (6.7 GB fetched per launch).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int njx = ((a.h.nseqtile - xcd + 7) >> 3) * a.h.nchunk;  // work items of this XCD
    const size_t wlo = (size_t)F * F;
    const float sp = a.scales[1];
    for (int jx = slot; jx < njx; jx += nslot) {
    const int bn = jx % a.h.nchunk, bm = (jx / a.h.nchunk) * 8 + xcd;
    const int f0 = bn * TF, s0 = bm * TS;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    HsStage sah, sal;
    HsStageF sb;
    hs_fetch<NT>(sah, a.w_hi, f0, F, 0, F, tid);
    hs_fetch<NT>(sal, a.w_hi + wlo, f0, F, 0, F, tid);
    hs_fetch_f32<NT>(sb, a.h.pooled, s0, a.h.N, 0, F, tid);
    for (int k0 = 0; k0 < F; k0 += kHsBK) {
        __syncthreads();
        hs_commit<NT>(sah, Ah, tid);
        hs_commit<NT>(sal, Al, tid);
        hs_commit_split<NT>(sb, Bh, Bl, sp, tid);
        __syncthreads();
        if (k0 + kHsBK < F) {
            hs_fetch<NT>(sah, a.w_hi, f0, F, k0 + kHsBK, F, tid);
            hs_fetch<NT>(sal, a.w_hi + wlo, f0, F, k0 + kHsBK, F, tid);
            hs_fetch_f32<NT>(sb, a.h.pooled, s0, a.h.N, k0 + kHsBK, F, tid);
        }
#pragma unroll
        for (int ks = 0; ks < kHsBK / 16; ++ks) {
            f16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ao = (wm * 64 + i * 32 + n) * kHsStride + 16 * ks + 8 * hh;
                ah[i] = *reinterpret_cast<const f16x8*>(Ah + ao);
                al[i] = *reinterpret_cast<const f16x8*>(Al + ao);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int bo = (wn * (32 * NJ) + j * 32 + n) * kHsStride + 16 * ks + 8 * hh;
                bh[j] = *reinterpret_cast<const f16x8*>(Bh + bo);
                bl[j] = *reinterpret_cast<const f16x8*>(Bl + bo);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) {
                    acc[i][jn] = TG_MFMA16(ah[i], bh[jn], acc[i][jn]);
                    acc[i][jn] = TG_MFMA16(ah[i], bl[jn], acc[i][jn]);
                    acc[i][jn] = TG_MFMA16(al[i], bh[jn], acc[i][jn]);
                }
        }
    }
    const float inv = a.scales[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    __syncthreads();  // the last tile's reads are done before the epilogue reuses the space
    hw_epilogue_rows<NJ, TF, TS, NT>(a.h, acc, reinterpret_cast<float*>(tiles), f0, s0, bn, wm, wn, n, hh, tid);
    }
}

template <int NJ, int TS, class ARGS>
__device__ __forceinline__ void hw_epilogue(const ARGS& a, f32x16 (&acc)[2][NJ], float (*red)[TS][kMaxOut], int nwm,
                                            int f0, int s0, int bn, int wm, int wn, int n, int hh, int tid) {
    const int F = a.w.F, n_out = a.w.n_out;
    // ---- epilogue: gate, dropout, partial feature2out dot (lane = sequence, regs = features)
    float part[NJ][kMaxOut];
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) part[jn][o] = 0.f;
    const bool vec = (F & 3) == 0;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
        const int seq = s0 + wn * (32 * NJ) + jn * 32 + n;
        const int seqc = tg_min(seq, a.N - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int fb = f0 + wm * 64 + i * 32 + 8 * rq + 4 * hh;  // 4 consecutive features
                if (fb >= F) continue;
                float pv[4], uv[4];
                const size_t o4 = (size_t)seqc * F + fb;
                if (vec) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(a.pooled + o4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[e] = t[e];
                    if (a.dropout_u) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(a.dropout_u + o4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) uv[e] = u[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[e] = (fb + e < F) ? a.pooled[o4 + e] : 0.f;
                        uv[e] = (a.dropout_u && fb + e < F) ? a.dropout_u[o4 + e] : 0.f;
                    }
                }
                float fv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fi = tg_min(fb + e, F - 1);
                    const float h = acc[i][jn][4 * rq + e] + a.w.hw_b[fi];
                    const float sg = tg_sigmoid(h);
                    fv[e] = sg * __builtin_fmaxf(h, 0.f) + (1.f - sg) * pv[e];
                }
                if (a.feature_out && seq < a.N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (fb + e < F) a.feature_out[o4 + e] = fv[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (fb + e >= F) continue;
                    float d = fv[e];
                    if (a.dropout_u) d = (uv[e] < a.keep) ? d * a.scale : 0.f;
                    else if (a.drop_gen)
                        d = tg_drop_keep((uint32_t)seq * (uint32_t)F + (uint32_t)(fb + e), a.drop_seed, a.keep_u32) ? d * a.scale : 0.f;
#pragma unroll
                    for (int o = 0; o < kMaxOut; ++o)
                        if (o < n_out) part[jn][o] = __builtin_fmaf(d, a.w.out_w[(size_t)o * F + fb + e], part[jn][o]);
                }
            }
        }
    }
    // lanes n and n+32 hold the same sequence; then the feature wave groups (wm = 0..nwm-1, fixed order)
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) part[jn][o] += __shfl_xor(part[jn][o], 32, 64);
    if (hh == 0) {
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) red[wm][wn * (32 * NJ) + jn * 32 + n][o] = part[jn][o];
    }
    __syncthreads();
    if (tid < TS) {
        const int seq = s0 + tid;
        if (seq < a.N) {
            float* po = a.partial + ((size_t)seq * a.nchunk + bn) * kMaxOut;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) {
                float v = red[0][tid][o];
                for (int g = 1; g < nwm; ++g) v += red[g][tid][o];
                po[o] = v;
            }
        }
    }
}

struct FinalArgs {
    const float* partial;  // [N, nchunk, kMaxOut]
    const float* out_b;
    float* reward_out;     // [N] or null
    float* logits_out;     // [N, n_out] or null
    int N, nchunk, n_out, col;
};

__global__ __launch_bounds__(256) void k_dis_final(const FinalArgs a) {
    const int seq = blockIdx.x * 256 + threadIdx.x;
    if (seq >= a.N) return;
    float lg[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) lg[o] = (o < a.n_out) ? a.out_b[o] : TG_NEG_INF;
    const float* p = a.partial + (size_t)seq * a.nchunk * kMaxOut;
    for (int c = 0; c < a.nchunk; ++c)
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
            if (o < a.n_out) lg[o] += p[c * kMaxOut + o];
    float M = TG_NEG_INF;
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) M = __builtin_fmaxf(M, lg[o]);
    float S = 0.f, pick = 0.f;
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) {
        const float e = tg_expf(lg[o] - M);
        S += e;
        pick = (o == a.col) ? e : pick;
        if (a.logits_out && o < a.n_out) a.logits_out[(size_t)seq * a.n_out + o] = lg[o];
    }
    if (a.reward_out) a.reward_out[seq] = pick / S;
}

int check_dis(const TgDisWeights* w, int L) {
    if (!w || !w->emb || !w->hw_w || !w->hw_b || !w->out_w || !w->out_b) {
        tg_set_error("discriminator weights: null pointer");
        return TG_ERR_ARG;
    }
    if (w->Ed != kEd) {
        tg_set_error("discriminator embedding dim %d not supported by this build (64 only)", w->Ed);
        return TG_ERR_ARG;
    }
    if (w->n_filters < 1 || w->n_filters > TG_MAX_FILTERS || w->n_out < 1 || w->n_out > kMaxOut) {
        tg_set_error("discriminator: n_filters=%d n_out=%d out of range", w->n_filters, w->n_out);
        return TG_ERR_ARG;
    }
    int F = 0;
    for (int k = 0; k < w->n_filters; ++k) {
        if (!w->conv_w[k] || !w->conv_b[k] || w->f[k] < 1 || w->n[k] < 1) {
            tg_set_error("discriminator: bad filter bank %d", k);
            return TG_ERR_ARG;
        }
        if (w->f[k] > L) {
            // the reference raises RuntimeError from conv2d here (kernel larger than input)
            tg_set_error("sequence length %d is shorter than filter width %d", L, w->f[k]);
            return TG_ERR_ARG;
        }
        F += w->n[k];
    }
    if (F != w->F) {
        tg_set_error("discriminator: feature dim %d != sum of filters %d", w->F, F);
        return TG_ERR_ARG;
    }
    return TG_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int tg_min_host(int a, int b) { return a < b ? a : b; }

}  // namespace

namespace {

// Mixed-width tiles from the banks' remainder rows (split path): remainders sorted by width, packed greedily 32
// rows to a tile; a tile with a single member is left to the bank's own partial tile.
struct MergePlan {
    int n_merged;
    MergedTile tile[kMaxMerged];
    bool bank_merged[TG_MAX_FILTERS];  // the bank's remainder rows live in a merged tile
    int pad_rows;
};

MergePlan plan_merge(const TgDisWeights* w) {
    MergePlan pl = {};
    if (w->precision != TG_DIS_SPLIT_F16) return pl;
    int order[TG_MAX_FILTERS], no = 0;
    for (int k = 0; k < w->n_filters; ++k)
        if (w->n[k] % 32 != 0 && w->n[k] < 256 && w->f[k] < 256) order[no++] = k;
    for (int i = 1; i < no; ++i) {  // by width
        const int t = order[i];
        int j = i - 1;
        while (j >= 0 && w->f[order[j]] > w->f[t]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = t;
    }
    int i = 0;
    while (i < no && pl.n_merged < kMaxMerged) {
        int rows = 0, j = i;
        while (j < no && rows + w->n[order[j]] % 32 <= 32) { rows += w->n[order[j]] % 32; ++j; }
        if (j - i >= 2) {
            MergedTile& t = pl.tile[pl.n_merged++];
            t.nrow = 0;
            t.fmin = (uint8_t)w->f[order[i]];
            t.fmax = (uint8_t)w->f[order[j - 1]];
            for (int q = i; q < j; ++q) {
                const int k = order[q], base = (w->n[k] / 32) * 32;
                for (int r = base; r < w->n[k]; ++r) { t.k[t.nrow] = (uint8_t)k; t.j[t.nrow] = (uint8_t)r; ++t.nrow; }
                pl.bank_merged[k] = true;
            }
            if (t.fmax - t.fmin > pl.pad_rows) pl.pad_rows = t.fmax - t.fmin;
        }
        i = (j > i) ? j : i + 1;
    }
    return pl;
}

int prepare_conv(ConvArgs& ca, const TgDisWeights* w, int N, int L, size_t* smem) {
    ca.w = *w;
    ca.L = L;
    ca.stamps = tg_diag_pointer("TG_CONV_STAMPS");
    const MergePlan pl = plan_merge(w);
    const int LP = L + pl.pad_rows;  // LDS rows per sequence
    int S = 32, log2S = 5;
    while (S > 1 && (size_t)LP * S * kERow * sizeof(float) + 64 > 144 * 1024) { S >>= 1; --log2S; }
    // few sequences (the base sentences of the prefix pass): fewer per workgroup so that the grid fills the chip
    while (S > 1 && (N + S - 1) / S < 512) { S >>= 1; --log2S; }
    if ((size_t)LP * S * kERow * sizeof(float) + 64 > 160 * 1024) {
        tg_set_error("tg_dis_score: sequence length %d too long for the LDS-resident conv kernel", L);
        return TG_ERR_ARG;
    }
    ca.S = S;
    ca.log2S = log2S;
    ca.pad_rows = pl.pad_rows;
    ca.n_merged = pl.n_merged;
    int off = 0, ni = 0;
    struct Tmp { int k, mt; long cost; } tmp[kMaxItems];
    for (int k = 0; k < w->n_filters; ++k) {
        ca.foff[k] = off;
        off += w->n[k];
        const int ntile = pl.bank_merged[k] ? w->n[k] / 32 : (w->n[k] + 31) / 32;
        for (int mt = 0; mt < ntile; ++mt) {
            if (ni >= kMaxItems) {
                tg_set_error("tg_dis_score: too many filter tiles (> %d)", kMaxItems);
                return TG_ERR_ARG;
            }
            tmp[ni].k = k; tmp[ni].mt = mt; tmp[ni].cost = (long)w->f[k] * (L - w->f[k] + 1);
            ++ni;
        }
    }
    for (int m = 0; m < pl.n_merged; ++m) {
        if (ni >= kMaxItems) {
            tg_set_error("tg_dis_score: too many filter tiles (> %d)", kMaxItems);
            return TG_ERR_ARG;
        }
        ca.mfmax[m] = pl.tile[m].fmax;
        ca.mfmin[m] = pl.tile[m].fmin;
        tmp[ni].k = 0xFF; tmp[ni].mt = m; tmp[ni].cost = (long)pl.tile[m].fmax * (L - pl.tile[m].fmin + 1);
        ++ni;
    }
    for (int i = 1; i < ni; ++i) {  // insertion sort, heaviest first
        Tmp t = tmp[i];
        int j = i - 1;
        while (j >= 0 && tmp[j].cost < t.cost) { tmp[j + 1] = tmp[j]; --j; }
        tmp[j + 1] = t;
    }
    for (int i = 0; i < ni; ++i) { ca.items[i].k = (uint8_t)tmp[i].k; ca.items[i].mt = (uint8_t)tmp[i].mt; }
    ca.n_items = ni;
    *smem = (size_t)LP * S * kERow * sizeof(float) + 64;
    return TG_OK;
}

template <int MODE, bool SPLIT>
int launch_conv_t(const ConvArgs& ca, size_t sm, hipStream_t stream) {
    static TgLdsAttr attr;
    if (int rc = tg_lds_attr(attr, reinterpret_cast<const void*>(&k_dis_conv<MODE, SPLIT>), sm, "k_dis_conv")) return rc;
    tg_prof_begin(MODE == kConvStoreAll ? TG_KERNEL_DIS_PREFIX : TG_KERNEL_DIS_CONV, stream);
    // a grid of fewer workgroups than CUs (the sentence pass: B sequences) is widened over the (width, tile) items
    const int gx = (ca.N + ca.S - 1) / ca.S;
    const int gy = (MODE == kConvStoreAll && gx < 256) ? tg_min_host(tg_min_host(8, ca.n_items), (512 + gx - 1) / gx) : 1;
    hipLaunchKernelGGL((k_dis_conv<MODE, SPLIT>), dim3(gx, gy), dim3(kConvThreads), sm, stream, ca);
    tg_prof_end(stream);
    return TG_OK;
}

template <int MODE>
int launch_conv(const ConvArgs& ca, size_t sm, hipStream_t stream) {
    return ca.w16 ? launch_conv_t<MODE, true>(ca, sm, stream) : launch_conv_t<MODE, false>(ca, sm, stream);
}

// fp16 halves of one bank in fragment order (hi and lo, filter count padded to a multiple of 32); an upper bound
// when the bank's remainder rows live in a merged tile
size_t conv_frag_halves(const TgDisWeights* w, int k) { return (size_t)((w->n[k] + 31) / 32) * w->f[k] * 8 * 64 * 8; }
size_t merged_frag_halves(int fmax) { return (size_t)fmax * 8 * 64 * 8; }

size_t conv_weight_count(const TgDisWeights* w) {  // (hi, lo) pairs, merged tiles and their row tables included
    size_t n = 0;
    int fmax = 1;
    for (int k = 0; k < w->n_filters; ++k) {
        n += conv_frag_halves(w, k) / 2;
        fmax = w->f[k] > fmax ? w->f[k] : fmax;
    }
    return n + kMaxMerged * (merged_frag_halves(fmax) / 2 + sizeof(MergedMeta) / 4 + 64);
}

size_t ws_bytes(const TgDisWeights* w, int N, int L, int B_prefix) {
    const int nchunk = (w->F + kHwTile - 1) / kHwTile;
    return align256(sizeof(float) * (size_t)N * w->F) + align256(sizeof(float) * (size_t)N * nchunk * kMaxOut) +
           align256(sizeof(float) * (size_t)B_prefix * L * w->F) +
           align256(2 * sizeof(_Float16) * conv_weight_count(w)) + 256 +
           align256(2 * sizeof(_Float16) * (size_t)w->F * w->F);
}

// Split-fp16 preparation (every call: the weights are live module parameters): |max| of the embedding table and
// of all conv banks, the two power-of-two scales, and the (hi, lo) halves of the conv weights.
// reuse: the workspace still holds the scales and halves an earlier call wrote for the SAME weights and N (the
// position loop of tg_leak_rollout): only the offsets are recomputed and the pooled |max| slot is cleared.
int prepare_split(ConvArgs& ca, const TgDisWeights* w, _Float16* w16, void* scal, hipStream_t stream, bool reuse) {
    unsigned* absmax = (unsigned*)scal;          // [0] conv weights, [1] embedding table
    float* scales = (float*)((char*)scal + 16);  // {scale_w, scale_e, 1/(scale_w*scale_e), -} x 2 (conv, highway)
    if (hipMemsetAsync(reuse ? absmax + 3 : absmax, 0, reuse ? 4 : 16, stream) != hipSuccess) {
        tg_set_error("hipMemsetAsync failed");
        return TG_ERR_LAUNCH;
    }
    if (!reuse) tg_prof_begin(TG_KERNEL_DIS_PREFIX, stream);
    SplitArgs sa = {};
    const MergePlan pl = plan_merge(w);
    int off = 0, maxcount = 0, maxelems = 0, foff = 0;
    for (int k = 0; k < w->n_filters; ++k) {
        const int cnt = w->n[k] * w->f[k] * kEd;
        sa.src[k] = w->conv_w[k];
        sa.bias[k] = w->conv_b[k];
        sa.woff[k] = off;
        sa.nk[k] = w->n[k];
        sa.nfull[k] = pl.bank_merged[k] ? (w->n[k] / 32) * 32 : w->n[k];
        sa.f[k] = w->f[k];
        sa.foff[k] = foff;
        foff += w->n[k];
        ca.woff[k] = off;
        const int pieces = (int)(conv_frag_halves(w, k) / 8);
        off += pieces * 8;
        maxcount = pieces > maxcount ? pieces : maxcount;
        maxelems = cnt > maxelems ? cnt : maxelems;
    }
    sa.n_filters = w->n_filters;
    sa.n_merged = pl.n_merged;
    for (int m = 0; m < pl.n_merged; ++m) {
        sa.merged[m] = pl.tile[m];
        sa.mwoff[m] = off;
        ca.mwoff[m] = off;
        off += (int)merged_frag_halves(pl.tile[m].fmax);
    }
    off = (off + 127) & ~127;  // the row tables behind the fragments, 256-B aligned
    sa.mmeta = reinterpret_cast<MergedMeta*>(w16 + off);
    ca.mmeta = sa.mmeta;
    sa.w16 = w16;
    sa.scales = scales;
    if (!reuse) {
        hipLaunchKernelGGL(k_absmax_banks, dim3(tg_min_host((maxelems + 255) / 256, 64), w->n_filters), dim3(256), 0,
                           stream, sa, absmax);
        const size_t ne = (size_t)w->V * kEd;
        hipLaunchKernelGGL(k_absmax, dim3(tg_min_host((int)((ne + 255) / 256), 256)), dim3(256), 0, stream, w->emb, ne,
                           absmax + 1);
        hipLaunchKernelGGL(k_make_scales, dim3(1), dim3(64), 0, stream, absmax, scales);
        hipLaunchKernelGGL(k_split_weights, dim3(tg_min_host((maxcount + 255) / 256, 256), w->n_filters + pl.n_merged),
                           dim3(256), 0, stream, sa);
        tg_prof_end(stream);
    }
    ca.w16 = w16;
    ca.scales = scales;
    ca.pooled_absmax = absmax + 3;  // read by the highway's scale (score_impl)
    return TG_OK;
}

template <int WF, int WS, int NJ>
int launch_highway_split(HsArgs& hs, hipStream_t stream) {
    constexpr int TF = 64 * WF, TS = 32 * NJ * WS;
    constexpr size_t sm = sizeof(_Float16) * 2 * (TF + TS) * kHsStride;
    static TgLdsAttr attr;
    if (int rc = tg_lds_attr(attr, reinterpret_cast<const void*>(&k_dis_highway_split<WF, WS, NJ>), sm, "k_dis_highway_split")) return rc;
    hs.h.nchunk = (hs.h.w.F + TF - 1) / TF;
    hs.h.nseqtile = (hs.h.N + TS - 1) / TS;
    tg_prof_begin(TG_KERNEL_DIS_HIGHWAY, stream);
    // persistent: as many workgroups as are resident at once (sm bytes of LDS each), a multiple of the 8 XCDs
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = (int)((160 * 1024) / sm) > 0 ? (int)((160 * 1024) / sm) : 1;
    const long total = (long)hs.h.nseqtile * hs.h.nchunk;
    long grid = (long)ncu * per_cu;
    if (grid > total) grid = total;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((k_dis_highway_split<WF, WS, NJ>), dim3((unsigned)grid), dim3(64 * WF * WS), sm, stream, hs);
    tg_prof_end(stream);
    return TG_OK;
}

// sentences == nullptr: plain scoring of N arbitrary sequences.
// sentences != nullptr: seqs is [R,L,B,L] rollout completions (N = R*L*B), prefix reuse.
int score_impl(const TgDisWeights* w, const int64_t* seqs, int N, int L, const int64_t* sentences, int B, int col,
               float dropout_p, const float* dropout_uniforms, float* reward_out, float* logits_out,
               float* feature_out, void* workspace, size_t workspace_bytes, hipStream_t stream, bool reuse = false) {
    if (N < 0 || L < 1 || !seqs) {
        tg_set_error("tg_dis_score: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    int rc = check_dis(w, L);
    if (rc) return rc;
    if (reward_out && (col < 0 || col >= w->n_out)) {
        tg_set_error("tg_dis_score: softmax column %d outside [0,%d)", col, w->n_out);
        return TG_ERR_ARG;
    }
    if (dropout_p < 0.f || dropout_p >= 1.f) {
        tg_set_error("tg_dis_score: dropout p=%f outside [0,1)", dropout_p);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    const size_t need = ws_bytes(w, N, L, sentences ? B : 0);
    if (!workspace || workspace_bytes < need) {
        tg_set_error("tg_dis_score: workspace too small (%zu < %zu)", workspace_bytes, need);
        return TG_ERR_WORKSPACE;
    }
    float* pooled = (float*)workspace;
    const int nchunk = (w->F + kHwTile - 1) / kHwTile;
    float* partial = (float*)((char*)workspace + align256(sizeof(float) * (size_t)N * w->F));
    float* cv = (float*)((char*)partial + align256(sizeof(float) * (size_t)N * nchunk * kMaxOut));
    _Float16* w16 = (_Float16*)((char*)cv + align256(sizeof(float) * (size_t)(sentences ? B : 0) * L * w->F));
    void* scal = (char*)w16 + align256(2 * sizeof(_Float16) * conv_weight_count(w));
    _Float16* hw16 = (_Float16*)((char*)scal + 256);

    // ---- conv
    ConvArgs ca = {};
    size_t sm = 0;
    rc = prepare_conv(ca, w, sentences ? B : N, L, &sm);
    if (rc) return rc;
    if (w->precision == TG_DIS_SPLIT_F16) {
        rc = prepare_split(ca, w, w16, scal, stream, reuse);
        if (rc) return rc;
    } else if (w->precision != TG_DIS_FP32) {
        tg_set_error("tg_dis_score: unknown precision %d", w->precision);
        return TG_ERR_ARG;
    }
    if (sentences) {
        // (1) every window of the B base sentences, (2) running max over time, (3) completions: fresh windows only
        ca.seqs = sentences;
        ca.N = B;
        ca.cv = cv;
        rc = launch_conv<kConvStoreAll>(ca, sm, stream);
        if (rc) return rc;
        tg_prof_begin(TG_KERNEL_DIS_PREFIX, stream);
        hipLaunchKernelGGL(k_dis_prefix_scan, dim3((B * w->F + 255) / 256), dim3(256), 0, stream, cv, B, L, w->F);
        tg_prof_end(stream);
        rc = prepare_conv(ca, w, N, L, &sm);
        if (rc) return rc;
        ca.seqs = seqs;
        ca.pooled = pooled;
        ca.cv = cv;
        ca.N = N;
        ca.B = B;
        rc = launch_conv<kConvPrefix>(ca, sm, stream);
    } else {
        ca.seqs = seqs;
        ca.pooled = pooled;
        ca.N = N;
        rc = launch_conv<kConvPlain>(ca, sm, stream);
    }
    if (rc) return rc;

    // ---- highway + gate + dropout + partial logits
    HwArgs ha = {};
    ha.w = *w;
    ha.pooled = pooled;
    ha.dropout_u = (dropout_p > 0.f) ? dropout_uniforms : nullptr;
    ha.drop_gen = (dropout_p > 0.f && !dropout_uniforms) ? 1 : 0;
    ha.drop_seed = w->dropout_seed;
    {
        const double k32 = (1.0 - (double)dropout_p) * 4294967296.0;
        ha.keep_u32 = k32 >= 4294967295.0 ? 0xffffffffu : (uint32_t)k32;
    }
    if (ha.drop_gen && (size_t)N * w->F > 0xffffffffull) {
        tg_set_error("tg_dis_score: counter-based dropout indexes N*F = %zu elements with 32 bits; score in chunks",
                     (size_t)N * w->F);
        return TG_ERR_ARG;
    }
    ha.feature_out = feature_out;
    ha.partial = partial;
    ha.keep = 1.f - dropout_p;
    ha.scale = 1.f / (1.f - dropout_p);
    ha.N = N;
    ha.nchunk = nchunk;
    int nchunk_used = nchunk;
    ha.nseqtile = (N + kHwTile - 1) / kHwTile;
    if (w->precision == TG_DIS_SPLIT_F16 && (w->F & 7) == 0) {
        // scales from |max| of hw_w and of the pooled features (the conv kernels' epilogue left that one), then the
        // (hi, lo) halves of hw_w
        unsigned* absmax = (unsigned*)scal + 2;
        float* scales = (float*)((char*)scal + 16) + 4;
        const size_t nw = (size_t)w->F * w->F;
        tg_prof_begin(TG_KERNEL_DIS_PREFIX, stream);
        if (!reuse)
            hipLaunchKernelGGL(k_absmax, dim3(tg_min_host((int)((nw + 255) / 256), 512)), dim3(256), 0, stream, w->hw_w,
                               nw, absmax);
        // the pooled features' |max| is this call's own (the conv epilogue left it): the scales are always rebuilt
        hipLaunchKernelGGL(k_make_scales, dim3(1), dim3(64), 0, stream, absmax, scales);
        if (!reuse)
            hipLaunchKernelGGL(k_split, dim3(tg_min_host((int)((nw / 4 + 255) / 256) + 1, 2048)), dim3(256), 0, stream,
                               w->hw_w, nw, hw16, hw16 + nw, scales);
        tg_prof_end(stream);
        HsArgs hs = {};
        hs.h = ha;
        hs.w_hi = hw16;
        hs.scales = scales;
        // 128 x 128 tiles, two workgroups per CU.  (256 x 256 with 512 threads -- <4,2,4> -- halves the operand traffic
        // but measured 1.7x slower on MI355X: one workgroup per CU, 86 spilled registers, barrier stalls uncovered.)
        rc = launch_highway_split<2, 2, 2>(hs, stream);
        if (rc) return rc;
        nchunk_used = hs.h.nchunk;
    } else {
        tg_prof_begin(TG_KERNEL_DIS_HIGHWAY, stream);
        hipLaunchKernelGGL(k_dis_highway, dim3(ha.nseqtile * nchunk), dim3(kHwThreads), 0, stream, ha);
        tg_prof_end(stream);
    }

    // ---- final: fixed-order partial sum, softmax column
    FinalArgs fa = {};
    fa.partial = partial;
    fa.out_b = w->out_b;
    fa.reward_out = reward_out;
    fa.logits_out = logits_out;
    fa.N = N;
    fa.nchunk = nchunk_used;
    fa.n_out = w->n_out;
    fa.col = col;
    tg_prof_begin(TG_KERNEL_DIS_FINAL, stream);
    hipLaunchKernelGGL(k_dis_final, dim3((N + 255) / 256), dim3(256), 0, stream, fa);
    tg_prof_end(stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("tg_dis_score launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

}  // namespace

namespace {

struct RescoreArgs {
    const float* feature;   // [N,F]
    const float* uniforms;  // [R,N,F] or null
    const float* out_w;     // [n_out,F]
    const float* out_b;
    float* reward_out;      // [R,N]
    int N, R, F, n_out, col;
    float keep, scale;
    int drop_gen;
    uint32_t drop_seed, keep_u32;
};

// One workgroup per sample: the feature row and the output layer's columns stay in registers over the R masks.
// Fixed summation order (thread-strided partials, xor tree, waves in order): bit-reproducible.
constexpr int kRescorePer = 8;  // features per thread: F <= 256 * 8
__global__ __launch_bounds__(256) void k_dis_rescore(const RescoreArgs a) {
    __shared__ float red[4][kMaxOut];
    const int nidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float fv[kRescorePer], ow[kMaxOut][kRescorePer];
#pragma unroll
    for (int i = 0; i < kRescorePer; ++i) {
        const int f = tid + 256 * i;
        fv[i] = (f < a.F) ? a.feature[(size_t)nidx * a.F + f] : 0.f;
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) ow[o][i] = (f < a.F && o < a.n_out) ? a.out_w[(size_t)o * a.F + f] : 0.f;
    }
    for (int r = 0; r < a.R; ++r) {
        float part[kMaxOut];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) part[o] = 0.f;
        const float* u = a.uniforms ? a.uniforms + ((size_t)r * a.N + nidx) * a.F : nullptr;
#pragma unroll
        for (int i = 0; i < kRescorePer; ++i) {
            const int f = tid + 256 * i;
            float d = fv[i];
            if (u) d = (f < a.F && u[f] < a.keep) ? d * a.scale : 0.f;
            else if (a.drop_gen)
                d = tg_drop_keep(((uint32_t)r * (uint32_t)a.N + (uint32_t)nidx) * (uint32_t)a.F + (uint32_t)f, a.drop_seed,
                                 a.keep_u32) ? d * a.scale : 0.f;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) part[o] = __builtin_fmaf(d, ow[o][i], part[o]);
        }
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) part[o] += __shfl_xor(part[o], off, 64);
        __syncthreads();  // red is free again
        if (lane == 0)
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) red[wv][o] = part[o];
        __syncthreads();
        if (tid == 0) {
            float lg[kMaxOut], M = TG_NEG_INF;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) {
                lg[o] = (o < a.n_out) ? a.out_b[o] + (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]) : TG_NEG_INF;
                M = __builtin_fmaxf(M, lg[o]);
            }
            float S = 0.f, pick = 0.f;
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) {
                const float e = tg_expf(lg[o] - M);
                S += e;
                pick = (o == a.col) ? e : pick;
            }
            a.reward_out[(size_t)r * a.N + nidx] = pick / S;
        }
    }
}

}  // namespace

extern "C" int tg_dis_rescore_dropout(const TgDisWeights* w, const float* feature, int N, int R, int col,
                                      float dropout_p, const float* dropout_uniforms, float* reward_out,
                                      void* stream) {
    if (!w || !w->out_w || !w->out_b || !feature || !reward_out || N < 0 || R < 1) {
        tg_set_error("tg_dis_rescore_dropout: bad arguments (N=%d R=%d)", N, R);
        return TG_ERR_ARG;
    }
    if (w->F < 1 || w->F > 256 * kRescorePer || w->n_out < 1 || w->n_out > kMaxOut || col < 0 || col >= w->n_out) {
        tg_set_error("tg_dis_rescore_dropout: F=%d (max %d), n_out=%d, column %d out of range", w->F,
                     256 * kRescorePer, w->n_out, col);
        return TG_ERR_ARG;
    }
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) {
        tg_set_error("tg_dis_rescore_dropout: dropout_p=%g must be in [0,1)", (double)dropout_p);
        return TG_ERR_ARG;
    }
    if (dropout_p > 0.f && !dropout_uniforms && (size_t)R * N * w->F > 0xffffffffull) {
        tg_set_error("tg_dis_rescore_dropout: counter-based dropout indexes R*N*F elements with 32 bits");
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    RescoreArgs a = {};
    a.feature = feature;
    a.uniforms = dropout_p > 0.f ? dropout_uniforms : nullptr;
    a.out_w = w->out_w;
    a.out_b = w->out_b;
    a.reward_out = reward_out;
    a.N = N; a.R = R; a.F = w->F; a.n_out = w->n_out; a.col = col;
    a.keep = 1.f - dropout_p;
    a.scale = 1.f / (1.f - dropout_p);
    a.drop_gen = (dropout_p > 0.f && !dropout_uniforms) ? 1 : 0;
    a.drop_seed = w->dropout_seed;
    {
        const double k32 = (1.0 - (double)dropout_p) * 4294967296.0;
        a.keep_u32 = k32 >= 4294967295.0 ? 0xffffffffu : (uint32_t)k32;
    }
    tg_prof_begin(TG_KERNEL_DIS_FINAL, (hipStream_t)stream);
    hipLaunchKernelGGL(k_dis_rescore, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
    tg_prof_end((hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("k_dis_rescore launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

extern "C" size_t tg_dis_workspace_bytes(const TgDisWeights* w, int N, int L) {
    if (!w || N < 1) return 0;
    return ws_bytes(w, N, L, 0);
}

extern "C" int tg_dis_score(const TgDisWeights* w, const int64_t* seqs, int N, int L, int col, float dropout_p,
                            const float* dropout_uniforms, float* reward_out, float* logits_out, float* feature_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return score_impl(w, seqs, N, L, nullptr, 0, col, dropout_p, dropout_uniforms, reward_out, logits_out,
                      feature_out, workspace, workspace_bytes, (hipStream_t)stream);
}

// get_feature for the LeakGAN rollout's position loop (tg_leak.hip): tg_dis_score restricted to feature_out, with
// `reuse` = the workspace already holds this call's prepared weights (same w, same N as the call that wrote them).
int tg_dis_features(const TgDisWeights* w, const int64_t* seqs, int N, int L, float* feature_out, void* workspace,
                    size_t workspace_bytes, bool reuse, hipStream_t stream) {
    return score_impl(w, seqs, N, L, nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, feature_out, workspace,
                      workspace_bytes, stream, reuse);
}

extern "C" size_t tg_dis_score_rollout_workspace_bytes(const TgDisWeights* w, int B, int L, int R) {
    if (!w || B < 1 || L < 1 || R < 1) return 0;
    return ws_bytes(w, R * L * B, L, B);
}

extern "C" int tg_dis_score_rollout(const TgDisWeights* w, const int64_t* sentences, const int64_t* completions,
                                    int B, int L, int R, int col, float dropout_p, const float* dropout_uniforms,
                                    float* reward_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sentences || B < 1 || R < 0 || (long)R * L * B > 0x7fffffffL) {
        tg_set_error("tg_dis_score_rollout: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    return score_impl(w, completions, R * L * B, L, sentences, B, col, dropout_p, dropout_uniforms, reward_out,
                      nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}
