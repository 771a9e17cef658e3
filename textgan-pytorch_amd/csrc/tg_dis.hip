// K-DIS: the CNN discriminator scorer (gfx950).
//
//   k_dis_conv     embedding gather -> LDS; all filter widths as implicit GEMMs on
//                  v_mfma_f32_32x32x2_f32 (M = filters, N = (sequence, time), K = f*Ed);
//                  bias + ReLU + max-over-time in registers; pooled [N,F] out.
//                  Split-fp16 arithmetic (default): operands as (hi, lo) fp16 pairs on
//                  v_mfma_f32_32x32x16_f16, pooled features out as the highway GEMM's B pieces.
//   k_dis_highway_frag  h = pooled . W_hw^T on fragment-ordered split-fp16 operands: persistent,
//                  XCD-aware, LDS-DMA double buffered; fused sigmoid/relu gate, dropout mask and
//                  the partial feature2out dot; partial logits [N, slots, n_out] out.
//   k_dis_highway  the same stage in exact fp32 (TG_DIS_FP32, or F % 8 != 0): 128x128 tiles.
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
    int valid_len;        // positions t >= valid_len are read as token 0 (padding) whatever seqs holds there; Plain
                          // mode evaluates the windows that start below it only and takes the others, all padding,
    const float* padpool; // from here: [F] pooled features of an all-padding sequence (null: valid_len == L)
    int S, log2S;         // sequences per workgroup (power of two <= 32), TT = 32 / S
    int n_items;
    int foff[TG_MAX_FILTERS];  // feature offset of each width
    // split-fp16 path: weights pre-split into scaled (hi, lo) halves, bank k's hi part at w16 + woff[k],
    // its lo part n_k*f_k*64 halves later; scales = {scale_w, scale_e, 1/(scale_w*scale_e)} on the device
    const _Float16* w16;
    const float* scales;
    _Float16* pfrag;           // nullable (split path, F % 8 == 0): the pooled features go out as the highway GEMM's B
                               // fragments [seq/32][KBp][hi|lo][64 lanes] x 8 halves, scaled by scales[5], instead of
                               // the fp32 matrix pooled[N,F]
    int KBp;                   // 16-feature k blocks per sequence block in pfrag
    int woff[TG_MAX_FILTERS];
    int n_merged;              // mixed-width remainder tiles (items with k == 0xFF, mt = tile index); split path only
    int pad_rows;              // zero rows kept after the sequence in LDS: max (fmax - fmin) over the merged tiles
    int mwoff[kMaxMerged];     // halves offset of a merged tile's fragments in w16
    uint8_t mfmax[kMaxMerged], mfmin[kMaxMerged];
    const MergedMeta* mmeta;   // [n_merged]
    ConvItem items[kMaxItems]; // sorted by cost, heaviest first
    long long* stamps;         // diagnostic (TG_CONV_STAMPS env = device pointer): per wave [8] values (8 waves per workgroup)
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kERowH = 2 * kERow;  // LDS halves per embedded token in the split path: [64 hi | 64 lo | 8 pad]
#define TG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// Pooled features -> the highway GEMM's B operand in fragment order (k_dis_highway_frag): piece
// ((seq/32 * KB + f/16) * 2 + part) * 64 + lane, lane = seq%32 + 32*((f/8)&1), holding features 8*(f/8) .. +7 of that
// sequence as 8 halves (part 0: hi = fp16(x*sP), part 1: lo = fp16(x*sP - hi)).  The conv epilogue holds 4 consecutive
// features per lane (f % 4 == 0): half a piece, one 8-byte store per part.
__device__ __forceinline__ void pf_store4(_Float16* pf, int KB, int seq, int f, const float (&o)[4], float sP) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = o[e] * sP;
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    // (sequence block, k block) fits 32 bits: the piece array (2 KiB per block) is a workspace in HBM
    const unsigned blk = (unsigned)(seq >> 5) * (unsigned)KB + (unsigned)(f >> 4);
    const unsigned in_blk = (unsigned)(seq & 31) + 32u * ((f >> 3) & 1);
    f16x4* d = reinterpret_cast<f16x4*>(pf + ((size_t)blk << 10) + in_blk * 8) + ((f >> 2) & 1);
    d[0] = h;
    d[128] = l;  // the lo piece: 64 pieces = 128 f16x4 further
}
__device__ __forceinline__ void pf_store1(_Float16* pf, int KB, int seq, int f, float v, float sP) {
    const float x = v * sP;
    const _Float16 h = (_Float16)x;
    const size_t piece = (((size_t)(seq >> 5) * KB + (f >> 4)) * 2) * 64 + (seq & 31) + 32 * ((f >> 3) & 1);
    _Float16* d = pf + piece * 8 + (f & 7);
    d[0] = h;
    d[512] = (_Float16)(x - (float)h);
}

// Max over the lanes n, n+S, n+2S, ... of each 32-lane half (S a power of two), left in every one of them, on the VALU
// alone: rotations inside the 16-lane rows (DPP row_ror) and, for the two rows of a half, v_permlane16_swap of the
// register with itself (one result holds row 0 | row 0 | row 2 | row 2, the other row 1 | row 1 | row 3 | row 3).
// __shfl_xor would go through the LDS crossbar, which the partner wave's operand reads are using.
template <int CTRL>
__device__ __forceinline__ void row_ror_max16(float (&x)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        x[r] = __builtin_fmaxf(x[r], __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x[r]), CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ void group_max16(float (&x)[16], int S) {
    if (S <= 1) row_ror_max16<0x121>(x);
    if (S <= 2) row_ror_max16<0x122>(x);
    if (S <= 4) row_ror_max16<0x124>(x);
    if (S <= 8) row_ror_max16<0x128>(x);
    if (S <= 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned u = __float_as_uint(x[r]);
            const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            x[r] = __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
    }
}

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
    // Workgroup -> sequence group.  Prefix mode with 8 | B/S: workgroup ids are dealt round-robin to the 8 XCDs, so XCD x
    // walks the groups (g, sentence block x + 8 j, rollout index i) with i innermost -- the R workgroups that read the
    // same rows of the prefix table (and gather the same prefix embeddings) run back to back in one L2 -- and g ascending,
    // i.e. the workgroups with the most fresh windows first.  The order changes nothing in the results.
    int grp = blockIdx.x;
    if (MODE == kConvPrefix && gridDim.y == 1) {
        const int nb = a.B >> log2S;              // sentence blocks per (i, g)
        const int RLnb = (int)gridDim.x;          // = R * L * nb when the remap applies
        if ((nb << log2S) == a.B && (nb & 7) == 0 && RLnb % (L * nb) == 0 && (size_t)RLnb * S == (size_t)a.N) {
            const int R = RLnb / (L * nb), x = grp & 7;
            int slot = grp >> 3;                  // this XCD's slot: (g, j, i), i fastest
            const int i = slot % R; slot /= R;
            const int j = slot % (nb >> 3), g = slot / (nb >> 3);
            grp = (i * L + g) * nb + x + 8 * j;
        }
    }
    const int seq0 = grp * S;

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
                tok[j] = t < a.valid_len ? (int)a.seqs[(size_t)seq * L + t] : 0;
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

    const float sP = (SPLIT && a.pfrag) ? a.scales[5] : 1.f;   // scale of the pooled features' (hi, lo) halves
    const long long st1 = tg_clock();
    long long st_mfma = 0, st_fetch = 0, st_epi = 0;
    int st_items = 0;
    for (;;) {
        const long long st_top = a.stamps ? tg_clock() : 0;
        int item = 0;
        if (lane == 0) item = atomicAdd(ctr, 1);
        // gridDim.y > 1 (few sequences: the sentence pass): the items are dealt round-robin to the y workgroups
        item = __builtin_amdgcn_readfirstlane(item) * (int)gridDim.y + (int)blockIdx.y;
        if (item >= a.n_items) break;
        const long long st_item = tg_clock();
        if (a.stamps) st_fetch += st_item - st_top;
        const int k = a.items[item].k, mt = a.items[item].mt;
        const bool merged = SPLIT && k == 0xFF;  // mt = merged tile index
        const int f = merged ? a.mfmax[mt] : a.w.f[k];
        const int nk = merged ? 32 : a.w.n[k];
        const int K = f * kEd, T = L - (merged ? a.mfmin[mt] : f) + 1;
        const int Tv = (MODE == kConvPlain) ? tg_min(T, a.valid_len) : T;   // windows evaluated here: t < Tv
        const int arow = tg_min(mt * 32 + n, nk - 1);
        const float* wp = merged ? nullptr : a.w.conv_w[k] + (size_t)arow * K + 32 * hh;
        const MergedMeta* mm = merged ? a.mmeta + mt : nullptr;
        auto mrow = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };  // tile row of accumulator register r
        // windows t <= g - f lie inside the shared prefix (Prefix mode): tiles start at the first fresh window of the
        // workgroup's smallest given_num, not at a multiple of the tile's time span
        const int t_org = (MODE == kConvPrefix) ? tg_max(0, g_min - f + 1) : 0;
        const int ntile = tg_max(0, Tv - t_org + TT - 1) >> (5 - log2S);  // ceil((Tv - t_org) / TT), 0 if none is fresh

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
                const bool ok = (nt < nbc) && (tw < Tv);
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
#if defined(TG_CONV_HALF_B)   // timing ablation only (wrong results): every second tile reuses its neighbour's operand
                        const int ntl = nt & ~1;
#else
                        const int ntl = nt;
#endif
                        const f16x8 bh = *reinterpret_cast<const f16x8*>(ebuf_h + boff[ntl] + toff + 16 * ks);
                        const f16x8 bl = *reinterpret_cast<const f16x8*>(ebuf_h + boff[ntl] + toff + kEd + 16 * ks);
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
        const long long st_mend = a.stamps ? tg_clock() : 0;
        if (a.stamps) { st_mfma += st_mend - st_item; ++st_items; }
        if (MODE == kConvStoreAll) continue;
        // Plain banks: after the butterfly below every lane of a sequence holds all 16 maxima, so its TT lanes split the
        // four 4-filter quads between them (D reg r -> filter mt*32 + (r&3) + 8*(r>>2) + 4*hh; lane tsub takes quads
        // tsub, tsub + nq, ...): one pass of the store body per wave instead of four on a quarter of the lanes, and
        // its loads (bias, prefix maxima) are issued before the butterfly.  (Tried: issuing them during the item's last
        // MFMA stage into the idle weight-fragment registers -- the epilogue shrank by 15 % but the stage loop grew by
        // 8 %, spills; not kept.)
        const int nq = TT < 4 ? TT : 4;
        const bool act = !merged && tsub < nq && seq0 + s < a.N;
        const int fo = merged ? 0 : a.foff[k];
        const float* pb = merged ? nullptr : a.w.conv_b[k];
        // Prefix mode: max over the windows t <= g - f (bias included) of the base sentence
        // Plain mode with windows left out (all padding): their value, the same for every sequence
        const float* pm = (!merged && MODE == kConvPrefix && g_lane >= f)
                              ? a.cv + ((size_t)b_lane * L + (g_lane - f)) * a.w.F + fo
                              : (!merged && MODE == kConvPlain && a.padpool && Tv < T) ? a.padpool + fo : nullptr;
        const bool aligned = ((fo | a.w.F) & 3) == 0 && (reinterpret_cast<uintptr_t>(pb) & 15) == 0;
        auto quad_vec = [&](int rq) __attribute__((always_inline)) {   // this lane's quad rq goes the 16-byte way
            return act && aligned && mt * 32 + 8 * rq + 4 * hh + 4 <= nk;
        };
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, pv = {0.f, 0.f, 0.f, 0.f};
        auto load_quad = [&](int rq, f32x4& b4, f32x4& p4) __attribute__((always_inline)) {
            const int m0 = mt * 32 + 8 * rq + 4 * hh;
            if (quad_vec(rq)) {
                b4 = *reinterpret_cast<const f32x4*>(pb + m0);
                if (pm) p4 = *reinterpret_cast<const f32x4*>(pm + m0);
            }
        };
        auto finish_quad = [&](int rq, bool vq) __attribute__((always_inline)) {
            const int m0 = mt * 32 + 8 * rq + 4 * hh;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // two-level select on the bits of rq (a chain of == tests becomes an indexed
                                           // load and sends runmax to scratch memory)
                const float lo2 = (rq & 1) ? runmax[4 + e] : runmax[e], hi2 = (rq & 1) ? runmax[12 + e] : runmax[8 + e];
                v[e] = (rq & 2) ? hi2 : lo2;
            }
            if (vq) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o4[e] = __builtin_fmaxf(v[e] + bv[e], 0.f);
                    if (pm) o4[e] = __builtin_fmaxf(o4[e], pv[e]);
                }
                if (SPLIT && a.pfrag) pf_store4(a.pfrag, a.KBp, seq0 + s, fo + m0, o4, sP);
                else *reinterpret_cast<f32x4*>(a.pooled + (size_t)(seq0 + s) * a.w.F + fo + m0) = f32x4{o4[0], o4[1], o4[2], o4[3]};
            } else if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m0 + e < nk) {
                        float o = __builtin_fmaxf(v[e] + pb[m0 + e], 0.f);
                        if (pm) o = __builtin_fmaxf(o, pm[m0 + e]);
                        if (SPLIT && a.pfrag) pf_store1(a.pfrag, a.KBp, seq0 + s, fo + m0 + e, o, sP);
                        else a.pooled[(size_t)(seq0 + s) * a.w.F + fo + m0 + e] = o;
                    }
            }
        };
        const bool vq0 = quad_vec(tsub & 3);
        if (!merged) load_quad(tsub & 3, bv, pv);
        // combine the TT time sub-positions of a sequence (lanes n, n+S, n+2S, ...)
        group_max16(runmax, S);
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
                    if (MODE == kConvPlain && a.padpool && a.valid_len < L - fwr + 1)
                        o = __builtin_fmaxf(o, a.padpool[ft]);
                    if (SPLIT && a.pfrag) pf_store1(a.pfrag, a.KBp, seq0 + s, ft, o, sP);
                    else po[ft] = o;
                }
            }
            if (a.stamps) st_epi += tg_clock() - st_mend;
            continue;
        }
        finish_quad(tsub & 3, vq0);
        for (int j = nq; j < 4; j += nq) {   // TT < 4: the lane's further quads
            load_quad((tsub & 3) + j, bv, pv);
            finish_quad((tsub & 3) + j, quad_vec((tsub & 3) + j));
        }
        if (a.stamps) st_epi += tg_clock() - st_mend;
    }
    if (a.stamps && lane == 0) {
        const long long st2 = tg_clock();
        long long* o = a.stamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
        o[0] = st1 - st0; o[1] = st2 - st1; o[2] = st_mfma; o[3] = st_items; o[4] = st_fetch; o[5] = st_epi;
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

// workgroup bx of nb: |max| of x[0..n) as ordered bits (NaN sorts high: scale 1 below)
__device__ __forceinline__ unsigned absmax_part(const float* x, size_t n, int bx, int nb) {
    unsigned m = 0;
    size_t done = 0;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // 16 bytes per lane and load while the tensor allows it
        const size_t n4 = n >> 2;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
        }
        done = n4 << 2;
    }
    for (size_t i = done + (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nb * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    return m;
}

__device__ void make_scales_conv(const unsigned* absmax, float* scales) {
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
__device__ void split_weights_part(const SplitArgs& a, int y, int bx, int nb) {
    const float sc = a.scales[0];
    if (y >= a.n_filters) {  // a merged tile: [tau < fmax][8 pieces][64 lanes], zero past a row's width
        const int m = y - a.n_filters;
        const MergedTile& mt = a.merged[m];
        const int total = mt.fmax * 8 * 64;
        f16x8* dst = reinterpret_cast<f16x8*>(a.w16 + a.mwoff[m]);
        for (int i = bx * 256 + threadIdx.x; i < total; i += nb * 256) {
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
        if (bx == 0 && threadIdx.x < 32) {
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
    const int k = y;
    const int f = a.f[k], nk = a.nk[k], K = f * kEd;
    const int total = ((a.nfull[k] + 31) / 32) * f * 8 * 64;
    f16x8* dst = reinterpret_cast<f16x8*>(a.w16 + a.woff[k]);
    for (int i = bx * 256 + threadIdx.x; i < total; i += nb * 256) {
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

// ---- highway GEMM on fragment-ordered split-fp16 operands (r02): the structure of the generator's vocabulary GEMM
// (tg_genw.hip k_genw_gemm).  A = W_hw as pieces [feature/32][KB][hi|lo][lane] (k_hw_make_frag, once per call),
// B = the pooled features as pieces [seq/32][KB][hi|lo][lane] written by the conv epilogue itself (pf_store4): no
// operand is transposed, converted or padded inside the GEMM, every stage is a linear copy made by LDS-DMA
// (global_load_lds_dwordx4, no staging registers), every ds_read_b128 a conflict-free 1-KiB row.  K = F padded to a
// multiple of 32 (zero pieces), stages of 32 through two LDS buffers, one barrier per stage.  Workgroups are
// persistent and walk their XCD's tile list (8 row tiles per A tile group), so the W_hw slab of a feature tile is
// fetched once per XCD pass instead of once per 128 sequences.
// Epilogue per (32 features x 32 sequences) accumulator: h = acc/(sW sP) + b; gate sigma(h) relu(h) + (1 - sigma(h))
// pooled with pooled = (hi + lo)/sP read back from the B pieces (8 bytes per lane and 4 features); dropout;
// partial feature2out dot, summed over the wave's features in a fixed order -> partial[seq][chunk][o].
struct Hw2Args {
    HwArgs h;            // weights, dropout, outputs (h.pooled unused)
    const f32x4* Wf;     // [MB][KB][2][64]
    const f32x4* Pf;     // [RB][KB][2][64]
    int KB;              // 16-wide k blocks (even)
    int MT, RT, NB;      // feature tiles, sequence tiles of the workgroup shape; logical tiles 8 * MT * ceil(RT / 8)
    const float* scales; // [4] sW, [5] sP, [6] 1/(sW sP), [7] 1/sP
};

__device__ __forceinline__ void hw_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The epilogue's constants -- the highway bias and the output layer's rows of the tile's features -- are staged per
// tile in LDS behind the two operand stages (two buffers of [1 + n_out][32 MBT] floats, zero past F, written during
// the tile's first K step): read per quad from L2 each one is a dependent ~1 us round trip with the matrix pipes idle.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_dis_highway_frag(const Hw2Args a) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int MBT = WM * TM, RBT = WN * TN;
    constexpr int NLA = MBT * 256 / NTHR, NLB = RBT * 256 / NTHR;
    constexpr int STAGE = (MBT + RBT) * 256;
    constexpr int BM = 32 * MBT;
    constexpr int SPT = ((1 + kMaxOut) * BM + NTHR - 1) / NTHR;   // stash floats per thread, at most
    extern __shared__ __attribute__((aligned(16))) f32x4 hlds[];
    float* stash0 = reinterpret_cast<float*>(hlds + 2 * STAGE);
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wm = w / WN, wn = w % WN;
    const int per_xcd = a.NB >> 3, stride = gridDim.x >> 3;
    const int lb_end = ((int)blockIdx.x & 7) * per_xcd + per_xcd;
    const int per = 8 * a.MT;
    auto decode = [&](int lb, int& mb0, int& rb0) {
        const int rtg = lb / per, rem = lb - rtg * per;
        const int rt = rtg * 8 + (rem & 7);
        mb0 = MBT * (rem >> 3);
        rb0 = RBT * rt;
        return rt < a.RT;
    };
    auto next_valid = [&](int lb, int& mb0, int& rb0) {
        while (lb < lb_end && !decode(lb, mb0, rb0)) lb += stride;
        return lb;
    };
    int mb0 = 0, rb0 = 0;
    int lb = next_valid(((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3), mb0, rb0);
    if (lb >= lb_end) return;
    const int S = a.KB >> 1;
    const int pw = tid & 255;
    auto load_stage = [&](int mb, int rb, int s, int buf) {
        f32x4* d = hlds + buf * STAGE + tid;
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            hw_glds16(a.Wf + ((size_t)(mb + ((tid + i * NTHR) >> 8)) * a.KB + 2 * s) * 128 + pw, d + i * NTHR);
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            hw_glds16(a.Pf + ((size_t)(rb + ((tid + i * NTHR) >> 8)) * a.KB + 2 * s) * 128 + pw, d + MBT * 256 + i * NTHR);
    };
    f32x16 acc[TM][TN];
    load_stage(mb0, rb0, 0, 0);
    __syncthreads();
    int gs = 0, tile_no = 0;
    const int stash_n = (1 + a.h.w.n_out) * BM;   // floats per stash buffer
    for (;; ++tile_no) {
        int nmb0 = 0, nrb0 = 0;
        const int nlb = next_valid(lb + stride, nmb0, nrb0);
        const bool has_next = nlb < lb_end;
        float* stash = stash0 + (tile_no & 1) * stash_n;
        float sv[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {   // row 0: hw_b, row 1 + o: out_w[o]; features 32 mb0 .. + BM - 1
            const int i = tid + j * NTHR;
            const int row = i / BM, fidx = 32 * mb0 + (i - row * BM);
            sv[j] = 0.f;
            if (i < stash_n && fidx < a.h.w.F)
                sv[j] = row == 0 ? a.h.w.hw_b[fidx] : a.h.w.out_w[(size_t)(row - 1) * a.h.w.F + fidx];
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int s = 0; s < S; ++s, ++gs) {
            if (s + 1 < S) load_stage(mb0, rb0, s + 1, (gs + 1) & 1);
            else if (has_next) load_stage(nmb0, nrb0, 0, (gs + 1) & 1);
            const f32x4* la = hlds + (gs & 1) * STAGE + (TM * wm) * 256 + lane;
            const f32x4* lbp = hlds + (gs & 1) * STAGE + MBT * 256 + (TN * wn) * 256 + lane;
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl) {
                f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    bh[i] = __builtin_bit_cast(f16x8, lbp[i * 256 + kbl * 128]);
                    bl[i] = __builtin_bit_cast(f16x8, lbp[i * 256 + kbl * 128 + 64]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = __builtin_bit_cast(f16x8, la[i * 256 + kbl * 128]);
                    al[i] = __builtin_bit_cast(f16x8, la[i * 256 + kbl * 128 + 64]);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = TG_MFMA16(al[mi], bh[ni], acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = TG_MFMA16(ah[mi], bl[ni], acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = TG_MFMA16(ah[mi], bh[ni], acc[mi][ni]);
                __builtin_amdgcn_s_setprio(0);
            }
            if (s == 0) {   // the tile's stash: read by its epilogue, many barriers from here; the other buffer is the
                            // previous tile's, which slower waves may still be reading
#pragma unroll
                for (int j = 0; j < SPT; ++j)
                    if (tid + j * NTHR < stash_n) stash[tid + j * NTHR] = sv[j];
            }
            __syncthreads();
        }

        // ---- epilogue of this wave: feature blocks mbw .. mbw+TM-1, sequence blocks rbw .. rbw+TN-1.
        // Branch-free per quad and software-pipelined over the (sequence block, 64-feature slot) groups: the pooled
        // halves of group g+1 (global, long evicted from L2 by the K loop's stream) are in flight while group g is
        // computed; the highway bias and the output layer come from the tile's LDS stash.  The arithmetic and its
        // order are those of the first version of this kernel: results are bit-identical.
        {
            const int F = a.h.w.F, n_out = a.h.w.n_out;
            const float inv = a.scales[6], invP = a.scales[7];
            const int mbw = mb0 + TM * wm, rbw = rb0 + TN * wn;
            // partial slots are per 64 features whatever the tile shape: shards of a call (other N, maybe the other
            // tile shape) then sum their logits in the same order and stay bit-identical to the unsharded call
            static_assert(TM % 2 == 0, "two feature blocks per partial slot");
            constexpr int NG = TN * (TM / 2);
            f16x4 ph[2][8], pl[2][8];
            auto fetch = [&](int g, f16x4 (&h8)[8], f16x4 (&l8)[8]) {
                const int ni = g / (TM / 2), sl = g % (TM / 2);
                const int rb = rbw + ni;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int mb = mbw + 2 * sl + (it >> 2), rq = it & 3;
                    // pooled[seq][fb .. fb+3] from the B pieces: k block 2*mb + (rq >> 1), lane n + 32*(rq & 1), half hh
                    const int kb = tg_min(2 * mb + (rq >> 1), a.KB - 1);   // blocks past F: any piece, the value is dropped
                    const _Float16* pp = reinterpret_cast<const _Float16*>(a.Pf) +
                                         ((((size_t)rb * a.KB + kb) * 2) * 64 + n + 32 * (rq & 1)) * 8 + 4 * hh;
                    h8[it] = *reinterpret_cast<const f16x4*>(pp);
                    l8[it] = *reinterpret_cast<const f16x4*>(pp + 512);
                }
            };
            auto group = [&](int g, const f16x4 (&h8)[8], const f16x4 (&l8)[8]) {
                const int ni = g / (TM / 2), sl = g % (TM / 2);
                const int rb = rbw + ni;
                const int seq = rb * 32 + n;
                const bool seq_ok = seq < a.h.N;
                float part[kMaxOut];
#pragma unroll
                for (int o = 0; o < kMaxOut; ++o) part[o] = 0.f;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int mi = 2 * sl + (it >> 2), rq = it & 3;
                    const int fb = 32 * (mbw + mi) + 8 * rq + 4 * hh;   // 4 consecutive features (F % 8 == 0: all in or all out)
                    const bool f_ok = fb < F;
                    const int fl = 32 * (TM * wm + mi) + 8 * rq + 4 * hh;   // feature index inside the tile
                    const f32x4 hb = *reinterpret_cast<const f32x4*>(stash + fl);
                    float fv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pooled = ((float)h8[it][e] + (float)l8[it][e]) * invP;
                        const float h = __builtin_fmaf(acc[mi][ni][4 * rq + e], inv, hb[e]);
                        const float sg = tg_sigmoid(h);
                        fv[e] = sg * __builtin_fmaxf(h, 0.f) + (1.f - sg) * pooled;
                    }
                    const size_t o4 = (size_t)seq * F + fb;
                    if (a.h.feature_out && seq_ok && f_ok)
                        *reinterpret_cast<f32x4*>(a.h.feature_out + o4) = f32x4{fv[0], fv[1], fv[2], fv[3]};
                    if (a.h.dropout_u) {
                        if (seq_ok && f_ok) {
                            const f32x4 u = *reinterpret_cast<const f32x4*>(a.h.dropout_u + o4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) fv[e] = (u[e] < a.h.keep) ? fv[e] * a.h.scale : 0.f;
                        }
                    } else if (a.h.drop_gen) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            fv[e] = tg_drop_keep((uint32_t)seq * (uint32_t)F + (uint32_t)(fb + e), a.h.drop_seed, a.h.keep_u32)
                                        ? fv[e] * a.h.scale : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) fv[e] = f_ok ? fv[e] : 0.f;   // features past F add nothing
#pragma unroll
                    for (int o = 0; o < kMaxOut; ++o) {
                        if (o < n_out) {
                            const f32x4 ow = *reinterpret_cast<const f32x4*>(stash + (o + 1) * BM + fl);
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[o] = __builtin_fmaf(fv[e], ow[e], part[o]);
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < kMaxOut; ++o) part[o] += __shfl_xor(part[o], 32, 64);   // the two feature halves of a row
                if (hh == 0 && seq_ok) {
                    float* po = a.h.partial + ((size_t)seq * a.h.nchunk + (mbw >> 1) + sl) * kMaxOut;
#pragma unroll
                    for (int o = 0; o < kMaxOut; ++o)
                        if (o < n_out) po[o] = part[o];
                }
            };
            fetch(0, ph[0], pl[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) fetch(g + 1, ph[(g + 1) & 1], pl[(g + 1) & 1]);
                group(g, ph[g & 1], pl[g & 1]);
            }
        }
        if (!has_next) break;
        lb = nlb;
        mb0 = nmb0;
        rb0 = nrb0;
    }
}

// W_hw -> A pieces: piece ((mb*KB + kb)*2 + part)*64 + lane = W_hw[32mb + lane%32][16kb + 8(lane/32) + 0..7] * sW as
// 8 halves (hi or lo); rows / columns past F are zero.
__device__ void hw_make_frag_part(const float* __restrict__ hw_w, int F, int MB, int KB, const float* scales,
                                  f16x8* __restrict__ Wf, int bx, int nb) {
    const float sW = scales[4];
    const long total = (long)MB * KB * 128;
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nb * 256) {
        const int lane = (int)(i & 63), part = (int)((i >> 6) & 1);
        const long q = i >> 7;
        const int kb = (int)(q % KB), mb = (int)(q / KB);
        const int row = 32 * mb + (lane & 31), k0 = 16 * kb + 8 * (lane >> 5);
        f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < F) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (k0 + e < F) {
                    const float x = hw_w[(size_t)row * F + k0 + e] * sW;
                    const _Float16 h = (_Float16)x;
                    o[e] = part ? (_Float16)(x - (float)h) : h;
                }
            }
        }
        Wf[i] = o;
    }
}

// pieces of the pooled fragments whose features lie past F (the k padding up to a multiple of 32): zero, so that the
// GEMM meets 0 . 0 there and not whatever the workspace held
__device__ void pf_zero_pad_part(f16x8* __restrict__ Pf, int RB, int KB, int F, int bx, int nb) {
    const int kb_first = F >> 4;   // first k block with a feature >= F (F % 8 == 0)
    const int nkb = KB - kb_first;
    const long total = (long)RB * nkb * 128;
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nb * 256) {
        const int lane = (int)(i & 63), part = (int)((i >> 6) & 1);
        const long q = i >> 7;
        const int kb = kb_first + (int)(q % nkb);
        const long rb = q / nkb;
        if (16 * kb + 8 * (lane >> 5) >= F) Pf[((rb * KB + kb) * 2 + part) * 64 + lane] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}


// Scales of the fragment highway, all known before the conv runs: sW from |max| of W_hw (absmax[2]); sP from a BOUND
// on the pooled features, |conv| <= 64 f_max max|w| max|e| + max|b| (absmax[0], [1], [3]) -- the conv epilogue splits
// its outputs with it, so it cannot wait for their actual maximum.  fp16 is a floating format: an over-estimate by
// 2^10 still leaves the (hi, lo) pair 22 significant bits for every feature within 2^-6 of the largest.
__device__ void make_scales_hw(const unsigned* absmax, float* scales, int fmax) {
    auto pow2_for = [](float m) {
        int ex = 0;
        float sc = 1.f;
        if (m > 0.f && m < __builtin_inff()) {
            (void)frexpf(m, &ex);
            sc = ldexpf(1.f, 14 - ex);
        }
        return sc;
    };
    const float sW = pow2_for(__uint_as_float(absmax[2]));
    const float bound = 64.f * (float)fmax * __uint_as_float(absmax[0]) * __uint_as_float(absmax[1]) + __uint_as_float(absmax[3]);
    const float sP = pow2_for(bound);
    scales[4] = sW;
    scales[5] = sP;
    scales[6] = 1.f / (sW * sP);
    scales[7] = 1.f / sP;
}

// ---- the split-fp16 preparation of a call in three launches (it is a dependent chain in front of the conv: ten
// launches of a few microseconds each cost more in the gaps between them than in the kernels)
struct PrepArgs {
    SplitArgs s;
    const float* emb;
    size_t n_emb;
    const float* hw_w;     // null: no fragment highway (its |max|, scales, pieces and padding are skipped)
    size_t n_hw;
    unsigned* absmax;      // [0] conv weights, [1] embedding table, [2] W_hw, [3] conv biases; zeroed by the host
    float* scales;
    int fmax, F, MB, KB, RB;
    f16x8* Wf;
    f16x8* Pf;
};

// blockIdx.y: conv bank k | n_filters: embedding table | n_filters + 1: conv biases | n_filters + 2: W_hw
__global__ __launch_bounds__(256) void k_prep_absmax(const PrepArgs a) {
    const int y = blockIdx.y, nf = a.s.n_filters, bx = blockIdx.x, nb = gridDim.x;
    unsigned m = 0;
    unsigned* out = a.absmax;
    if (y < nf) {
        m = absmax_part(a.s.src[y], (size_t)a.s.nk[y] * a.s.f[y] * kEd, bx, nb);
    } else if (y == nf) {
        m = absmax_part(a.emb, a.n_emb, bx, nb);
        out += 1;
    } else if (y == nf + 1) {
        if (bx == 0)
            for (int k = 0; k < nf; ++k)
                for (int i = threadIdx.x; i < a.s.nk[k]; i += 256) m = max(m, __float_as_uint(a.s.bias[k][i]) & 0x7fffffffu);
        out += 3;
    } else {
        m = absmax_part(a.hw_w, a.n_hw, bx, nb);
        out += 2;
    }
    block_atomic_max(m, out);
}

__global__ void k_prep_scales(const PrepArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    make_scales_conv(a.absmax, a.scales);
    if (a.hw_w) make_scales_hw(a.absmax, a.scales, a.fmax);
}

// blockIdx.y: conv bank / merged tile | then 4 slices of W_hw's pieces | then the pooled pieces' k padding
__global__ __launch_bounds__(256) void k_prep_split(const PrepArgs a) {
    const int y = blockIdx.y, nc = a.s.n_filters + a.s.n_merged, bx = blockIdx.x, nb = gridDim.x;
    if (y < nc) split_weights_part(a.s, y, bx, nb);
    else if (y < nc + 4) hw_make_frag_part(a.hw_w, a.F, a.MB, a.KB, a.scales, a.Wf, (y - nc) * nb + bx, 4 * nb);
    else pf_zero_pad_part(a.Pf, a.RB, a.KB, a.F, bx, nb);
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
    // the kernels read the matrices 16 bytes at a time (include/textgan_hip.h, "Alignment")
    uintptr_t al = reinterpret_cast<uintptr_t>(w->emb) | reinterpret_cast<uintptr_t>(w->hw_w);
    for (int k = 0; k < w->n_filters; ++k) al |= reinterpret_cast<uintptr_t>(w->conv_w[k]);
    if ((al & 15) || ((reinterpret_cast<uintptr_t>(w->hw_b) | reinterpret_cast<uintptr_t>(w->out_w) |
                       reinterpret_cast<uintptr_t>(w->out_b)) & 3)) {
        tg_set_error("discriminator weights: embeddings, conv and highway matrices must be 16-byte aligned");
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

int prepare_conv(ConvArgs& ca, const TgDisWeights* w, int N, int L, size_t* smem, int min_S = 1, int valid_len = -1) {
    const int vl = valid_len < 0 ? L : valid_len;   // windows per width: min(L - f + 1, vl)
    ca.w = *w;
    ca.L = L;
    ca.stamps = tg_diag_pointer("TG_CONV_STAMPS");
    const MergePlan pl = plan_merge(w);
    const int LP = L + pl.pad_rows;  // LDS rows per sequence
    int S = 32, log2S = 5;
    while (S > 1 && (size_t)LP * S * kERow * sizeof(float) + 64 > 144 * 1024) { S >>= 1; --log2S; }
    // few sequences: fewer per workgroup so that the grid fills the chip -- down to min_S (1 for the base sentences of
    // the prefix pass; 8 where the pooled features are wanted: below that every workgroup streams all the weights for
    // a handful of columns and the tiles are mostly time padding, so launch_conv_t widens the grid over the (width,
    // tile) items instead)
    while (S > min_S && (N + S - 1) / S < 512) { S >>= 1; --log2S; }
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
            tmp[ni].k = k; tmp[ni].mt = mt; tmp[ni].cost = (long)w->f[k] * tg_min_host(L - w->f[k] + 1, vl);
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
        tmp[ni].k = 0xFF; tmp[ni].mt = m; tmp[ni].cost = (long)pl.tile[m].fmax * tg_min_host(L - pl.tile[m].fmin + 1, vl);
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
    // a grid of fewer workgroups than CUs is widened over the (width, tile) items: the y workgroups of a sequence group
    // gather the same embeddings and deal the items between them (every item writes features of its own)
    const int gx = (ca.N + ca.S - 1) / ca.S;
    const int gy = gx < 256 ? tg_min_host(tg_min_host(8, ca.n_items), (512 + gx - 1) / gx) : 1;
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

// fragment highway geometry: k blocks of 16 (an even count), feature blocks of 32 padded to the 256-wide tile, sequence
// blocks of 32 padded to 256 sequences
inline int hw_kb(int F) { return 2 * ((F + 31) / 32); }
inline int hw_mb(int F) { return ((F + 255) / 256) * 8; }
inline size_t hw_rb(int N) { return ((size_t)N + 255) / 256 * 8; }
inline size_t pooled_bytes(const TgDisWeights* w, int N) {   // fp32 [N,F] or the B pieces, whichever is larger
    const size_t a = sizeof(float) * (size_t)N * w->F, b = hw_rb(N) * hw_kb(w->F) * 2 * 1024;
    return a > b ? a : b;
}
inline size_t hw_weight_bytes(const TgDisWeights* w) {        // (hi, lo) halves of W_hw as A pieces
    return (size_t)hw_mb(w->F) * hw_kb(w->F) * 2 * 1024;
}
inline int hw_nchunk(const TgDisWeights* w) {                  // partial-logit slots per sequence (either highway kernel)
    const int a = (w->F + kHwTile - 1) / kHwTile, b = hw_mb(w->F) / 2;   // b: one slot per 64 features
    return a > b ? a : b;
}

// the part of a workspace that belongs to one call: pooled features (or their pieces), partial logits, prefix table
size_t call_bytes(const TgDisWeights* w, int N, int L, int B_prefix) {
    return align256(pooled_bytes(w, N)) + align256(sizeof(float) * (size_t)N * hw_nchunk(w) * kMaxOut) +
           align256(sizeof(float) * (size_t)B_prefix * L * w->F);
}
// the part that holds the prepared weights: conv pieces, scales, highway pieces; for tg_dis_features also the pooled
// features of an all-padding sequence and L padding tokens
size_t weights_bytes(const TgDisWeights* w, int L) {
    return align256(2 * sizeof(_Float16) * conv_weight_count(w)) + 256 + align256(hw_weight_bytes(w)) +
           align256(sizeof(float) * w->F) + align256(sizeof(int64_t) * L);
}
size_t ws_bytes(const TgDisWeights* w, int N, int L, int B_prefix) {
    return call_bytes(w, N, L, B_prefix) + weights_bytes(w, L);
}

__global__ __launch_bounds__(256) void k_pf_zero_pad(f16x8* Pf, int RB, int KB, int F) {
    pf_zero_pad_part(Pf, RB, KB, F, blockIdx.x, gridDim.x);
}

// Split-fp16 preparation (every call: the weights are live module parameters): |max| of the embedding table and
// of all conv banks, the two power-of-two scales, and the (hi, lo) halves of the conv weights.
// This function lays the pieces out (offsets into w16, the kernel arguments); launch_prepare fills them -- skipped when
// the workspace still holds what an earlier call wrote for the same weights (FeatOpts::reuse).
void prepare_split(ConvArgs& ca, const TgDisWeights* w, _Float16* w16, void* scal, SplitArgs& sa) {
    float* scales = (float*)((char*)scal + 16);  // {scale_w, scale_e, 1/(scale_w*scale_e), -} x 2 (conv, highway)
    sa = SplitArgs{};
    const MergePlan pl = plan_merge(w);
    int off = 0, foff = 0;
    for (int k = 0; k < w->n_filters; ++k) {
        sa.src[k] = w->conv_w[k];
        sa.bias[k] = w->conv_b[k];
        sa.woff[k] = off;
        sa.nk[k] = w->n[k];
        sa.nfull[k] = pl.bank_merged[k] ? (w->n[k] / 32) * 32 : w->n[k];
        sa.f[k] = w->f[k];
        sa.foff[k] = foff;
        foff += w->n[k];
        ca.woff[k] = off;
        off += (int)(conv_frag_halves(w, k) / 8) * 8;
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
    ca.w16 = w16;
    ca.scales = scales;
}

// The launches of the preparation: |max| of every tensor, the scales, the (hi, lo) pieces.  hw16 null: the conv's only.
int launch_prepare(const SplitArgs& sa, const TgDisWeights* w, void* scal, _Float16* hw16, void* pooled, int NL,
                   hipStream_t stream) {
    PrepArgs pa = {};
    pa.s = sa;
    pa.emb = w->emb;
    pa.n_emb = (size_t)w->V * kEd;
    pa.absmax = (unsigned*)scal;
    pa.scales = (float*)((char*)scal + 16);
    pa.fmax = 1;
    for (int k = 0; k < w->n_filters; ++k) pa.fmax = w->f[k] > pa.fmax ? w->f[k] : pa.fmax;
    if (hw16) {
        pa.hw_w = w->hw_w;
        pa.n_hw = (size_t)w->F * w->F;
        pa.F = w->F;
        pa.MB = hw_mb(w->F);
        pa.KB = hw_kb(w->F);
        pa.RB = (int)hw_rb(NL);
        pa.Wf = reinterpret_cast<f16x8*>(hw16);
        pa.Pf = reinterpret_cast<f16x8*>(pooled);
    }
    if (hipMemsetAsync(pa.absmax, 0, 16, stream) != hipSuccess) {
        tg_set_error("hipMemsetAsync failed");
        return TG_ERR_LAUNCH;
    }
    tg_prof_begin(TG_KERNEL_DIS_PREFIX, stream);
    hipLaunchKernelGGL(k_prep_absmax, dim3(256, w->n_filters + (hw16 ? 3 : 2)), dim3(256), 0, stream, pa);
    hipLaunchKernelGGL(k_prep_scales, dim3(1), dim3(64), 0, stream, pa);
    hipLaunchKernelGGL(k_prep_split, dim3(256, w->n_filters + sa.n_merged + (hw16 ? 5 : 0)), dim3(256), 0, stream, pa);
    tg_prof_end(stream);
    return TG_OK;
}

template <int WM, int WN, int TM, int TN>
int launch_highway_frag_t(Hw2Args& a, int MBp, int N, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr size_t sm0 = (size_t)2 * (WM * TM + WN * TN) * 256 * 16;
    // + two stash buffers of the epilogue's constants ([1 + n_out][BM] floats each): 18 KiB at most beside 128 KiB of
    // stages (256 x 256 tiles), 9 KiB beside 64 KiB (128 x 128: still two workgroups per CU)
    const size_t sm = sm0 + (size_t)2 * (1 + a.h.w.n_out) * BM * sizeof(float);
    static TgLdsAttr attr;
    if (int rc = tg_lds_attr(attr, reinterpret_cast<const void*>(&k_dis_highway_frag<WM, WN, TM, TN>), sm, "k_dis_highway_frag")) return rc;
    a.MT = MBp * 32 / BM;
    a.RT = (N + BN - 1) / BN;
    a.NB = (a.RT + 7) / 8 * 8 * a.MT;
    a.h.nchunk = MBp / 2;       // one partial slot per 64 features, whatever the tile shape
    const int resident = 256 * (int)((160 * 1024) / sm);
    const int grid = a.NB < resident ? a.NB : resident;
    tg_prof_begin(TG_KERNEL_DIS_HIGHWAY, stream);
    hipLaunchKernelGGL((k_dis_highway_frag<WM, WN, TM, TN>), dim3(grid), dim3(64 * WM * WN), sm, stream, a);
    tg_prof_end(stream);
    return TG_OK;
}

// 256 x 256 tiles (8 waves, 128 KiB LDS, one workgroup per CU) or 128 x 128 (4 waves, two workgroups per CU): whichever
// needs fewer passes over the chip.  In units of one 128 x 128 tile on a CU of its own: a big tile costs 4 x 0.93 (it
// moves half the operand bytes per flop), a small one 2 (its CU is shared); the chip holds 256 big or 512 small
// workgroups per pass.  C2 (52 224 sequences) takes the big tiles, the LeakGAN position loop (1 280) and SentiGAN's
// F = 800 scorer the small ones.
int launch_highway_frag(Hw2Args& a, int MBp, int N, hipStream_t stream) {
    const long big = (long)(MBp / 8) * ((N + 255) / 256), small = (long)(MBp / 4) * ((N + 127) / 128);
    const double cost_big = (double)((big + 255) / 256) * 4.0 * 0.93, cost_small = (double)((small + 511) / 512) * 2.0;
    if (cost_big <= cost_small) return launch_highway_frag_t<2, 4, 4, 2>(a, MBp, N, stream);
    return launch_highway_frag_t<2, 2, 2, 2>(a, MBp, N, stream);
}

// tg_dis_features' extras (all defaults: tg_dis_score)
struct FeatOpts {
    enum { kRun = 0, kPrepareOnly = 1, kPadOnly = 2 };
    bool reuse = false;          // the prepared weights are in place (same w, same N_layout)
    int valid_len = -1;          // >= 0: tokens at positions >= valid_len are padding (and the all-padding windows skipped)
    int N_layout = -1;           // sequences the workspace is laid out for (>= N)
    void* weights_ws = nullptr;  // the prepared weights live here (weights_bytes) and not behind the per-call part
    int mode = kRun;             // kPrepareOnly: weights, padpool, this workspace's k padding; kPadOnly: the k padding
};

// sentences == nullptr: plain scoring of N arbitrary sequences.
// sentences != nullptr: seqs is [R,L,B,L] rollout completions (N = R*L*B), prefix reuse.
int score_impl(const TgDisWeights* w, const int64_t* seqs, int N, int L, const int64_t* sentences, int B, int col,
               float dropout_p, const float* dropout_uniforms, float* reward_out, float* logits_out,
               float* feature_out, void* workspace, size_t workspace_bytes, hipStream_t stream,
               const FeatOpts& fo = FeatOpts()) {
    const bool reuse = fo.reuse;
    const int valid_len = fo.valid_len;
    const int NL = fo.N_layout < N ? N : fo.N_layout;   // sequences the workspace is laid out for
    if (N < 0 || L < 1 || (N > 0 && !seqs)) {   // an empty batch is a no-op
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
    const size_t need = call_bytes(w, NL, L, sentences ? B : 0) + (fo.weights_ws ? 0 : weights_bytes(w, L));
    if (!workspace || workspace_bytes < need) {
        tg_set_error("tg_dis_score: workspace too small (%zu < %zu)", workspace_bytes, need);
        return TG_ERR_WORKSPACE;
    }
    float* pooled = (float*)workspace;
    const int nchunk = (w->F + kHwTile - 1) / kHwTile;   // fp32 highway; the fragment highway sets its own count
    float* partial = (float*)((char*)workspace + align256(pooled_bytes(w, NL)));
    float* cv = (float*)((char*)partial + align256(sizeof(float) * (size_t)NL * hw_nchunk(w) * kMaxOut));
    _Float16* w16 = fo.weights_ws ? (_Float16*)fo.weights_ws
                                  : (_Float16*)((char*)cv + align256(sizeof(float) * (size_t)(sentences ? B : 0) * L * w->F));
    void* scal = (char*)w16 + align256(2 * sizeof(_Float16) * conv_weight_count(w));
    _Float16* hw16 = (_Float16*)((char*)scal + 256);
    float* padpool = (float*)((char*)hw16 + align256(hw_weight_bytes(w)));
    int64_t* padtok = (int64_t*)((char*)padpool + align256(sizeof(float) * w->F));

    // ---- conv
    ConvArgs ca = {};
    size_t sm = 0;
    rc = prepare_conv(ca, w, sentences ? B : N, L, &sm, sentences ? 1 : 8, valid_len);
    if (rc) return rc;
    ca.valid_len = valid_len < 0 ? L : valid_len;
    SplitArgs sa;
    if (w->precision == TG_DIS_SPLIT_F16) {
        prepare_split(ca, w, w16, scal, sa);
    } else if (w->precision != TG_DIS_FP32) {
        tg_set_error("tg_dis_score: unknown precision %d", w->precision);
        return TG_ERR_ARG;
    }
    // split arithmetic with F % 8 == 0: the pooled features leave the conv as the highway GEMM's B pieces
    // (its epilogue reads the highway bias, the output layer, the dropout uniforms and writes the features 16 bytes at
    // a time: tensors that are not 16-byte aligned -- views into a larger buffer -- take the exact-fp32 highway too)
    const uintptr_t al16 = reinterpret_cast<uintptr_t>(w->hw_b) | reinterpret_cast<uintptr_t>(w->out_w) |
                           reinterpret_cast<uintptr_t>(dropout_uniforms) | reinterpret_cast<uintptr_t>(feature_out);
    const bool frag = ca.w16 != nullptr && (w->F & 7) == 0 && (al16 & 15) == 0 && (size_t)N < ((size_t)1 << 31) - 256;
    const int KB = hw_kb(w->F), MBp = hw_mb(w->F);
    if (fo.mode == FeatOpts::kPadOnly) {   // another call workspace beside an already prepared one: its k padding
        if (frag)
            hipLaunchKernelGGL(k_pf_zero_pad, dim3(64), dim3(256), 0, stream, reinterpret_cast<f16x8*>(pooled),
                               (int)hw_rb(NL), KB, w->F);
        return TG_OK;
    }
    if (ca.w16 && !reuse) {
        rc = launch_prepare(sa, w, scal, frag ? hw16 : nullptr, pooled, NL, stream);
        if (rc) return rc;
    }
    if (frag) {
        ca.pfrag = reinterpret_cast<_Float16*>(pooled);
        ca.KBp = KB;
    }
    if (!sentences && valid_len >= 0) {
        // tg_dis_features: the windows that start at or behind valid_len are all padding and the same for every
        // sequence -- their pooled value comes from one pass over a single all-padding sequence (same kernel, same
        // arithmetic, so the features are the bits a pass over all windows gives), once per prepared workspace
        if (!reuse) {
            ConvArgs cp = ca;
            size_t smp = 0;
            rc = prepare_conv(cp, w, 1, L, &smp, 8);
            if (rc) return rc;
            if (hipMemsetAsync(padtok, 0, sizeof(int64_t) * L, stream) != hipSuccess) {
                tg_set_error("hipMemsetAsync failed");
                return TG_ERR_LAUNCH;
            }
            cp.seqs = padtok;
            cp.N = 1;
            cp.valid_len = L;
            cp.padpool = nullptr;
            cp.pfrag = nullptr;      // fp32 [1, F]
            cp.pooled = padpool;
            rc = launch_conv<kConvPlain>(cp, smp, stream);
            if (rc) return rc;
        }
        ca.padpool = padpool;
    }
    if (fo.mode == FeatOpts::kPrepareOnly && !sentences) return TG_OK;
    if (sentences) {
        // (1) every window of the B base sentences, (2) running max over time, (3) completions: fresh windows only
        if (!reuse) {
            ca.seqs = sentences;
            ca.N = B;
            ca.cv = cv;
            rc = launch_conv<kConvStoreAll>(ca, sm, stream);
            if (rc) return rc;
            tg_prof_begin(TG_KERNEL_DIS_PREFIX, stream);
            hipLaunchKernelGGL(k_dis_prefix_scan, dim3((B * w->F + 255) / 256), dim3(256), 0, stream, cv, B, L, w->F);
            tg_prof_end(stream);
        }
        if (fo.mode == FeatOpts::kPrepareOnly) {   // everything that needs the weights and the sentences only
            hipError_t e0 = hipGetLastError();
            if (e0 != hipSuccess) {
                tg_set_error("tg_dis_score_rollout_prepare launch: %s", hipGetErrorString(e0));
                return TG_ERR_LAUNCH;
            }
            return TG_OK;
        }
        rc = prepare_conv(ca, w, N, L, &sm, 8);
        if (rc) return rc;
        ca.valid_len = L;
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
    if (frag) {
        Hw2Args h2 = {};
        h2.h = ha;
        h2.Wf = reinterpret_cast<const f32x4*>(hw16);
        h2.Pf = reinterpret_cast<const f32x4*>(pooled);
        h2.KB = KB;
        h2.scales = (const float*)((char*)scal + 16);
        rc = launch_highway_frag(h2, MBp, N, stream);
        if (rc) return rc;
        nchunk_used = h2.h.nchunk;
    } else {
        tg_prof_begin(TG_KERNEL_DIS_HIGHWAY, stream);
        hipLaunchKernelGGL(k_dis_highway, dim3(ha.nseqtile * nchunk), dim3(kHwThreads), 0, stream, ha);
        tg_prof_end(stream);
    }

    // ---- final: fixed-order partial sum, softmax column (nothing to write when only the features are asked for)
    if (!reward_out && !logits_out) {
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) {
            tg_set_error("tg_dis_score launch: %s", hipGetErrorString(e0));
            return TG_ERR_LAUNCH;
        }
        return TG_OK;
    }
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
    if (!w || !w->out_w || !w->out_b || (N > 0 && (!feature || !reward_out)) || N < 0 || R < 1) {
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

// tg_dis_score in two parts, like tg_dis_score_rollout: what needs only the weights (|max|, scales, pieces), and the rest
extern "C" int tg_dis_score_prepare(const TgDisWeights* w, int N, int L, const float* dropout_uniforms,
                                    const float* feature_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0 && L >= 1) return TG_OK;
    if (N < 0 || !workspace) {
        tg_set_error("tg_dis_score_prepare: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    FeatOpts fo;
    fo.mode = FeatOpts::kPrepareOnly;
    // no sequence is read and nothing is written through the output pointers in this mode: the workspace stands in for
    // the sequences' pointer check, the caller's pointers only decide (by their alignment) which highway is prepared
    return score_impl(w, reinterpret_cast<const int64_t*>(workspace), N, L, nullptr, 0, 0, 0.f, dropout_uniforms, nullptr,
                      nullptr, const_cast<float*>(feature_out), workspace, workspace_bytes, (hipStream_t)stream, fo);
}

extern "C" int tg_dis_score_prepared(const TgDisWeights* w, const int64_t* seqs, int N, int L, int col, float dropout_p,
                                     const float* dropout_uniforms, float* reward_out, float* logits_out,
                                     float* feature_out, void* workspace, size_t workspace_bytes, void* stream) {
    FeatOpts fo;
    fo.reuse = true;
    return score_impl(w, seqs, N, L, nullptr, 0, col, dropout_p, dropout_uniforms, reward_out, logits_out,
                      feature_out, workspace, workspace_bytes, (hipStream_t)stream, fo);
}

// get_feature for the LeakGAN rollout's position loop (tg_leak.hip): tg_dis_score restricted to feature_out, reading
// the tokens at positions >= valid_len as padding (token 0).  The prepared weights live in weights_ws, shared by any
// number of per-call workspaces (each laid out for N_layout sequences): tg_dis_features_prepare once per (w, N_layout),
// on a stream every later call is ordered behind.
size_t tg_dis_features_weights_bytes(const TgDisWeights* w, int L) { return w && L >= 1 ? weights_bytes(w, L) : 0; }
size_t tg_dis_features_call_bytes(const TgDisWeights* w, int N_layout, int L) {
    return w && N_layout >= 1 && L >= 1 ? call_bytes(w, N_layout, L, 0) : 0;
}

int tg_dis_features_prepare(const TgDisWeights* w, int N_layout, int L, float* feature_out, void* const* call_ws, int n_ws,
                            size_t call_ws_bytes, void* weights_ws, hipStream_t stream) {
    if (!call_ws || n_ws < 1 || !weights_ws || N_layout < 1) {
        tg_set_error("tg_dis_features_prepare: bad arguments");
        return TG_ERR_ARG;
    }
    for (int q = 0; q < n_ws; ++q) {
        FeatOpts fo;
        fo.valid_len = L;
        fo.N_layout = N_layout;
        fo.weights_ws = weights_ws;
        fo.mode = q == 0 ? FeatOpts::kPrepareOnly : FeatOpts::kPadOnly;
        // no sequence is read in these modes: any non-null pointer passes the argument check
        const int rc = score_impl(w, reinterpret_cast<const int64_t*>(weights_ws), N_layout, L, nullptr, 0, 0, 0.f, nullptr,
                                  nullptr, nullptr, feature_out, call_ws[q], call_ws_bytes, stream, fo);
        if (rc) return rc;
    }
    return TG_OK;
}

int tg_dis_features(const TgDisWeights* w, const int64_t* seqs, int N, int N_layout, int L, int valid_len,
                    float* feature_out, void* call_ws, size_t call_ws_bytes, void* weights_ws, hipStream_t stream) {
    if (valid_len < 0 || valid_len > L || N_layout < N || !weights_ws) {
        tg_set_error("tg_dis_features: valid_len %d outside [0, %d], N_layout %d < N %d or no weights workspace", valid_len,
                     L, N_layout, N);
        return TG_ERR_ARG;
    }
    FeatOpts fo;
    fo.reuse = true;
    fo.valid_len = valid_len;
    fo.N_layout = N_layout;
    fo.weights_ws = weights_ws;
    return score_impl(w, seqs, N, L, nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, feature_out, call_ws, call_ws_bytes,
                      stream, fo);
}

extern "C" size_t tg_dis_score_rollout_workspace_bytes(const TgDisWeights* w, int B, int L, int R) {
    if (!w || B < 1 || L < 1 || R < 1) return 0;
    return ws_bytes(w, R * L * B, L, B);
}

extern "C" int tg_dis_score_rollout(const TgDisWeights* w, const int64_t* sentences, const int64_t* completions,
                                    int B, int L, int R, int col, float dropout_p, const float* dropout_uniforms,
                                    float* reward_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (B == 0 && R >= 0 && L >= 1) return TG_OK;   // no sentence: nothing to score
    if (!sentences || B < 1 || R < 0 || (long)R * L * B > 0x7fffffffL) {
        tg_set_error("tg_dis_score_rollout: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    return score_impl(w, completions, R * L * B, L, sentences, B, col, dropout_p, dropout_uniforms, reward_out,
                      nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

// tg_dis_score_rollout in two parts: what needs only the weights and the sentences (|max|, scales, weight pieces, the
// sentences' window values and their running maxima), and the rest.  The first part can run beside the generator.
extern "C" int tg_dis_score_rollout_prepare(const TgDisWeights* w, const int64_t* sentences, int B, int L, int R,
                                            const float* dropout_uniforms, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    if (B == 0 && R >= 0 && L >= 1) return TG_OK;
    if (!sentences || B < 1 || R < 1 || (long)R * L * B > 0x7fffffffL) {
        tg_set_error("tg_dis_score_rollout_prepare: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    FeatOpts fo;
    fo.mode = FeatOpts::kPrepareOnly;
    // no completion is read in this mode: the sentences stand in for the pointer check
    return score_impl(w, sentences, R * L * B, L, sentences, B, 0, 0.f, dropout_uniforms, nullptr, nullptr, nullptr,
                      workspace, workspace_bytes, (hipStream_t)stream, fo);
}

extern "C" int tg_dis_score_rollout_prepared(const TgDisWeights* w, const int64_t* sentences, const int64_t* completions,
                                             int B, int L, int R, int col, float dropout_p,
                                             const float* dropout_uniforms, float* reward_out, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    if (B == 0 && R >= 0 && L >= 1) return TG_OK;
    if (!sentences || B < 1 || R < 0 || (long)R * L * B > 0x7fffffffL) {
        tg_set_error("tg_dis_score_rollout_prepared: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    FeatOpts fo;
    fo.reuse = true;
    return score_impl(w, completions, R * L * B, L, sentences, B, col, dropout_p, dropout_uniforms, reward_out,
                      nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, fo);
}
