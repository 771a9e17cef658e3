// K-GEN: the LSTM generator as one persistent kernel per 32-row tile (gfx950).
//
// One workgroup (4 waves) owns 32 rows (completions / sequences) for ALL of their
// timesteps: W_ih/W_hh live in VGPRs as MFMA A-operands, h lives in LDS (and in
// VGPRs as the MFMA B-operand), c lives in VGPRs.  Per step:
//   cell      gates[128 x 32rows] = [W_ih|W_hh] . [x;h]   32 x v_mfma_f32_32x32x2_f32 per wave
//             (rows permuted so that each lane ends up with i,f,g,o of its own 4 units)
//   project   logits[V x 32rows] = W_out . h + b          W_out streamed from L2, 16 MFMA per
//             32x32 tile; the vocab is cut into 8 contiguous spans, one per (wave, lane-half),
//             so every lane sees its span in natural vocab order, 16 entries per tile
//   softmax   lane-local online max / sums per tile; per-group records (max, sum) in LDS
//   sample    canonical inverse-CDF draw (fp64 above tile level) from a pre-drawn uniform,
//             the chosen group's entries recomputed cooperatively; the token goes straight
//             into the completions buffer and back into the next step's embedding gather.
// Logits never touch HBM.  Replaces models/generator.py:40-84, utils/rollout.py:25-55 and
// the forward inside metrics/nll.py:59-61 of the reference.
//
// Two operand placements, one kernel body (template <SAMPLE, HT = H/32, REG>):
//   REG  (E = H = 32, every SeqGAN / SentiGAN / MaliGAN / oracle config of the reference):
//        gate weights and the h / x B-operands stay in VGPRs for the whole kernel;
//   !REG (H in {32,64,128,256,512}, E any multiple of 32 -- the scaled config): h and the
//        step's embedded inputs live in LDS ([32][H+4], [32][E+4]), gate weights stream from L2;
//        each wave owns H/32 gate tiles (its c state: H/8 VGPRs).
#include <stdlib.h>

#include <mutex>

#include <type_traits>

#include "tg_common.h"
#include "tg_genw.h"
#include "tg_pgw.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_{f16,bf16}: A lane l -> A[i=l&31][k=8*(l>>5)+0..7]; B lane l -> B[k=8*(l>>5)+0..7][j=l&31];
// D as for 32x32x2.  32 cycles per SIMD.
#define TG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define TG_MFMA16BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int kRows = 32;     // rows per workgroup (MFMA N)
constexpr int kThreads = 256; // 4 waves
constexpr int kNSpan = 8;
constexpr float kGenHScale = 8192.f;  // split projection: |h * log2(e)| < 1.4427 -> below 2^14 in fp16

struct SamplerGeom {
    int Ls;  // span length (entries)
    int NT;  // tiles per span
    int GK;  // tiles per group
    int NG;  // groups per span
};

struct GenArgs {
    TgGenWeights w;
    SamplerGeom geo;
    // ---- sampling modes (rollout / sample)
    const int64_t* sentences;  // rollout: [B,L]
    const float* states;       // rollout: [L+1,B,2,H] prefix states
    const float* uniforms;
    const int64_t* forced;     // nullable (replay mode)
    int64_t* out;              // rollout: [R,L,B,L]; sample: [N,L]
    int64_t start_letter;
    int rollout;               // 1 = rollout tiles, 0 = free-running sample
    int B, L, R, g_lo, i_lo;
    int rows_g, tiles_g;       // rollout: rows and tiles per given_num
    int N;                     // sample / teacher-forced rows
    // ---- teacher-forced mode
    const int64_t* inp;        // [N,L]
    const int64_t* target;     // [N,L] or null
    float* logp_out;           // [N,L] or null
    float* lse_out;            // [N,L] or null
    float* logits_out;         // [N,L,V] or null
    float* states_out;         // [L+1,N,2,H] or null
    const float* states_in;    // [N,2,H] or null: (h,c) the teacher-forced pass starts from (null = zeros)
    int need_proj;
    // ---- split-fp16 projection (TG_GEN_SPLIT_F16): W_out as MFMA A fragments in load order, see k_gen_make_frag
    const f32x4* frag;         // [4 waves][NT tiles][5][64 lanes] 16-B pieces
    const float* gscale;       // [0] = W_out scale, [1] = h scale, [2] = 1/(both)
    long long* stamps;  // diagnostic (TG_STAMPS env = device pointer): per-workgroup s_memtime sums [grid][4]
};


// One row of W_out (H floats) dotted with h in the MFMA's own k order (per 32-wide chunk the
// k pairs (j, 16+j), j = 0..15, then the bias), so that a recomputed logit reproduces
// the MFMA result.
template <int HT>
__device__ __forceinline__ float dot_mfma_order(const float* __restrict__ wrow, const float* hrow, float bias,
                                                float hscale = 1.f, bool bias_prescaled = false) {
    float acc = 0.f;
#pragma unroll 2
    for (int c = 0; c < HT; ++c) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(wrow + 32 * c);
        f32x4 wl[4], wh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { wl[i] = w4[i]; wh[i] = w4[4 + i]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = __builtin_fmaf(wl[j >> 2][j & 3], hrow[32 * c + j] * hscale, acc);
            acc = __builtin_fmaf(wh[j >> 2][j & 3], hrow[32 * c + 16 + j] * hscale, acc);
        }
    }
    return acc + (bias_prescaled ? bias : bias * hscale);  // the bias enters last, as one more k step (a = b_out[v] * hscale, b = 1)
}

// 16 consecutive floats (64 B, 16-B aligned) -> registers
__device__ __forceinline__ void load16(const float* p, float (&dst)[16]) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = p4[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * q + e] = v[e];
    }
}

__device__ __forceinline__ int clamp_tok(int64_t t, int V) { return tg_min(tg_max((int)t, 0), V - 1); }

// WPE = waves per SIMD the kernel is built for: 2 (256 VGPRs) or, for the split kernel on grids that fill the chip
// three times over, 3 (168 VGPRs: gate weights re-read per step, two A fragments in flight instead of three).
template <bool SAMPLE, int HT, bool REG, bool SPLIT, int WPE = 2>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_gen(const GenArgs a) {
    static_assert(WPE == 2 || (WPE == 3 && SPLIT), "three waves per SIMD: split kernel only");
    constexpr bool kLean = WPE == 3;
    static_assert(!SPLIT || REG, "the split-fp16 projection exists for the E = H = 32 kernels");
    constexpr int kH = 32 * HT;
    constexpr int kHStride = kH + 4;  // hbuf row stride (floats): 16-B aligned, conflict-free b128 reads
    static_assert(!REG || HT == 1, "register-resident operands need H = 32");
    const int E = REG ? 32 : a.w.E;
    const int ET = E >> 5;
    const int kEStride = E + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = reinterpret_cast<float*>(smem);                     // [32][H+4]
    float* mbuf = hbuf + kRows * kHStride;                             // [8][32]
    double* sbuf = reinterpret_cast<double*>(mbuf + kNSpan * kRows);   // [8][32]
    int* tokbuf = reinterpret_cast<int*>(sbuf + kNSpan * kRows);       // [32] next input token per row
    int* selv = tokbuf + kRows;                                        // [32] chosen group's first vocab id
    float* Mbuf = reinterpret_cast<float*>(selv + kRows);              // [32] row max
    double* selp = reinterpret_cast<double*>(Mbuf + kRows);            // [32] prefix before the group
    double* tgt = selp + kRows;                                        // [32] target = u*S
    float* gbl = reinterpret_cast<float*>(tgt + kRows);                // [128] b_ih + b_hh (lean kernel only)
    float* wbuf = gbl + (kLean ? 4 * kH : 0);                           // [32][EG+1] recomputed weights
    float* xbuf = wbuf;                                                // !REG: [32][E+4] step inputs (aliases wbuf/rec)
    const int EG = a.geo.GK * 16;
    float2* rec = reinterpret_cast<float2*>(wbuf + kRows * (EG + 1));  // [NG][8][32] (32*(EG+1) is even)

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n = lane & 31;
    const int hh = lane >> 5;
    const int sp = 2 * w + hh;
    const int V = a.w.V, L = a.L;
    const int Ls = a.geo.Ls, NT = a.geo.NT, GK = a.geo.GK;

    // ------------------------------------------------------------------ row setup
    bool valid;
    int t0 = 0;
    int tok;                       // pending input token of row n
    const float* state0 = nullptr; // initial (h,c) of row n or null = zeros
    int64_t* out_row = nullptr;
    const int64_t* forced_row = nullptr;
    const float* u_base = nullptr;
    size_t u_stride = 0;
    size_t tf_row = 0;             // teacher-forced: row index
    if (SAMPLE) {
        if (a.rollout) {
            const int g = a.g_lo + blockIdx.x / a.tiles_g;
            const int fr = (blockIdx.x % a.tiles_g) * kRows + n;
            valid = fr < a.rows_g;
            const int frc = valid ? fr : a.rows_g - 1;
            const int i = a.i_lo + frc / a.B, b = frc % a.B;
            const size_t row = ((size_t)i * L + (g - 1)) * a.B + b;
            out_row = a.out + row * L;
            forced_row = a.forced ? a.forced + row * L : nullptr;
            u_base = a.uniforms + ((size_t)i * L + (g - 1)) * L * a.B + b;
            u_stride = a.B;
            t0 = g;
            // prefix copy: the 8 (wave, half) threads of a row interleave over positions
            if (valid)
                for (int s = sp; s < g; s += kNSpan) out_row[s] = a.sentences[(size_t)b * L + s];
            if (g >= L) return;  // given_num == L: the sentence itself (uniform per workgroup)
            tok = clamp_tok(a.sentences[(size_t)b * L + g - 1], V);
            state0 = a.states + ((size_t)(g - 1) * a.B + b) * 2 * kH;
        } else {
            const int fr = blockIdx.x * kRows + n;
            valid = fr < a.N;
            const int frc = valid ? fr : a.N - 1;
            out_row = a.out + (size_t)frc * L;
            forced_row = a.forced ? a.forced + (size_t)frc * L : nullptr;
            u_base = a.uniforms + frc;
            u_stride = a.N;
            tok = clamp_tok(a.start_letter, V);
        }
    } else {
        const int fr = blockIdx.x * kRows + n;
        valid = fr < a.N;
        tf_row = valid ? fr : a.N - 1;
        tok = clamp_tok(a.inp[tf_row * L], V);  // ids must be < V; clamped for memory safety
        if (a.states_in) state0 = a.states_in + tf_row * 2 * kH;   // a carried (h,c): generator.py:40-51
        if (a.states_out && valid && w == 0) {  // slot 0 = the initial state (zeros without states_in)
            float* s0 = a.states_out + tf_row * 2 * kH;
            for (int k = hh * kH; k < hh * kH + kH; ++k) s0[k] = state0 ? state0[k] : 0.f;
        }
    }
    if (sp == 0) tokbuf[n] = tok;

    // ------------------------------------------------------------------ resident weights (REG)
    // gate tile T covers hidden units [8T, 8T+8): tile row i -> gate i>>3, unit 8T + (i&7), so that
    // D reg r of lane (n, hh) is gate r>>2 of unit 8T + 4hh + (r&3): i,f,g,o of a unit share a lane.
    // The lean kernel (three waves per SIMD, 168 VGPRs) re-reads its gate weights (L2-resident) at every step and
    // keeps the summed gate bias in LDS instead of holding 48 registers through the projection.
    float wih[16], whh[16];
    f32x16 gbias;
    if constexpr (kLean) {
        if (tid < 4 * kH) gbl[tid] = a.w.b_ih[tid] + a.w.b_hh[tid];
    } else if constexpr (REG) {
        const int i = lane & 31;
        const int grow = (i >> 3) * kH + 8 * w + (i & 7);
        load16(a.w.w_ih + (size_t)grow * 32 + 16 * hh, wih);
        load16(a.w.w_hh + (size_t)grow * kH + 16 * hh, whh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = (r >> 2) * kH + 8 * w + 4 * hh + (r & 3);
            gbias[r] = a.w.b_ih[gr] + a.w.b_hh[gr];
        }
    }

    // ------------------------------------------------------------------ initial state
    // wave w owns gate tiles w*HT .. w*HT+HT-1; this lane's units of tile j: 8*(w*HT+j) + 4hh + 0..3
    f32x4 c4[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
        const int ub = 8 * (w * HT + j) + 4 * hh;
        f32x4 h4 = {0.f, 0.f, 0.f, 0.f};
        c4[j] = h4;
        if (state0) {
            h4 = *reinterpret_cast<const f32x4*>(state0 + ub);
            c4[j] = *reinterpret_cast<const f32x4*>(state0 + kH + ub);
        }
        *reinterpret_cast<f32x4*>(hbuf + n * kHStride + ub) = h4;
    }
    __syncthreads();
    float hb[16];  // REG: MFMA B operand h[n][16hh + j], kept in registers
    if constexpr (REG) load16(hbuf + n * kHStride + 16 * hh, hb);
    __syncthreads();  // everyone has read hbuf before the first cell overwrites it

    // A-operand row of this lane in the projection tiles
    const int am = lane & 31;
    const int a_sp = 2 * w + ((am >> 2) & 1);
    const int a_r = (am & 3) + 4 * (am >> 3);
    const int last_span = (V - 1) / Ls;

    long long st_cell = 0, st_proj = 0, st_draw = 0, st_n = 0, st_d[4] = {0, 0, 0, 0};
    for (int t = t0; t < L; ++t) {
        const long long c0 = tg_clock();
        // ============================================================== LSTM cell
        if constexpr (REG) {
            float xb[16];
            load16(a.w.emb + (size_t)tok * 32 + 16 * hh, xb);
            if constexpr (kLean) {
                const float* pih = a.w.w_ih;
                const float* phh = a.w.w_hh;
                asm volatile("" : "+s"(pih), "+s"(phh));  // keeps the loads inside the step loop
                const int i = lane & 31;
                const int grow = (i >> 3) * kH + 8 * w + (i & 7);
                load16(pih + (size_t)grow * 32 + 16 * hh, wih);
                load16(phh + (size_t)grow * kH + 16 * hh, whh);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(gbl + q4 * kH + 8 * w + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gbias[4 * q4 + e] = b4[e];
                }
            }
            f32x16 acc = gbias;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = TG_MFMA32(wih[j], xb[j], acc);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = TG_MFMA32(whh[j], hb[j], acc);
            f32x4 h4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = tg_sigmoid(acc[e]);
                const float fg = tg_sigmoid(acc[4 + e]);
                const float gg = tg_tanh(acc[8 + e]);
                const float og = tg_sigmoid(acc[12 + e]);
                const float cn = fg * c4[0][e] + ig * gg;
                c4[0][e] = cn;
                h4[e] = og * tg_tanh(cn);
            }
            const int ub = 8 * w + 4 * hh;
            *reinterpret_cast<f32x4*>(hbuf + n * kHStride + ub) = h4;  // old h is only in registers by now
            if (!SAMPLE && a.states_out && valid) {
                float* st = a.states_out + ((size_t)(t + 1) * a.N + tf_row) * 2 * kH;
                *reinterpret_cast<f32x4*>(st + ub) = h4;
                *reinterpret_cast<f32x4*>(st + kH + ub) = c4[0];
            }
            __syncthreads();
            load16(hbuf + n * kHStride + 16 * hh, hb);
        } else {
            // stage the step's embedded inputs: xbuf[r][:] = emb[tokbuf[r]]
            const int e4 = E >> 2;
            for (int idx = tid; idx < kRows * e4; idx += kThreads) {
                const int r = idx / e4, c = idx - r * e4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(a.w.emb + (size_t)tokbuf[r] * E + 4 * c);
                *reinterpret_cast<f32x4*>(xbuf + r * kEStride + 4 * c) = v;
            }
            __syncthreads();
            f32x4 hnew[HT];
#pragma unroll
            for (int j = 0; j < HT; ++j) {
                const int T = w * HT + j;
                const int i = lane & 31;
                const int grow = (i >> 3) * kH + 8 * T + (i & 7);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gr = (r >> 2) * kH + 8 * T + 4 * hh + (r & 3);
                    acc[r] = a.w.b_ih[gr] + a.w.b_hh[gr];
                }
                float av[16], bv[16];
                for (int c = 0; c < ET; ++c) {
                    load16(a.w.w_ih + (size_t)grow * E + 32 * c + 16 * hh, av);
                    load16(xbuf + n * kEStride + 32 * c + 16 * hh, bv);
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = TG_MFMA32(av[k], bv[k], acc);
                }
                for (int c = 0; c < HT; ++c) {
                    load16(a.w.w_hh + (size_t)grow * kH + 32 * c + 16 * hh, av);
                    load16(hbuf + n * kHStride + 32 * c + 16 * hh, bv);
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = TG_MFMA32(av[k], bv[k], acc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ig = tg_sigmoid(acc[e]);
                    const float fg = tg_sigmoid(acc[4 + e]);
                    const float gg = tg_tanh(acc[8 + e]);
                    const float og = tg_sigmoid(acc[12 + e]);
                    const float cn = fg * c4[j][e] + ig * gg;
                    c4[j][e] = cn;
                    hnew[j][e] = og * tg_tanh(cn);
                }
            }
            __syncthreads();  // every wave is done reading the old h (and xbuf)
#pragma unroll
            for (int j = 0; j < HT; ++j) {
                const int ub = 8 * (w * HT + j) + 4 * hh;
                *reinterpret_cast<f32x4*>(hbuf + n * kHStride + ub) = hnew[j];
                if (!SAMPLE && a.states_out && valid) {
                    float* st = a.states_out + ((size_t)(t + 1) * a.N + tf_row) * 2 * kH;
                    *reinterpret_cast<f32x4*>(st + ub) = hnew[j];
                    *reinterpret_cast<f32x4*>(st + kH + ub) = c4[j];
                }
            }
            __syncthreads();
        }
        if (!SAMPLE && !a.need_proj) {
            if (t + 1 < L) {
                tok = clamp_tok(a.inp[tf_row * L + t + 1], V);
                if (sp == 0) tokbuf[n] = tok;
            }
            __syncthreads();
            continue;
        }

        const long long c1 = tg_clock();
        // ============================================================== projection + online softmax
        // The pipelined path works in base 2: h (B operand) and the bias are pre-multiplied by log2(e), so the
        // MFMA delivers l' = l*log2(e) and a weight is one v_exp_f32 (1 ulp) of (l' - m'): 2 VALU per entry
        // instead of 7.  Costs one extra fp32 rounding on h (same size as the GEMM's own rounding noise).
        const bool b2 = REG && (SAMPLE || !a.logits_out);
        const float kL2E = 1.44269502162933349609375f;
        float m_run = TG_NEG_BIG;  // finite sentinel: exp(TG_NEG_BIG - m) is exactly 0, never NaN
        float c_grp = 0.f;         // SAMPLE: current group's sum at scale m_run; TF: running sum of the span
        const int sp_lo = sp * Ls;
        {
            // one 32x32 tile: 16 consecutive vocab entries of this lane's span, all 32 rows
            auto tile = [&](int q, const float (&areg)[16], auto full_tag) {
                constexpr bool full = decltype(full_tag)::value;  // every lane's 16 entries are < V
                const int vb = sp_lo + 16 * q;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                if constexpr (REG) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc = TG_MFMA32(areg[j], hb[j], acc);
                } else {
                    const float* wrow = a.w.w_out + (size_t)tg_min(a_sp * Ls + 16 * q + a_r, V - 1) * kH + 16 * hh;
                    const float* hrow = hbuf + n * kHStride + 16 * hh;
                    float av[16], bv[16];
                    for (int c = 0; c < HT; ++c) {
                        load16(wrow + 32 * c, av);
                        load16(hrow + 32 * c, bv);
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc = TG_MFMA32(av[j], bv[j], acc);
                    }
                }
                if (full) {
                    const f32x4* pb = reinterpret_cast<const f32x4*>(a.w.b_out + vb);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x4 bv = pb[qq];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * qq + e] += bv[e];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += a.w.b_out[tg_min(vb + r, V - 1)];
                }
                if (!SAMPLE && a.logits_out && valid) {
                    float* lo = a.logits_out + (tf_row * L + t) * (size_t)V + vb;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (vb + r < V) lo[r] = acc[r];
                }
                if (!full) {  // vocabulary tail (last span only): masked entries weigh exactly 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = (vb + r < V) ? acc[r] : TG_NEG_BIG;
                }
                float tmax = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = __builtin_fmaxf(tmax, acc[r]);
                const float m_new = __builtin_fmaxf(m_run, tmax);
                const bool grp_start = SAMPLE && (q % GK == 0);
                // rescale what was accumulated at the old max
                c_grp = grp_start ? 0.f : c_grp * tg_expf_neg(m_run - m_new);
                float ts = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ts += tg_expf_neg(acc[r] - m_new);
                c_grp += ts;
                m_run = m_new;
                if (SAMPLE && ((q % GK == GK - 1) || q == NT - 1))
                    rec[((q / GK) * kNSpan + sp) * kRows + n] = make_float2(m_run, c_grp);
            };
            // wave-uniform choice: the odd span of this wave reaches the vocabulary's end first
            auto tile_any = [&](int q, const float (&areg)[16]) {
                if ((2 * w + 1) * Ls + 16 * q + 16 <= V) tile(q, areg, std::true_type{});
                else tile(q, areg, std::false_type{});
            };
            int q_done = 0;
            if constexpr (REG) {
                if (SAMPLE || !a.logits_out) {
                    // Software-pipelined loop over all tiles: branch-free body, the MFMAs of tile q+1 are issued
                    // ahead of tile q's softmax VALU work so that one wave keeps both pipes busy.  Two register
                    // sets for A and for the accumulator.  Entries past the vocabulary's end are masked through
                    // the bias operand (TG_NEG_BIG -> weight exactly 0), so the tail needs no separate path.
                    const int nfull = NT;
                    const int gk_mask = GK - 1, gk_shift = __builtin_ctz(GK);
                    // A fragment of a tile.
                    // fp32:  16 floats of W_out[v] plus, as a 17th k step, the bias of v on the k = 0 lanes (against
                    //        b = 1): one dword per lane instead of a 64-B broadcast bias read.
                    // split: four 16-B pieces (hi, lo halves of this lane's 16 scaled weights, two k blocks each) and a
                    //        fifth holding the scaled bias as three bf16 parts (exact fp32), from the fragment-major
                    //        copy made by k_gen_make_frag: every load instruction of a wave reads 1 KiB contiguous.
                    struct AFrag { f32x4 w[4]; f32x4 bias; };
                    const float bone = hh ? 0.f : 1.f;
                    bf16x8 bone16;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bone16[e] = (__bf16)((hh == 0 && e < 3) ? 1.f : 0.f);
                    float inv = 1.f;
                    if constexpr (SPLIT) inv = a.gscale[2];
                    float hb2[16];
                    f16x8 hbh[4];  // split: (hi k0-7, hi k8-15, lo k0-7, lo k8-15) of h * log2(e) * 2^13
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        hb2[j] = hb[j] * kL2E;
                        if constexpr (SPLIT) {
                            const float x = hb2[j] * kGenHScale;
                            const _Float16 xh = (_Float16)x;
                            hbh[j >> 3][j & 7] = xh;
                            hbh[2 + (j >> 3)][j & 7] = (_Float16)(x - (float)xh);
                        }
                    }
                    auto load_a = [&](int q, AFrag& dst) {
                        if constexpr (SPLIT) {
                            const f32x4* p = a.frag + ((size_t)(w * NT + q) * 5) * 64 + lane;
#pragma unroll
                            for (int i = 0; i < 4; ++i) dst.w[i] = p[i * 64];
                            dst.bias = p[4 * 64];
                        } else {
                            const int v = a_sp * Ls + 16 * q + a_r;
                            const f32x4* p = reinterpret_cast<const f32x4*>(a.w.w_out + (size_t)tg_min(v, V - 1) * kH + 16 * hh);
#pragma unroll
                            for (int i = 0; i < 4; ++i) dst.w[i] = p[i];
                            // every lane loads (the k = 1 lanes meet b = 0 in the MFMA, any finite value does): a
                            // select on hh turns into a divergent branch with a full vmcnt(0) wait in the pipelined block
                            const float bv = a.w.b_out[tg_min(v, V - 1)] * kL2E;
                            dst.bias[0] = v < V ? bv : TG_NEG_BIG;
                        }
                    };
                    auto mfma_tile = [&](int q, const AFrag& af) {
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        if constexpr (SPLIT) {
                            // small terms first: lo.hi, hi.lo, hi.hi, then the bias (fp32 accumulation throughout)
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[2]), hbh[0], acc);
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[3]), hbh[1], acc);
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[0]), hbh[2], acc);
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[1]), hbh[3], acc);
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[0]), hbh[0], acc);
                            acc = TG_MFMA16(__builtin_bit_cast(f16x8, af.w[1]), hbh[1], acc);
                            acc = TG_MFMA16BF(__builtin_bit_cast(bf16x8, af.bias), bone16, acc);
                        } else {
#pragma unroll
                            for (int j = 0; j < 16; ++j) acc = TG_MFMA32(af.w[j >> 2][j & 3], hb2[j], acc);
                            acc = TG_MFMA32(af.bias[0], bone, acc);
                        }
                        return acc;
                    };
                    auto softmax_tile = [&](const f32x16& acc, int q) {
                        float tmax = acc[0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) tmax = __builtin_fmaxf(tmax, acc[r]);
                        if constexpr (SPLIT) tmax *= inv;  // accumulators are in scaled units; inv is a power of two
                        const float m_new = __builtin_fmaxf(m_run, tmax);
                        // branch-free group restart (a select on the scalar condition would split the block
                        // and serialise the MFMA chain ahead of this VALU work)
                        const float keep = (SAMPLE && (q & gk_mask) == 0) ? 0.f : 1.f;
                        c_grp = c_grp * __builtin_amdgcn_exp2f(m_run - m_new) * keep;
                        float ev[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            ev[r] = SPLIT ? __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], inv, -m_new))  // one rounding
                                          : __builtin_amdgcn_exp2f(acc[r] - m_new);
                        float ts = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) ts += ev[r];  // canonical order: sequential
                        c_grp += ts;
                        m_run = m_new;
                        // stored after every tile: the group's last tile leaves the final value
                        if (SAMPLE) rec[((q >> gk_shift) * kNSpan + sp) * kRows + n] = make_float2(m_run, c_grp);
                    };
                    // ask the scheduler for MFMA and VALU alternating in program order (the in-order wave otherwise
                    // parks on the dependent MFMA chain with the softmax work queued behind it)
                    auto interleave = [&]() {
                        if constexpr (SPLIT) {
#pragma unroll
                            for (int i = 0; i < 7; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0); // VALU
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 17; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // VALU
                            }
                        }
                    };
                    if constexpr (kLean) {
                        // two fragments in flight (coalesced 1-KiB loads, three waves per SIMD to cover them)
                        AFrag s0, s1;
                        f32x16 acc0, acc1;
                        load_a(0, s0);
                        if (nfull > 1) load_a(1, s1);
                        acc0 = mfma_tile(0, s0);
                        if (nfull > 2) load_a(2, s0);
                        int q = 0;
                        // invariant at loop head: acc0 = tile q (q even), s1 = tile q+1, s0 = tile q+2 in flight
                        for (; q + 2 < nfull; q += 2) {
                            acc1 = mfma_tile(q + 1, s1); softmax_tile(acc0, q); interleave();
                            load_a(tg_min(q + 3, nfull - 1), s1);
                            acc0 = mfma_tile(q + 2, s0); softmax_tile(acc1, q + 1); interleave();
                            load_a(tg_min(q + 4, nfull - 1), s0);
                        }
                        softmax_tile(acc0, q);
                        if (q + 1 < nfull) { acc1 = mfma_tile(q + 1, s1); softmax_tile(acc1, q + 1); }
                    } else if (nfull > 0) {
                        // A fragments are fetched three tiles ahead (an L2 round trip under load is ~2 tiles long)
                        AFrag a0, a1, a2;
                        f32x16 acc0, acc1;
                        load_a(0, a0);
                        if (nfull > 1) load_a(1, a1);
                        if (nfull > 2) load_a(2, a2);
                        acc0 = mfma_tile(0, a0);
                        int q = 0;
                        // invariant at loop head: acc0 = tile q (q % 6 == 0), a1 = tile q+1, a2 = tile q+2 in flight.
                        // Loads past the last tile are clamped, not skipped: a branch would cut the body into several
                        // basic blocks, the interleave below works within one, and without a store to pin it (teacher-
                        // forced mode) the softmax work sinks into the last block with six accumulators live.
                        for (; q + 6 < nfull; q += 6) {
                            load_a(q + 3, a0); acc1 = mfma_tile(q + 1, a1); softmax_tile(acc0, q);     interleave();
                            load_a(q + 4, a1); acc0 = mfma_tile(q + 2, a2); softmax_tile(acc1, q + 1); interleave();
                            load_a(q + 5, a2); acc1 = mfma_tile(q + 3, a0); softmax_tile(acc0, q + 2); interleave();
                            load_a(q + 6, a0); acc0 = mfma_tile(q + 4, a1); softmax_tile(acc1, q + 3); interleave();
                            load_a(tg_min(q + 7, nfull - 1), a1);
                            acc1 = mfma_tile(q + 5, a2); softmax_tile(acc0, q + 4); interleave();
                            load_a(tg_min(q + 8, nfull - 1), a2);
                            acc0 = mfma_tile(q + 6, a0); softmax_tile(acc1, q + 5); interleave();
                        }
                        // tail (at most 6 tiles): acc0 = tile q; a1 = tile q+1, a2 = tile q+2 if they exist
                        softmax_tile(acc0, q);
                        if (q + 1 < nfull) { acc1 = mfma_tile(q + 1, a1); softmax_tile(acc1, q + 1); }
                        if (q + 2 < nfull) { acc0 = mfma_tile(q + 2, a2); softmax_tile(acc0, q + 2); }
                        for (int qq = q + 3; qq < nfull; ++qq) {
                            load_a(qq, a0);
                            acc0 = mfma_tile(qq, a0);
                            softmax_tile(acc0, qq);
                        }
                    }
                    q_done = nfull;
                }
                float dummy[16];
                for (int q = q_done; q < NT; ++q) {
                    load16(a.w.w_out + (size_t)tg_min(a_sp * Ls + 16 * q + a_r, V - 1) * kH + 16 * hh, dummy);
                    tile_any(q, dummy);
                }
            } else {
                float dummy[16] = {};
                for (int q = 0; q < NT; ++q) tile_any(q, dummy);
            }
        }
        const long long c2 = tg_clock();
        mbuf[sp * kRows + n] = m_run;
        __syncthreads();
        const long long d0 = tg_clock();
        float M = mbuf[n];
#pragma unroll
        for (int s = 1; s < kNSpan; ++s) M = __builtin_fmaxf(M, mbuf[s * kRows + n]);

        if (!SAMPLE) {
            // ---------------------------------------------------------- teacher forced: lse + picked log-prob
            sbuf[sp * kRows + n] = (double)c_grp * (b2 ? exp2((double)(m_run - M)) : exp((double)(m_run - M)));  // empty span: c_grp = 0
            __syncthreads();
            if (tid < kRows) {
                double S = 0.0;
#pragma unroll
                for (int s = 0; s < kNSpan; ++s) S += sbuf[s * kRows + n];
                const float lse = b2 ? (float)(((double)M + log2(S)) * 0.69314718055994530942) : (float)((double)M + log(S));
                if (valid) {
                    const size_t o = tf_row * L + t;
                    if (a.lse_out) a.lse_out[o] = lse;
                    if (a.target && a.logp_out) {
                        const int tv = clamp_tok(a.target[o], V);
                        const float lg =
                            dot_mfma_order<HT>(a.w.w_out + (size_t)tv * kH, hbuf + n * kHStride, a.w.b_out[tv]);
                        a.logp_out[o] = lg - lse;
                    }
                }
            }
            if (t + 1 < L) {
                tok = clamp_tok(a.inp[tf_row * L + t + 1], V);
                if (sp == 0) tokbuf[n] = tok;
            }
            __syncthreads();
            continue;
        }

        // ============================================================== canonical inverse-CDF draw
        const int sp_hi = tg_min(V, sp_lo + Ls);
        const int ng_valid = (sp_hi > sp_lo) ? (sp_hi - sp_lo + EG - 1) / EG : 0;
        {
            // the running max of a span settles early: consecutive records mostly share their scale factor
            double s = 0.0, fac = 1.0;
            float fx = M;
            // batches of 16 groups (one batch up to V = 16384; GK stops at 8, so larger vocabularies have more groups)
#pragma unroll 1
            for (int qb = 0; qb < ng_valid; qb += 16) {
                float2 rq[16];  // all reads of a batch in flight before the (sequential, fp64) sum starts
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    rq[q] = rec[(tg_min(qb + q, tg_max(ng_valid - 1, 0)) * kNSpan + sp) * kRows + n];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (qb + q < ng_valid) {
                        const float2 r = rq[q];
                        if (r.x != fx) { fx = r.x; fac = b2 ? exp2((double)fx - (double)M) : exp((double)fx - (double)M); }
                        const double c = (double)r.y * fac;
                        s += c;
                        // the record is only read again by the group search below, which wants this product: leave it
                        // in place of the (max, sum) pair (same 8 bytes, same owner thread)
                        reinterpret_cast<double*>(rec)[((qb + q) * kNSpan + sp) * kRows + n] = c;
                    }
                }
            }
            sbuf[sp * kRows + n] = s;
        }
        __syncthreads();
        const long long d1 = tg_clock();
        {
            double S = 0.0;
#pragma unroll
            for (int s = 0; s < kNSpan; ++s) S += sbuf[s * kRows + n];
            const double target = (double)u_base[(size_t)t * u_stride] * S;
            int sel = last_span;
            double p = 0.0;
            for (int s = 0; s < kNSpan; ++s) {
                const double ss = sbuf[s * kRows + n];
                if (p + ss > target || s == last_span) { sel = s; break; }
                p += ss;
            }
            if (sel == sp) {  // exactly one of the row's 8 threads owns the draw
                int gq = ng_valid - 1;
                bool found = false;
#pragma unroll 1
                for (int qb = 0; qb < ng_valid && !found; qb += 16) {
                    double cq[16];  // the groups' contributions at the row's scale, left by the span-sum pass
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        cq[q] = reinterpret_cast<const double*>(rec)[(tg_min(qb + q, tg_max(ng_valid - 1, 0)) * kNSpan + sp) * kRows + n];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (qb + q < ng_valid && !found) {
                            if (p + cq[q] > target || qb + q == ng_valid - 1) { gq = qb + q; found = true; }
                            else p += cq[q];
                        }
                    }
                }
                selv[n] = sp_lo + gq * EG;
                selp[n] = p;
                tgt[n] = target;
                Mbuf[n] = M;
            }
        }
        __syncthreads();  // selections visible; every read of rec is done (wbuf / xbuf alias-free from here)
        const long long d2 = tg_clock();
        // cooperative recompute of the chosen groups: 8 threads per row, each EG/8 entries of that row with the
        // row's h held in registers (REG) or read from LDS per chunk
        if constexpr (SPLIT) {
            // The first pass ran on split-fp16 MFMAs, whose rounding a VALU chain cannot reproduce, so the recompute
            // is free to pick its own summation order: thread (row, c) holds floats [4c, 4c+4) of the row's h and
            // takes the same 16-B column slice of every W_out row of the group -- the 8 threads of a row read each
            // 128-B weight row as one contiguous line -- and the 8 partial dots are summed across lanes (DPP).
            const int rr = tid >> 3, c = tid & 7;
            const int gv = selv[rr];
            const int vend = tg_min(V, (gv / Ls + 1) * Ls);
            const float Mr = Mbuf[rr];
            f32x4 h4 = *reinterpret_cast<const f32x4*>(hbuf + rr * kHStride + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[e] *= kL2E;
            auto dpp_add = [](float x, auto ctrl) {
                constexpr int C = decltype(ctrl)::value;
                const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, 0xf, 0xf, true);
                return x + __builtin_bit_cast(float, y);
            };
            for (int i0 = 0; i0 < EG; i0 += 8) {
                f32x4 wv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    wv[i] = *reinterpret_cast<const f32x4*>(a.w.w_out + (size_t)tg_min(gv + i0 + i, V - 1) * kH + 4 * c);
                float mine = 0.f;  // entry i0 + c ends up with this thread
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float p = wv[i][0] * h4[0];
                    p = __builtin_fmaf(wv[i][1], h4[1], p);
                    p = __builtin_fmaf(wv[i][2], h4[2], p);
                    p = __builtin_fmaf(wv[i][3], h4[3], p);
                    p = dpp_add(p, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]: lane ^ 1
                    p = dpp_add(p, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]: lane ^ 2
                    p = dpp_add(p, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad of the 8
                    mine = (i == c) ? p : mine;
                }
                const int e = i0 + c;
                const int v = gv + e;
                float wgt = 0.f;
                if (v < vend) wgt = __builtin_amdgcn_exp2f((mine + a.w.b_out[v] * kL2E) - Mr);
                wbuf[rr * (EG + 1) + e] = wgt;
            }
        } else {
            const int rr = tid >> 3, e0 = tid & 7;
            const int gv = selv[rr];
            const int vend = tg_min(V, (gv / Ls + 1) * Ls);
            const float Mr = Mbuf[rr];
            const float hs = b2 ? kL2E : 1.f;
            if constexpr (REG) {
                float hrow[32];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(hbuf + rr * kHStride + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hrow[4 * q4 + e] = v4[e] * hs;
                }
                for (int e = e0; e < EG; e += 8) {
                    const int v = gv + e;
                    float wv = 0.f;
                    if (v < vend) {
                        const float lg = dot_mfma_order<1>(a.w.w_out + (size_t)v * kH, hrow, a.w.b_out[v] * hs, 1.f, true);
                        wv = b2 ? __builtin_amdgcn_exp2f(lg - Mr) : tg_expf(lg - Mr);
                    }
                    wbuf[rr * (EG + 1) + e] = wv;
                }
            } else {
                for (int e = e0; e < EG; e += 8) {
                    const int v = gv + e;
                    float wv = 0.f;
                    if (v < vend) {
                        const float lg = dot_mfma_order<HT>(a.w.w_out + (size_t)v * kH, hbuf + rr * kHStride, a.w.b_out[v]);
                        wv = tg_expf(lg - Mr);
                    }
                    wbuf[rr * (EG + 1) + e] = wv;
                }
            }
        }
        __syncthreads();
        const long long d3 = tg_clock();
        if (tid < kRows) {
            const int gv = selv[n];
            const int vend = tg_min(V, (gv / Ls + 1) * Ls);
            const int g_hi = tg_min(vend, gv + EG);
            const int nt_valid = (g_hi - gv + 15) / 16;
            const double target = tgt[n];
            double p = selp[n];
            const float* wr = wbuf + n * (EG + 1);
            int tq = nt_valid - 1;
            bool found = false;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {  // GK <= 8; same arithmetic as a loop with an early exit
                if (tt < nt_valid && !found) {
                    float ts = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ts += wr[tt * 16 + r];  // entries >= vend are 0
                    if (p + (double)ts > target || tt == nt_valid - 1) { tq = tt; found = true; }
                    else p += (double)ts;
                }
            }
            const int t_lo = gv + tq * 16;
            const int t_hi = tg_min(g_hi, t_lo + 16);
            int drawn = t_hi - 1;
            float wt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) wt[r] = wr[tq * 16 + r];
            float pp = 0.f;
            bool hit = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pp += wt[r];
                if (!hit && t_lo + r < t_hi && p + (double)pp > target) { drawn = t_lo + r; hit = true; }
            }
            if (valid) out_row[t] = drawn;
            tokbuf[n] = forced_row ? clamp_tok(forced_row[t], V) : drawn;
        }
        __syncthreads();
        tok = tokbuf[n];
        const long long c3 = tg_clock();
        st_cell += c1 - c0; st_proj += c2 - c1; st_draw += c3 - c2; st_n += 1;
        st_d[0] += d0 - c2; st_d[1] += d1 - d0; st_d[2] += d2 - d1; st_d[3] += d3 - d2;
    }
    if (SAMPLE && a.stamps && tid == 0) {
        long long* o = a.stamps + (size_t)blockIdx.x * 8;
        o[0] = st_cell; o[1] = st_proj; o[2] = st_draw; o[3] = st_n;
        o[4] = st_d[0]; o[5] = st_d[1]; o[6] = st_d[2]; o[7] = st_d[3];
    }
}

// ---- split-fp16 projection: per-call preparation (the weights are live module parameters)
__global__ __launch_bounds__(256) void k_gen_absmax(const float* x, size_t n, unsigned* out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);  // |x| as ordered bits
    __shared__ unsigned red[4];  // one atomic per workgroup: same-address atomics serialise
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m) atomicMax(out, m);
    }
}

__device__ __forceinline__ unsigned short tg_bf16_rn(float x) {  // round to nearest even; x finite
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// W_out / b_out -> the A fragments of the split projection, in the order k_gen's waves load them:
// piece ((w*NT + q)*5 + i)*64 + lane, 16 B each.  Lane (am, hh) of wave w owns vocab row
// v = (2w + ((am>>2)&1))*Ls + 16q + (am&3) + 4(am>>3) (so that every D lane ends up with 16 consecutive entries of
// its span) and the k range [16hh, 16hh+16):
//   i = 0,1  fp16 hi of W_out[v][k]*sW, k blocks [0,8) and [8,16) of the lane's range
//   i = 2,3  fp16 lo (the residual) of the same
//   i = 4    lanes hh = 0: b_out[v]*log2(e)*sW*sH as three bf16 parts (sums back to the fp32 value exactly), or
//            TG_NEG_BIG past the vocabulary's end; lanes hh = 1: zeros.
// sW = power of two that puts max|W_out| in [2^13, 2^14); sH = 2^13.  Block (0,0) also publishes the scales.
__global__ __launch_bounds__(64) void k_gen_make_frag(const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                      int V, int Ls, int NT, const unsigned* absmax, float* gscale,
                                                      f32x4* frag) {
    const int q = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const float m = __uint_as_float(*absmax);
    float sW = 1.f;
    if (m > 0.f && m < __builtin_inff()) {
        int ex = 0;
        (void)frexpf(m, &ex);       // m < 2^ex
        sW = ldexpf(1.f, 14 - ex);  // m*sW in [2^13, 2^14)
    }
    if (q == 0 && w == 0 && lane == 0) {
        gscale[0] = sW;
        gscale[1] = kGenHScale;
        gscale[2] = 1.f / (sW * kGenHScale);
    }
    const int am = lane & 31, hh = lane >> 5;
    const int v = (2 * w + ((am >> 2) & 1)) * Ls + 16 * q + (am & 3) + 4 * (am >> 3);
    const float* row = w_out + (size_t)tg_min(v, V - 1) * 32 + 16 * hh;
    f16x8 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float x = row[j] * sW;
        const _Float16 xh = (_Float16)x;
        hi[j >> 3][j & 7] = xh;
        lo[j >> 3][j & 7] = (_Float16)(x - (float)xh);
    }
    f32x4* dst = frag + ((size_t)(w * NT + q) * 5) * 64 + lane;
    dst[0] = __builtin_bit_cast(f32x4, hi[0]);
    dst[64] = __builtin_bit_cast(f32x4, hi[1]);
    dst[128] = __builtin_bit_cast(f32x4, lo[0]);
    dst[192] = __builtin_bit_cast(f32x4, lo[1]);
    unsigned p01 = 0, p2 = 0;
    if (hh == 0) {
        const float kL2E = 1.44269502162933349609375f;
        const float x = (v < V) ? (b_out[v] * kL2E) * (sW * kGenHScale) : TG_NEG_BIG;
        const unsigned short b0 = tg_bf16_rn(x);
        const float r1 = x - __uint_as_float((unsigned)b0 << 16);
        const unsigned short b1 = tg_bf16_rn(r1);
        const float r2 = r1 - __uint_as_float((unsigned)b1 << 16);
        const unsigned short b2 = tg_bf16_rn(r2);
        p01 = (unsigned)b0 | ((unsigned)b1 << 16);
        p2 = (unsigned)b2;
    }
    f32x4 bias;
    bias[0] = __uint_as_float(p01);
    bias[1] = __uint_as_float(p2);
    bias[2] = 0.f;
    bias[3] = 0.f;
    dst[256] = bias;
}

SamplerGeom make_geom(int V) {
    SamplerGeom g;
    g.Ls = 16 * ((V + kNSpan * 16 - 1) / (kNSpan * 16));
    g.NT = g.Ls / 16;
    g.GK = 1;
    while ((g.NT + g.GK - 1) / g.GK > 16 && g.GK < 8) g.GK *= 2;
    g.NG = (g.NT + g.GK - 1) / g.GK;
    return g;
}

bool is_reg(int E, int H) { return E == 32 && H == 32; }

// TG_GEN_SPLIT_F16 applies to the E = H = 32 shape (sampling kernels, tg_gen_nll); everything else computes in fp32
bool use_split(const TgGenWeights& w) { return w.precision == TG_GEN_SPLIT_F16 && is_reg(w.E, w.H); }

// bytes of the split-projection scratch: 256-B header (|max| bits, scales) + fragments
size_t split_bytes(const SamplerGeom& g) { return 256 + (size_t)4 * g.NT * 5 * 64 * sizeof(f32x4); }

// Fills the scratch (absmax -> scales -> fragments) on `stream` and points the kernel arguments at it.
int prepare_split(GenArgs& a, void* scratch, hipStream_t stream) {
    unsigned* absmax = reinterpret_cast<unsigned*>(scratch);
    float* gscale = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + 16);
    f32x4* frag = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(scratch) + 256);
    if (hipMemsetAsync(absmax, 0, 16, stream) != hipSuccess) {
        tg_set_error("hipMemsetAsync failed");
        return TG_ERR_LAUNCH;
    }
    const size_t n = (size_t)a.w.V * 32;
    const int blocks = (int)((n + 1023) / 1024);
    hipLaunchKernelGGL(k_gen_absmax, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, stream, a.w.w_out, n, absmax);
    hipLaunchKernelGGL(k_gen_make_frag, dim3(a.geo.NT, 4), dim3(64), 0, stream, a.w.w_out, a.w.b_out, a.w.V, a.geo.Ls,
                       a.geo.NT, absmax, gscale, frag);
    a.frag = frag;
    a.gscale = gscale;
    return TG_OK;
}

size_t smem_bytes(const SamplerGeom& g, bool sample, int E, int H, bool lean = false) {
    size_t b = (lean ? sizeof(float) * 4 * H : 0) + sizeof(float) * kRows * (H + 4) + sizeof(float) * kNSpan * kRows + sizeof(double) * kNSpan * kRows +
               sizeof(int) * kRows * 2 + sizeof(float) * kRows + sizeof(double) * kRows * 2;
    size_t region = 16;
    if (sample) {
        const int EG = g.GK * 16;
        region = sizeof(float) * (size_t)kRows * (EG + 1) + sizeof(float2) * (size_t)g.NG * kNSpan * kRows;
    }
    if (!is_reg(E, H)) {
        const size_t xb = sizeof(float) * (size_t)kRows * (E + 4);
        region = region > xb ? region : xb;
    }
    return b + region;
}

// wide_ok: the entry point has a GEMM-path implementation (tg_genw.hip), which takes any E, H multiple of 32
int check_weights(const TgGenWeights* w, bool wide_ok = false) {
    if (!w || !w->emb || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out) {
        tg_set_error("generator weights: null pointer");
        return TG_ERR_ARG;
    }
    // the kernels read the matrices 16 bytes at a time (include/textgan_hip.h, "Alignment")
    if ((reinterpret_cast<uintptr_t>(w->emb) | reinterpret_cast<uintptr_t>(w->w_ih) | reinterpret_cast<uintptr_t>(w->w_hh) |
         reinterpret_cast<uintptr_t>(w->w_out) | reinterpret_cast<uintptr_t>(w->b_ih) | reinterpret_cast<uintptr_t>(w->b_hh) |
         reinterpret_cast<uintptr_t>(w->b_out)) & 15) {
        tg_set_error("generator weights: every tensor must be 16-byte aligned");
        return TG_ERR_ARG;
    }
    const int sup = tg_gen_supported(w->V, w->E, w->H);
    if (sup == 0 || (sup == 1 && wide_ok && tg_genw_applies(w))) return TG_OK;
    if (sup == 1)
        tg_set_error("generator shape V=%d E=%d H=%d is served by the split-fp16 GEMM path only (tg_gen_rollout, "
                     "tg_gen_sample, tg_gen_nll with TG_GEN_SPLIT_F16, V <= 65536); the fp32 kernels need "
                     "H in {32,64,128,256,512} and E <= 512", w->V, w->E, w->H);
    else
        tg_set_error("generator shape V=%d E=%d H=%d not supported by this build (E, H multiples of 32 up to 1024)",
                     w->V, w->E, w->H);
    return TG_ERR_ARG;
}

template <bool SAMPLE, int HT, bool REG, bool SPLIT = false, int WPE = 2>
int launch_t(const GenArgs& a, int grid, hipStream_t stream) {
    size_t sm = smem_bytes(a.geo, SAMPLE, a.w.E, a.w.H, WPE == 3);
    if (sm > 160 * 1024) {
        tg_set_error("generator kernel needs %zu bytes of LDS (> 160 KiB) for V=%d E=%d H=%d", sm, a.w.V, a.w.E,
                     a.w.H);
        return TG_ERR_ARG;
    }
    static TgLdsAttr attr;
    if (int rc = tg_lds_attr(attr, reinterpret_cast<const void*>(&k_gen<SAMPLE, HT, REG, SPLIT, WPE>), sm, "k_gen")) return rc;
    tg_prof_begin(SAMPLE ? TG_KERNEL_GEN_SAMPLE : TG_KERNEL_GEN_TEACHER, stream);
    hipLaunchKernelGGL((k_gen<SAMPLE, HT, REG, SPLIT, WPE>), dim3(grid), dim3(kThreads), sm, stream, a);
    tg_prof_end(stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("k_gen launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

template <bool SAMPLE>
int launch(const GenArgs& a, int grid, hipStream_t stream) {
    if constexpr (SAMPLE) {
        // three workgroups per CU pay off once the grid fills them (4 waves x 3 on each of 256 CUs)
        if (a.frag) return grid >= 3 * 256 * 2 ? launch_t<true, 1, true, true, 3>(a, grid, stream)
                                               : launch_t<true, 1, true, true, 2>(a, grid, stream);
    }
    if constexpr (!SAMPLE) {
        if (a.frag && !a.logits_out) return launch_t<false, 1, true, true, 2>(a, grid, stream);  // tg_gen_nll
    }
    if (is_reg(a.w.E, a.w.H)) return launch_t<SAMPLE, 1, true>(a, grid, stream);
    switch (a.w.H) {
        case 32: return launch_t<SAMPLE, 1, false>(a, grid, stream);
        case 64: return launch_t<SAMPLE, 2, false>(a, grid, stream);
        case 128: return launch_t<SAMPLE, 4, false>(a, grid, stream);
        case 256: return launch_t<SAMPLE, 8, false>(a, grid, stream);
        case 512: return launch_t<SAMPLE, 16, false>(a, grid, stream);
    }
    tg_set_error("generator hidden size %d not compiled", a.w.H);
    return TG_ERR_ARG;
}

}  // namespace

extern "C" int tg_gen_supported(int V, int E, int H) {
    if (V < 1 || E < 32 || H < 32 || E > 1024 || H > 1024 || (E % 32) || (H % 32)) return -1;
    const bool h_pow2 = H == 32 || H == 64 || H == 128 || H == 256 || H == 512;
    return (h_pow2 && E <= 512) ? 0 : 1;
}
namespace {
int teacher_forced_impl(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                        const float* states_in, float* logp_out, float* lse_out, float* logits_out, float* states_out,
                        void* stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    if (N < 0 || L < 1 || (N > 0 && !inp)) {   // an empty batch (N = 0, any pointers) is a no-op
        tg_set_error("tg_gen_teacher_forced: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    GenArgs a = {};
    a.w = *w;
    a.geo = make_geom(w->V);
    a.N = N;
    a.L = L;
    a.inp = inp;
    a.target = target;
    a.logp_out = logp_out;
    a.lse_out = lse_out;
    a.logits_out = logits_out;
    a.states_out = states_out;
    a.states_in = states_in;
    a.need_proj = ((target && logp_out) || lse_out || logits_out) ? 1 : 0;
    return launch<false>(a, (N + kRows - 1) / kRows, (hipStream_t)stream);
}
}  // namespace

extern "C" int tg_gen_teacher_forced(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                                     float* logp_out, float* lse_out, float* logits_out, float* states_out,
                                     void* stream) {
    return teacher_forced_impl(w, inp, target, N, L, nullptr, logp_out, lse_out, logits_out, states_out, stream);
}

extern "C" size_t tg_gen_teacher_forced_workspace_bytes(const TgGenWeights* w, int N, int L, int have_logits_out,
                                                        int have_states_out) {
    if (!w || N < 1 || L < 1 || tg_gen_supported(w->V, w->E, w->H) != 1) return 0;
    return tg_pgw_forward_workspace_bytes(w, N, L, have_logits_out != 0, have_states_out != 0);
}

extern "C" int tg_gen_teacher_forced_ws(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                                        const float* states_in, float* logp_out, float* lse_out, float* logits_out,
                                        float* states_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !w->emb || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out) {
        tg_set_error("generator weights: null pointer");
        return TG_ERR_ARG;
    }
    const int sup = tg_gen_supported(w->V, w->E, w->H);
    if (states_in && ((uintptr_t)states_in & 15)) {
        tg_set_error("tg_gen_teacher_forced_ws: states_in must be 16-byte aligned");
        return TG_ERR_ARG;
    }
    if (sup == 0)
        return teacher_forced_impl(w, inp, target, N, L, states_in, logp_out, lse_out, logits_out, states_out, stream);
    if (sup < 0) return check_weights(w);
    if (N < 0 || L < 1 || (N > 0 && !inp) || (size_t)N * L >= ((size_t)1 << 31) - 256) {
        tg_set_error("tg_gen_teacher_forced_ws: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    return tg_pgw_forward(w, inp, target, N, L, logp_out, lse_out, logits_out, states_out, workspace, workspace_bytes,
                          (hipStream_t)stream, states_in);
}

extern "C" size_t tg_gen_nll_workspace_bytes(const TgGenWeights* w, int N) {
    if (w && tg_genw_applies(w)) return tg_genw_nll_workspace_bytes(w, N);
    return (w && use_split(*w)) ? split_bytes(make_geom(w->V)) : 0;
}

extern "C" int tg_gen_nll(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                          float* logp_out, float* lse_out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_weights(w, true);
    if (rc) return rc;
    if (N < 0 || L < 1 || (N > 0 && (!inp || !target || !logp_out))) {
        tg_set_error("tg_gen_nll: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    if (tg_genw_applies(w))
        return tg_genw_nll(w, inp, target, N, L, logp_out, lse_out, workspace, workspace_bytes, (hipStream_t)stream);
    GenArgs a = {};
    a.w = *w;
    a.geo = make_geom(w->V);
    a.N = N;
    a.L = L;
    a.inp = inp;
    a.target = target;
    a.logp_out = logp_out;
    a.lse_out = lse_out;
    a.need_proj = 1;
    if (use_split(*w)) {
        if (!workspace || workspace_bytes < tg_gen_nll_workspace_bytes(w, N)) {
            tg_set_error("tg_gen_nll: workspace too small (%zu < %zu)", workspace_bytes, tg_gen_nll_workspace_bytes(w, N));
            return TG_ERR_WORKSPACE;
        }
        rc = prepare_split(a, workspace, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch<false>(a, (N + kRows - 1) / kRows, (hipStream_t)stream);
}

static size_t states_bytes(int B, int L, int H) {
    return (sizeof(float) * (size_t)(L + 1) * B * 2 * H + 255) / 256 * 256;
}

extern "C" size_t tg_gen_rollout_workspace_bytes(const TgGenWeights* w, int B, int L, int R) {
    if (!w || B < 1 || L < 1) return 0;
    if (tg_genw_applies(w)) return states_bytes(B, L, w->H) + tg_genw_rollout_workspace_bytes(w, B, L, R);
    return states_bytes(B, L, w->H) + (use_split(*w) ? split_bytes(make_geom(w->V)) : 0);
}

extern "C" size_t tg_gen_sample_workspace_bytes(const TgGenWeights* w, int N) {
    if (w && tg_genw_applies(w)) return tg_genw_sample_workspace_bytes(w, N);
    return (w && use_split(*w)) ? split_bytes(make_geom(w->V)) : 0;
}

extern "C" int tg_gen_rollout(const TgGenWeights* w, const int64_t* sentences, int B, int L, int R, int g_lo, int g_hi,
                              int i_lo, int i_hi, const float* uniforms, const int64_t* forced, int64_t* completions,
                              void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_weights(w, true);
    if (rc) return rc;
    const bool empty = B == 0 || i_lo == i_hi;   // no sentence or no rollout index: a no-op, whatever the pointers
    if ((!empty && (!sentences || !uniforms || !completions)) || B < 0 || L < 1 || R < 1 || g_lo < 1 || g_hi > L ||
        g_lo > g_hi || i_lo < 0 || i_hi > R || i_lo > i_hi) {
        tg_set_error("tg_gen_rollout: bad arguments (B=%d L=%d R=%d g=[%d,%d] i=[%d,%d))", B, L, R, g_lo, g_hi, i_lo,
                     i_hi);
        return TG_ERR_ARG;
    }
    if (empty) return TG_OK;
    if (!workspace || workspace_bytes < tg_gen_rollout_workspace_bytes(w, B, L, i_hi - i_lo)) {
        tg_set_error("tg_gen_rollout: workspace too small (%zu < %zu)", workspace_bytes,
                     tg_gen_rollout_workspace_bytes(w, B, L, i_hi - i_lo));
        return TG_ERR_WORKSPACE;
    }
    if (tg_genw_applies(w)) {  // shapes beyond E = H = 32 in split-fp16: step-synchronous GEMMs (tg_genw.hip)
        const size_t sb = states_bytes(B, L, w->H);
        return tg_genw_rollout(w, sentences, B, L, g_lo, g_hi, i_lo, i_hi, uniforms, forced, completions,
                               (float*)workspace, reinterpret_cast<char*>(workspace) + sb, workspace_bytes - sb,
                               (hipStream_t)stream);
    }
    // 1) prefix states: teacher-forced pass over the sentences, no projection (rollout.py:35-38)
    rc = tg_gen_teacher_forced(w, sentences, nullptr, B, L, nullptr, nullptr, nullptr, (float*)workspace, stream);
    if (rc) return rc;
    // 2) all completions; tiles ordered by given_num ascending (longest first)
    GenArgs a = {};
    a.w = *w;
    a.geo = make_geom(w->V);
    a.sentences = sentences;
    a.states = (const float*)workspace;
    a.uniforms = uniforms;
    a.forced = forced;
    a.out = completions;
    a.rollout = 1;
    a.stamps = tg_diag_pointer("TG_STAMPS");
    a.B = B;
    a.L = L;
    a.R = R;
    a.g_lo = g_lo;
    a.i_lo = i_lo;
    a.rows_g = (i_hi - i_lo) * B;
    a.tiles_g = (a.rows_g + kRows - 1) / kRows;
    const long grid = (long)a.tiles_g * (g_hi - g_lo + 1);
    if (grid > 0x7fffffffL) {
        tg_set_error("tg_gen_rollout: grid too large");
        return TG_ERR_ARG;
    }
    if (use_split(*w)) {
        rc = prepare_split(a, reinterpret_cast<char*>(workspace) + states_bytes(B, L, w->H), (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch<true>(a, (int)grid, (hipStream_t)stream);
}

extern "C" int tg_gen_sample(const TgGenWeights* w, int N, int L, int64_t start_letter, const float* uniforms,
                             const int64_t* forced, int64_t* out, void* workspace, size_t workspace_bytes,
                             void* stream) {
    int rc = check_weights(w, true);
    if (rc) return rc;
    if (!uniforms || !out || N < 0 || L < 1 || start_letter < 0 || start_letter >= w->V) {
        tg_set_error("tg_gen_sample: bad arguments (N=%d L=%d start=%lld)", N, L, (long long)start_letter);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    if (tg_genw_applies(w))
        return tg_genw_sample(w, N, L, start_letter, uniforms, forced, out, workspace, workspace_bytes, (hipStream_t)stream);
    GenArgs a = {};
    a.w = *w;
    a.geo = make_geom(w->V);
    a.uniforms = uniforms;
    a.forced = forced;
    a.out = out;
    a.start_letter = start_letter;
    a.rollout = 0;
    a.N = N;
    a.L = L;
    if (use_split(*w)) {
        if (!workspace || workspace_bytes < tg_gen_sample_workspace_bytes(w, N)) {
            tg_set_error("tg_gen_sample: workspace too small (%zu < %zu)", workspace_bytes,
                         tg_gen_sample_workspace_bytes(w, N));
            return TG_ERR_WORKSPACE;
        }
        rc = prepare_split(a, workspace, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch<true>(a, (N + kRows - 1) / kRows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// tg_gen_sample_static: tg_gen_sample for callers that cannot allocate (HIP graph capture, serving loops, harnesses
// that forbid hipMalloc).  Same kernels as tg_gen_sample with TG_GEN_SPLIT_F16 at E = H = 32; the split projection's
// scratch (|max| header + W_out fragments, what tg_gen_sample takes as `workspace`) lives in a file-scope device
// array instead.  Every GPU the library runs on holds its own copy of the array until the process ends; its device
// address is looked up once per device (hipGetSymbolAddress) and only that pointer is ever handed to kernels or
// memsets -- the array's name in host code is a host-side stand-in, not a device address.
namespace {
constexpr int kStaticMaxV = TG_STATIC_MAX_V;
constexpr size_t kStaticBytes = 256 + (size_t)4 * ((kStaticMaxV + 127) / 128) * 5 * 64 * sizeof(f32x4);  // split_bytes at kStaticMaxV
constexpr int kStaticMaxDevices = 64;
void* g_static_ptr[kStaticMaxDevices] = {};
std::mutex g_static_mu;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

// Device address of this device's copy of `scratch` (first call on a device looks it up; no allocation).
static int static_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kStaticMaxDevices) {
        tg_set_error("tg_gen_sample_static: hipGetDevice failed or device index %d >= %d", dev, kStaticMaxDevices);
        return TG_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_static_mu);
    if (!g_static_ptr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            tg_set_error("tg_gen_sample_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return TG_ERR_LAUNCH;
        }
        g_static_ptr[dev] = p;
    }
    *out = g_static_ptr[dev];
    return TG_OK;
}

extern "C" size_t tg_gen_sample_static_bytes(void) { return kStaticBytes; }

extern "C" int tg_gen_sample_static(const TgGenWeights* w, int N, int L, int64_t start_letter, const float* uniforms,
                                    const int64_t* forced, int64_t* out, void* stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    if (!use_split(*w)) {
        tg_set_error("tg_gen_sample_static: E = H = 32 with TG_GEN_SPLIT_F16 only (got E=%d H=%d precision=%d); "
                     "TG_GEN_FP32 needs no scratch: call tg_gen_sample with workspace = NULL", w->E, w->H, w->precision);
        return TG_ERR_ARG;
    }
    if (w->V > kStaticMaxV) {
        tg_set_error("tg_gen_sample_static: vocabulary %d exceeds the static scratch's maximum %d (%zu bytes per GPU); "
                     "use tg_gen_sample with a caller-provided workspace", w->V, kStaticMaxV, kStaticBytes);
        return TG_ERR_ARG;
    }
    const SamplerGeom geo = make_geom(w->V);
    if (split_bytes(geo) > kStaticBytes) {  // cannot happen for V <= kStaticMaxV; kept as the guard of the array bound
        tg_set_error("tg_gen_sample_static: scratch of %zu bytes exceeds the static array (%zu)", split_bytes(geo), kStaticBytes);
        return TG_ERR_WORKSPACE;
    }
    void* ptr = nullptr;
    rc = static_scratch(&ptr);
    if (rc) return rc;
    // prepare_split clears the |max| header (the one part that must start at zero) and rewrites every fragment the
    // kernel will read for this vocabulary; what an earlier call left behind past them is never read.
    return tg_gen_sample(w, N, L, start_letter, uniforms, forced, out, ptr, kStaticBytes, stream);
}
