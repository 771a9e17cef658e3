// K-GENW: the LSTM generator for shapes beyond E = H = 32 (the scaled config: V 32k, E = H = 512,
// millions of completions) as step-synchronous GEMMs on split-fp16 MFMAs (gfx950).
//
// At H = 512 the gate weights (8 MB) and W_out (65 MB) cannot live in a workgroup, so the per-row-tile
// persistent kernel of tg_gen.hip re-streams them for every 32 rows.  Here all rows of a chunk advance one
// timestep together and each step is three launches over the rows that are active at it:
//   k_genw_gemm<CELL>  gates[4H x rows] = [W_ih|W_hh] . [x;h]   (+ bias, sigmoid/tanh, c and h update)
//   k_genw_gemm<PROJ>  logits[V x rows] = W_out . h (+ bias), per 16-entry tile a (max, sum 2^(l-max))
//                      record -- logits never reach HBM
//   k_genw_draw        canonical inverse-CDF draw from the records; only the 16
//                      logits of the chosen tile are recomputed            (tg_gen_rollout, tg_gen_sample)
//   k_genw_lse         log-sum-exp from the records + the target's logit    (tg_gen_nll)
// Both GEMMs: 128 x 128 or 256 x 256 tile, K steps of 32 through LDS (two buffers, one barrier per step, next
// step's operands in flight in registers), 32x32x16 fp16 MFMA tiles, three products per operand pair
// (hi.hi + hi.lo + lo.hi, fp32 accumulation) exactly as the split arithmetic of tg_gen.hip / tg_dis.hip.
// Every operand is kept in HBM in MFMA fragment order ("piece" = the 16 B one lane feeds to one MFMA), so
// staging is a linear copy and every LDS read is a conflict-free 1-KiB row.
// Replaces models/generator.py:40-84, utils/rollout.py:25-55 and the forward inside metrics/nll.py:59-61 of the
// reference for those shapes.
#include <stdlib.h>

#include <type_traits>

#include "tg_common.h"
#include "tg_genw.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define TG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr float kHScale = 8192.f;  // h in (-1, 1) -> below 2^13 in fp16
constexpr float kL2E = 1.44269502162933349609375f;
constexpr int kChunkRows = 1 << 18;  // rows (completions) advanced together
constexpr int kMaxNG = 64;           // groups per span the draw kernel handles (V <= 65536)
constexpr int kDrawRows = 16;        // rows per workgroup of the record-scanning kernels (draw, lse)

struct Scales {  // device side, written by k_genw_prep
    float sWout, invProj;  // W_out*log2(e) scale, 1/(sWout * kHScale)
    float sI, sX, sHH, invCell;
};

struct Geom {
    int V, E, H;
    int KBh, KBe, KBc;    // 16-wide k blocks of H, E, E+H
    int MBp;              // 32-entry vocab blocks, padded to whole 256-entry tiles
    int NT16;             // ceil(V/16): 16-entry tiles that hold vocabulary
    int Ls, NT, GK, NG;   // canonical sampler geometry (tg_gen.hip make_geom / oracle orc_span_len)
};

Geom make_geom(const TgGenWeights& w) {
    Geom g;
    g.V = w.V; g.E = w.E; g.H = w.H;
    g.KBh = w.H / 16; g.KBe = w.E / 16; g.KBc = g.KBh + g.KBe;
    g.MBp = 8 * ((w.V + 255) / 256);
    g.NT16 = (w.V + 15) / 16;
    g.Ls = 16 * ((w.V + 127) / 128);
    g.NT = g.Ls / 16;
    g.GK = 1;
    while ((g.NT + g.GK - 1) / g.GK > 16 && g.GK < 8) g.GK *= 2;
    g.NG = (g.NT + g.GK - 1) / g.GK;
    return g;
}

size_t al256(size_t x) { return (x + 255) / 256 * 256; }

struct Carve {  // workspace layout
    unsigned* absmax; Scales* sc; float* bias2; float* gb;
    f32x4* a_proj; f32x4* a_cell; _Float16* emb_hi; _Float16* emb_lo;
    f32x4* hfrag[2]; f32x4* cfrag; float2* rec; int* tok; float* gx;
    size_t bytes;
};

Carve carve(const Geom& g, size_t rowsP, void* base) {
    Carve c;
    char* p = reinterpret_cast<char*>(base);
    size_t o = 0;
    auto take = [&](size_t n) { char* q = p + o; o += al256(n); return q; };
    c.absmax = reinterpret_cast<unsigned*>(take(256));
    c.sc = reinterpret_cast<Scales*>(reinterpret_cast<char*>(c.absmax) + 64);
    c.bias2 = reinterpret_cast<float*>(take(sizeof(float) * g.MBp * 32));
    c.gb = reinterpret_cast<float*>(take(sizeof(float) * 4 * g.H));
    c.a_proj = reinterpret_cast<f32x4*>(take((size_t)g.MBp * g.KBh * 128 * 16));
    c.a_cell = reinterpret_cast<f32x4*>(take((size_t)(4 * g.H / 32) * g.KBc * 128 * 16));
    c.emb_hi = reinterpret_cast<_Float16*>(take((size_t)g.V * g.E * 2));
    c.emb_lo = reinterpret_cast<_Float16*>(take((size_t)g.V * g.E * 2));
    c.gx = reinterpret_cast<float*>(take((size_t)g.V * 4 * g.H * sizeof(float)));
    const size_t RB = rowsP / 32;
    c.hfrag[0] = reinterpret_cast<f32x4*>(take(RB * g.KBh * 128 * 16));
    c.hfrag[1] = reinterpret_cast<f32x4*>(take(RB * g.KBh * 128 * 16));
    c.cfrag = reinterpret_cast<f32x4*>(take(RB * (g.H / 8) * 64 * 16));
    c.rec = reinterpret_cast<float2*>(take((size_t)g.MBp * 2 * rowsP * sizeof(float2)));
    c.tok = reinterpret_cast<int*>(take(rowsP * sizeof(int)));
    c.bytes = o;
    return c;
}

size_t pad128(size_t r) { return (r + 255) / 256 * 256; }  // whole row tiles of the larger GEMM shape

// ------------------------------------------------------------------ per-call weight preparation
__global__ __launch_bounds__(256) void k_genw_absmax(const float* x, size_t n, unsigned* out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
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

// exponent e with m * 2^e in [2^13, 2^14) (0 for m = 0 / inf / nan)
__device__ __forceinline__ int scale_exp(float m) {
    if (!(m > 0.f) || m == __builtin_inff()) return 0;
    int ex = 0;
    (void)frexpf(m, &ex);
    return 14 - ex;
}

// absmax[0..3] = |max| bits of W_out, W_ih, W_hh, emb.  Publishes the power-of-two scales (the two products of the
// cell, W_ih.x and W_hh.h, share one accumulator and therefore one combined scale) and the fp32 side operands:
// bias2[v] = b_out[v]*log2(e) (TG_NEG_BIG past the vocabulary's end), gb = b_ih + b_hh.
__global__ __launch_bounds__(256) void k_genw_prep(const unsigned* absmax, Scales* sc, const float* b_out, int V, int Vp,
                                                   float* bias2, const float* b_ih, const float* b_hh, int H4, float* gb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        const int eW = scale_exp(__uint_as_float(absmax[0]) * kL2E);
        const int eI = scale_exp(__uint_as_float(absmax[1]));
        const int eHH = scale_exp(__uint_as_float(absmax[2]));
        const int eX = scale_exp(__uint_as_float(absmax[3]));
        const int e = (eI + eX < eHH + 13) ? eI + eX : eHH + 13;
        Scales s;
        s.sWout = ldexpf(1.f, eW);
        s.invProj = ldexpf(1.f, -(eW + 13));
        s.sX = ldexpf(1.f, eX);
        s.sI = ldexpf(1.f, e - eX);
        s.sHH = ldexpf(1.f, e - 13);
        s.invCell = ldexpf(1.f, -e);
        *sc = s;
    }
    if (i < Vp) bias2[i] = i < V ? b_out[i] * kL2E : TG_NEG_BIG;
    if (i < H4) gb[i] = b_ih[i] + b_hh[i];
}

__device__ __forceinline__ void split8(const float (&x)[8], f32x4& hi, f32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 xh = (_Float16)x[j];
        h[j] = xh;
        l[j] = (_Float16)(x[j] - (float)xh);
    }
    hi = __builtin_bit_cast(f32x4, h);
    lo = __builtin_bit_cast(f32x4, l);
}

// A fragments.  Piece ((mb*KB + kb)*2 + part)*64 + lane: lane (am, kh) holds the 8 halves k = 16kb + 8kh + 0..7 of
// tile row am, part 0 = fp16 hi, 1 = residual.  Tile row am of block mb is
//   PROJ: vocab entry 32mb + 16((am>>2)&1) + (am&3) + 4(am>>3)   -> D reg r of lane (n, hh) is entry 32mb + 16hh + r:
//         16 consecutive entries (one sampler tile) per lane;   values W_out*log2(e)*sWout
//   CELL: gate row (am>>3)*H + 8mb + (am&7)                       -> D reg r of lane (n, hh) is gate r>>2 of hidden unit
//         8mb + 4hh + (r&3): i,f,g,o of a unit in one lane;      values W_ih*sI (k < E) | W_hh*sHH
template <bool PROJ>
__global__ __launch_bounds__(64) void k_genw_make_a(const float* __restrict__ w0, const float* __restrict__ w1, int V, int E,
                                                    int H, int KB, const Scales* sc, f32x4* __restrict__ dst) {
    const int kb = blockIdx.x, mb = blockIdx.y, lane = threadIdx.x;
    const int am = lane & 31, kh = lane >> 5;
    const int k = 16 * kb + 8 * kh;
    float x[8];
    if constexpr (PROJ) {
        const int v = 32 * mb + 16 * ((am >> 2) & 1) + (am & 3) + 4 * (am >> 3);
        const float s = sc->sWout;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v < V ? (w0[(size_t)v * H + k + j] * kL2E) * s : 0.f;
    } else {
        const int grow = (am >> 3) * H + 8 * mb + (am & 7);
        const bool xpart = k < E;
        const float* src = xpart ? w0 + (size_t)grow * E + k : w1 + (size_t)grow * H + (k - E);
        const float s = xpart ? sc->sI : sc->sHH;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[j] * s;
    }
    f32x4 hi, lo;
    split8(x, hi, lo);
    f32x4* d = dst + ((size_t)(mb * KB + kb) * 2) * 64 + lane;
    d[0] = hi;
    d[64] = lo;
}

// embedding table -> fp16 hi / residual, row-major [V][E], scaled by sX
__global__ __launch_bounds__(256) void k_genw_split_emb(const float* __restrict__ emb, size_t n8, const Scales* sc,
                                                        f32x4* __restrict__ hi, f32x4* __restrict__ lo) {
    const float s = sc->sX;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(emb)[2 * i], b = reinterpret_cast<const f32x4*>(emb)[2 * i + 1];
        const float x[8] = {a[0] * s, a[1] * s, a[2] * s, a[3] * s, b[0] * s, b[1] * s, b[2] * s, b[3] * s};
        f32x4 h, l;
        split8(x, h, l);
        hi[i] = h;
        lo[i] = l;
    }
}

// ------------------------------------------------------------------ state layouts
// hfrag: B fragments of h*2^13.  Piece ((rb*KBh + kb)*2 + part)*64 + lane, lane (j, kh): row 32rb + j, k = 16kb + 8kh + 0..7.
// cfrag: piece (rb*(H/8) + T)*64 + lane, lane (j, hh): c of units 8T + 4hh + 0..3 of row 32rb + j (the cell epilogue's
//        own accumulator layout; nothing else reads it).
__device__ __forceinline__ void store_h4(f32x4* hfrag, int KBh, int row, int T, int hh, const f32x4& h) {
    const int rb = row >> 5, j = row & 31, kb = T >> 1, kh = T & 1;
    const size_t piece = ((size_t)(rb * KBh + kb) * 2) * 64 + kh * 32 + j;
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = h[e] * kHScale;
        const _Float16 xh = (_Float16)x;
        hi[e] = xh;
        lo[e] = (_Float16)(x - (float)xh);
    }
    char* p = reinterpret_cast<char*>(hfrag);
    *reinterpret_cast<f16x4*>(p + piece * 16 + 8 * hh) = hi;
    *reinterpret_cast<f16x4*>(p + (piece + 64) * 16 + 8 * hh) = lo;
}

// rows [r0, r1) take the state after their prefix (tg_gen_teacher_forced's states_out for that step: [B][2][H]);
// row r0 + k belongs to sentence k % B.  src == nullptr: zero state (free-running sample).
__global__ __launch_bounds__(256) void k_genw_enter(const float* __restrict__ src, int B, int H, int KBh, int r0, int r1,
                                                    f32x4* hfrag, f32x4* cfrag) {
    const int q = H >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)(r1 - r0) * q) return;
    const int k = (int)(idx / q), uq = (int)(idx - (size_t)k * q);
    const int row = r0 + k, T = uq >> 1, hh = uq & 1;
    f32x4 h = {0.f, 0.f, 0.f, 0.f}, c = h;
    if (src) {
        const float* s = src + (size_t)(k % B) * 2 * H + 4 * uq;
        h = *reinterpret_cast<const f32x4*>(s);
        c = *reinterpret_cast<const f32x4*>(s + H);
    }
    store_h4(hfrag, KBh, row, T, hh, h);
    cfrag[((size_t)(row >> 5) * (H >> 3) + T) * 64 + hh * 32 + (row & 31)] = c;
}

// Sampler records: one (max, sum) pair per (row, 16-entry tile), in blocks of 16 rows -- [row/16][tile][row%16] -- so
// that the 16 rows one draw workgroup scans are one contiguous range.
__device__ __forceinline__ size_t rec_index(int row, int tile, int ntiles) {
    return ((size_t)(row >> 4) * ntiles + tile) * 16 + (row & 15);
}

// ------------------------------------------------------------------ the GEMM
struct GemmArgs {
    const f32x4* A;   // fragments [MB][KBA][2][64]; this GEMM runs over k blocks kbA0 .. kbA0 + KB - 1 of them
    int KBA, kbA0;
    int KB, MT, RT;   // k blocks, M tiles, N (completion) tiles of the workgroup shape
    int NB;           // logical tiles: 8 * MT * ceil(RT / 8)
    const f32x4* Bh;  // hfrag (PROJ: the whole B operand; CELL: k blocks KBe..)
    int KBh, KBe;
    const _Float16* emb_hi; const _Float16* emb_lo; const int* tok; int E, V;  // CELL: x = emb[tok], gathered
    const Scales* sc;
    const float* bias2; float2* rec; int ntiles;  // PROJ epilogue
    const float* gb; f32x4* cfrag; f32x4* hdst; int H;  // CELL epilogue
    float* gx;        // gate table [V][H/8][2][4][4]: W_ih . emb[v] + b_ih + b_hh in the CELL epilogue's lane order
    int table;        // 1: this launch fills gx (rows = vocabulary entries, x part only); 0: cell update reading it
};

// WM x WN waves, each TM x TN MFMA tiles of 32 x 32: workgroup tile (32 WM TM) x (32 WN TN).
//   <2,2,2,2>  128 x 128, 256 threads, 64 KiB LDS, two workgroups per CU   (small problems, 4H = 128)
//   <2,4,4,2>  256 x 256, 512 threads, 128 KiB LDS, one workgroup per CU: half the operand bytes per flop -- at
//              128 x 128 every CU pulls ~42 B/clk from L2, which bounds the kernel before the matrix pipes do
// PROJ epilogue of one wave: accumulators acc[mi][ni] = tile rows (vocabulary blocks mbw + mi) x completion blocks rbw + ni
template <int TM, int TN>
__device__ __forceinline__ void epilogue_proj(const GemmArgs& a, const f32x16 (&acc)[TM][TN], int mbw, int rbw, int n, int hh) {
    // per (row, 16-entry tile): m = max logit (base 2), s = sum 2^(l - m) in entry order
    const float inv = a.sc->invProj;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int mb = mbw + mi;
        float bias[16];
        const f32x4* pbias = reinterpret_cast<const f32x4*>(a.bias2 + 32 * mb + 16 * hh);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 b4 = pbias[q4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q4 + e] = b4[e];
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int row = (rbw + ni) * 32 + n;
            float l[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) l[r] = __builtin_fmaf(acc[mi][ni][r], inv, bias[r]);
            float m = l[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = __builtin_fmaxf(m, l[r]);
            float ev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ev[r] = __builtin_amdgcn_exp2f(l[r] - m);
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += ev[r];
            // streamed once, read once by the draw kernel: keep it out of the L2 the operand tiles live in
            __builtin_nontemporal_store(__builtin_bit_cast(double, make_float2(m, s)),
                                        reinterpret_cast<double*>(a.rec + rec_index(row, 2 * mb + hh, a.ntiles)));
        }
    }
}

// CELL epilogue of one wave: gate tiles mbw + mi (8 hidden units each) x completion blocks rbw + ni
template <int TM, int TN>
__device__ __forceinline__ void epilogue_cell(const GemmArgs& a, const f32x16 (&acc)[TM][TN], int mbw, int rbw, int n, int hh,
                                              int lane) {
    const float inv = a.sc->invCell;
    const int H = a.H;
    if (a.table) {
        // gx[v][T][hh][gate][unit] = W_ih . emb[v] + b_ih + b_hh: 64 B per lane, the order the update below reads
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int T = mbw + mi;
            const int u0 = 8 * T + 4 * hh;
            f32x4 gbias[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gbias[q] = *reinterpret_cast<const f32x4*>(a.gb + q * H + u0);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int v = (rbw + ni) * 32 + n;
                if (v >= a.V) continue;
                f32x4* d = reinterpret_cast<f32x4*>(a.gx + (((size_t)v * (H >> 3) + T) * 2 + hh) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(acc[mi][ni][4 * q + e], inv, gbias[q][e]);
                    __builtin_nontemporal_store(o, d + q);
                }
            }
        }
        return;
    }
    int tokv[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) tokv[ni] = tg_min(tg_max(a.tok[(rbw + ni) * 32 + n], 0), a.V - 1);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int T = mbw + mi;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int rb = rbw + ni;
            const f32x4* gp = reinterpret_cast<const f32x4*>(a.gx + (((size_t)tokv[ni] * (H >> 3) + T) * 2 + hh) * 16);
            f32x4 gbias[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gbias[q] = __builtin_nontemporal_load(gp + q);  // (a 262 MB table at C3: streamed, not cached)
            f32x4* cp = a.cfrag + ((size_t)rb * (H >> 3) + T) * 64 + lane;
            const f32x4 c_old = *cp;
            f32x4 c_new, h_new;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = tg_sigmoid(__builtin_fmaf(acc[mi][ni][e], inv, gbias[0][e]));
                const float fg = tg_sigmoid(__builtin_fmaf(acc[mi][ni][4 + e], inv, gbias[1][e]));
                const float gg = tg_tanh(__builtin_fmaf(acc[mi][ni][8 + e], inv, gbias[2][e]));
                const float og = tg_sigmoid(__builtin_fmaf(acc[mi][ni][12 + e], inv, gbias[3][e]));
                const float cn = fg * c_old[e] + ig * gg;
                c_new[e] = cn;
                h_new[e] = og * tg_tanh(cn);
            }
            *cp = c_new;
            store_h4(a.hdst, a.KBh, rb * 32 + n, T, hh, h_new);
        }
    }
}

// GLDS: stages go global -> LDS directly (global_load_lds_dwordx4: the image is lane-linear, so the DMA's
// "wave base + 16 * lane" destination is exactly the piece order) instead of through registers.
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <bool PROJ, int WM, int WN, int TM, int TN, bool GLDS = false>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_genw_gemm(const GemmArgs a) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int MBT = WM * TM, RBT = WN * TN;          // 32-blocks per workgroup tile
    constexpr int NLA = MBT * 256 / NTHR, NLB = RBT * 256 / NTHR;  // 16-B loads per thread and stage
    constexpr int STAGE = (MBT + RBT) * 256;             // pieces per stage buffer
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];  // [2 buffers][A MBT*256 | B RBT*256] pieces
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wm = w / WN, wn = w % WN;
    // XCD-aware order: the workgroups that share an XCD (id % 8) walk consecutive logical tile ids; 8 consecutive ids
    // share one A tile, and one group of 8 row tiles stays put while the M tiles sweep (its B stays in that L2).
    // a.NB logical tiles (a multiple of 8) over gridDim.x workgroups (a multiple of 8, <= NB): a workgroup takes every
    // (gridDim.x / 8)-th tile of its XCD's range and fetches the next tile's first stage under the current tile's
    // last one, so the fill latency is paid once per workgroup, not once per tile.
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
    auto next_valid = [&](int lb, int& mb0, int& rb0) {  // first tile at or after lb (in steps of stride) that has rows
        while (lb < lb_end && !decode(lb, mb0, rb0)) lb += stride;
        return lb;
    };
    int mb0 = 0, rb0 = 0;
    int lb = next_valid(((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3), mb0, rb0);
    if (lb >= lb_end) return;
    const int S = a.KB >> 1;
    const int Se = PROJ ? 0 : (a.KBe >> 1);

    // load i of a thread fetches piece p = tid + i*NTHR of the stage: 32-block p >> 8, piece p & 255 within the block's
    // two k blocks ([kbl][part][lane])
    const int pw = tid & 255;
    int tokr[NLB], tokn[NLB];
    auto load_tok = [&](int rb, int (&t)[NLB]) {
        if constexpr (!PROJ) {
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const int row = (rb + ((tid + i * NTHR) >> 8)) * 32 + (tid & 31);
                t[i] = tg_min(tg_max(a.table ? row : a.tok[row], 0), a.V - 1);  // table launch: row v is token v
            }
        }
    };
    load_tok(rb0, tokr);
    f32x4 pa[GLDS ? 1 : NLA], pb[GLDS ? 1 : NLB];
    auto load_stage = [&](int mb, int rb, const int (&tk)[NLB], int s, int buf) {
        f32x4* d = lds + buf * STAGE + tid;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const f32x4* g = PROJ ? a.A + ((size_t)(mb + ((tid + i * NTHR) >> 8)) * a.KB + 2 * s) * 128 + pw
                                  : a.A + ((size_t)(mb + ((tid + i * NTHR) >> 8)) * a.KBA + a.kbA0 + 2 * s) * 128 + pw;
            if constexpr (GLDS) glds16(g, d + i * NTHR);
            else pa[i] = *g;
        }
        if (PROJ || s >= Se) {
            const int sh = s - Se;
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const f32x4* g = a.Bh + ((size_t)(rb + ((tid + i * NTHR) >> 8)) * a.KBh + 2 * sh) * 128 + pw;
                if constexpr (GLDS) glds16(g, d + MBT * 256 + i * NTHR);
                else pb[i] = *g;
            }
        } else {
            const _Float16* tab = ((pw >> 6) & 1) ? a.emb_lo : a.emb_hi;
            const int k = (2 * s + (pw >> 7)) * 16 + 8 * ((pw >> 5) & 1);
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const f32x4* g = reinterpret_cast<const f32x4*>(tab + (size_t)tk[i] * a.E + k);
                if constexpr (GLDS) glds16(g, d + MBT * 256 + i * NTHR);
                else pb[i] = *g;
            }
        }
    };
    auto store_stage = [&](int buf) {
        if constexpr (!GLDS) {
            f32x4* d = lds + buf * STAGE + tid;
#pragma unroll
            for (int i = 0; i < NLA; ++i) d[i * NTHR] = pa[i];
#pragma unroll
            for (int i = 0; i < NLB; ++i) d[MBT * 256 + i * NTHR] = pb[i];
        }
    };
    f32x16 acc[TM][TN];

    load_stage(mb0, rb0, tokr, 0, 0);
    store_stage(0);
    __syncthreads();
    int gs = 0;  // stages done by this workgroup: buffer parity
    for (;;) {
    int nmb0 = 0, nrb0 = 0;
    const int nlb = next_valid(lb + stride, nmb0, nrb0);
    const bool has_next = nlb < lb_end;
    if (has_next) load_tok(nrb0, tokn);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int s = 0; s < S; ++s, ++gs) {
        const bool more = s + 1 < S || has_next;
        if (s + 1 < S) load_stage(mb0, rb0, tokr, s + 1, (gs + 1) & 1);
        else if (has_next) load_stage(nmb0, nrb0, tokn, 0, (gs + 1) & 1);
        const f32x4* la = lds + (gs & 1) * STAGE + (TM * wm) * 256 + lane;
        const f32x4* lbp = lds + (gs & 1) * STAGE + MBT * 256 + (TN * wn) * 256 + lane;
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
            // product-major: TM*TN independent accumulators between two MFMAs on the same one
            __builtin_amdgcn_s_setprio(1);  // the wave that has its operands runs its MFMA block through (-1.5 % on PROJ)
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
            // the next stage goes to the other buffer (last read before the previous barrier) while the first k
            // block's MFMAs are still draining
            if (kbl == 0 && more) store_stage((gs + 1) & 1);
        }
        __syncthreads();
    }

    if constexpr (PROJ) epilogue_proj<TM, TN>(a, acc, mb0 + TM * wm, rb0 + TN * wn, n, hh);
    else epilogue_cell<TM, TN>(a, acc, mb0 + TM * wm, rb0 + TN * wn, n, hh, lane);
    if (!has_next) break;
    lb = nlb;
    mb0 = nmb0;
    rb0 = nrb0;
#pragma unroll
    for (int i = 0; i < NLB; ++i) tokr[i] = tokn[i];
    }  // tiles of this workgroup
}

// prefix chain (the states rollout rows enter with, utils/rollout.py:35-38): rows 0..B-1 are the sentences themselves,
// advanced by teacher forcing; after each step their state is exported in tg_gen_teacher_forced's layout [B][2][H].
__global__ __launch_bounds__(256) void k_genw_tok_col(const int64_t* __restrict__ sentences, int B, int L, int t, int V,
                                                      int* __restrict__ tok) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) tok[b] = tg_min(tg_max((int)sentences[(size_t)b * L + t], 0), V - 1);
}

// Teacher-forced log-probabilities from the tile records (metrics/nll.py:59-61): 16 rows per workgroup, 32 threads per
// row.  lse = log-sum-exp over the records (per-thread online (max, sum), merged in fp64); the target's own logit is
// an fp32 dot product of its W_out row with h = (hi + lo) / 2^13.
struct LseArgs {
    Geom g;
    const float2* rec; int ntiles;
    int rows, t, L;
    const int64_t* target;  // [rows, L] (chunk base)
    float* logp; float* lse;  // [rows, L] (chunk base); lse nullable
    const f32x4* hfrag; const float* w_out; const float* b_out;
};
__global__ __launch_bounds__(512) void k_genw_lse(const LseArgs a) {
    __shared__ float mb[kDrawRows][32];
    __shared__ double sb[kDrawRows][32];
    __shared__ float db[kDrawRows][32];
    const int tid = threadIdx.x, r16 = tid & 15, sg = tid >> 4;
    const int row = blockIdx.x * kDrawRows + r16;
    const bool valid = row < a.rows;
    const int rowc = valid ? row : a.rows - 1;
    const int NT16 = a.g.NT16, V = a.g.V, H = a.g.H;
    const float2* rp = a.rec + rec_index(rowc, 0, a.ntiles);
    float m = TG_NEG_BIG, s = 0.f;
    for (int q0 = sg; q0 < NT16; q0 += 32 * 8) {
        float2 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = q0 + 32 * i;
            const float2 v = rp[tg_min(q, NT16 - 1) * 16];
            r[i] = q < NT16 ? v : make_float2(TG_NEG_BIG, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float mn = __builtin_fmaxf(m, r[i].x);
            s = s * __builtin_amdgcn_exp2f(m - mn) + r[i].y * __builtin_amdgcn_exp2f(r[i].x - mn);
            m = mn;
        }
    }
    mb[r16][sg] = m;
    // the target's logit: thread sg takes the 16-B slices sg, sg+32, .. of the weight row
    const size_t o = (size_t)rowc * a.L + a.t;
    const int tv = tg_min(tg_max((int)a.target[o], 0), V - 1);
    {
        const int rb = rowc >> 5, j = rowc & 31;
        const char* hb = reinterpret_cast<const char*>(a.hfrag);
        float d = 0.f;
        for (int k4 = sg; k4 < (H >> 2); k4 += 32) {
            const int k0 = 4 * k4;
            const size_t piece = ((size_t)(rb * a.g.KBh + (k0 >> 4)) * 2) * 64 + ((k0 >> 3) & 1) * 32 + j;
            const f16x4 hi = *reinterpret_cast<const f16x4*>(hb + piece * 16 + 2 * (k0 & 7));
            const f16x4 lo = *reinterpret_cast<const f16x4*>(hb + (piece + 64) * 16 + 2 * (k0 & 7));
            const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w_out + (size_t)tv * H + k0);
#pragma unroll
            for (int e = 0; e < 4; ++e) d = __builtin_fmaf(wv[e], ((float)hi[e] + (float)lo[e]) * (1.f / kHScale), d);
        }
        db[r16][sg] = d;
    }
    __syncthreads();
    float M = mb[r16][0];
#pragma unroll
    for (int i = 1; i < 32; ++i) M = __builtin_fmaxf(M, mb[r16][i]);
    sb[r16][sg] = (double)s * exp2((double)m - (double)M);
    __syncthreads();
    if (tid < kDrawRows && valid) {
        double S = 0.0;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { S += sb[r16][i]; d += db[r16][i]; }
        const float lse = (float)(((double)M + log2(S)) * 0.69314718055994530942);
        if (a.lse) a.lse[o] = lse;
        a.logp[o] = (d + a.b_out[tv]) - lse;
    }
}
__global__ __launch_bounds__(256) void k_genw_export(const f32x4* __restrict__ hfrag, const f32x4* __restrict__ cfrag, int B,
                                                     int H, int KBh, float* __restrict__ dst) {
    const int q = H >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * q) return;
    const int row = (int)(idx / q), uq = (int)(idx - (size_t)row * q);
    const int T = uq >> 1, hh = uq & 1;
    const int rb = row >> 5, j = row & 31, kb = T >> 1, kh = T & 1;
    const size_t piece = ((size_t)(rb * KBh + kb) * 2) * 64 + kh * 32 + j;
    const char* p = reinterpret_cast<const char*>(hfrag);
    const f16x4 hi = *reinterpret_cast<const f16x4*>(p + piece * 16 + 8 * hh);
    const f16x4 lo = *reinterpret_cast<const f16x4*>(p + (piece + 64) * 16 + 8 * hh);
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = ((float)hi[e] + (float)lo[e]) * (1.f / kHScale);  // exact: 22 bits, power-of-two scale
    float* d = dst + (size_t)row * 2 * H + 4 * uq;
    *reinterpret_cast<f32x4*>(d) = h;
    *reinterpret_cast<f32x4*>(d + H) = cfrag[((size_t)rb * (H >> 3) + T) * 64 + hh * 32 + j];
}


// ------------------------------------------------------------------ row <-> output mapping
struct RowMap {
    int rollout;        // 1: rows are (given_num, rollout index, sentence) of tg_gen_rollout; 0: tg_gen_sample rows
    int B, L, nI, g_lo, i_lo;
    int N, n0;          // sample: total rows, first row of the chunk
};
__device__ __forceinline__ void map_row(const RowMap& m, int row, int t, size_t& orow, size_t& uidx) {
    if (m.rollout) {
        const int nIB = m.nI * m.B;
        const int gi = row / nIB, rem = row - gi * nIB, il = rem / m.B, b = rem - il * m.B;
        const size_t ig = (size_t)(m.i_lo + il) * m.L + (m.g_lo + gi - 1);
        orow = ig * m.B + b;
        uidx = (ig * m.L + t) * m.B + b;
    } else {
        orow = (size_t)m.n0 + row;
        uidx = (size_t)t * m.N + orow;
    }
}

// out[i, g-1, b, :g] = sentences[b, :g] for every (i, g, b) of the chunk, g = L included (utils/rollout.py:42)
__global__ __launch_bounds__(256) void k_genw_prefix(const int64_t* __restrict__ sentences, int64_t* __restrict__ out, int B,
                                                     int L, int nI, int i_lo, int g_lo, int nG) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)nG * nI * B * L;
    if (idx >= total) return;
    const int s = (int)(idx % L);
    const size_t r = idx / L;
    const int b = (int)(r % B);
    const size_t r2 = r / B;
    const int il = (int)(r2 % nI), gi = (int)(r2 / nI);
    const int g = g_lo + gi;
    if (s < g) out[(((size_t)(i_lo + il) * L + (g - 1)) * B + b) * L + s] = sentences[(size_t)b * L + s];
}

// ------------------------------------------------------------------ canonical draw from the tile records
struct DrawArgs {
    Geom g;
    RowMap map;
    const float2* rec; int ntiles;
    int A, t;                    // active rows, position being drawn
    const float* uniforms; const int64_t* forced; int64_t* out; int* tok;
    const f32x4* hfrag; const float* w_out; const float* bias2;
};

// 16 rows per workgroup of 512 threads.  Same decisions as oracle canonical_draw: tile sums (fp32, entry order) ->
// group sums (fp32, tile order) -> span sums / total / target / every prefix comparison in fp64.  Tile sums come from
// the records at their own maximum and are moved to the group's maximum in fp32, the group sums to the row's maximum
// in fp64.
//   scan     thread (row, span, quarter): GPT of the span's <= 4 GPT groups, their records fetched 32 at a time
//            (GPT = 8: up to 32 groups per span, V <= 32768; GPT = 16: up to 64, V <= 65536)
//   select   thread (row, span): span sum in group order, then the owner of the chosen span walks groups and tiles
//   logits   32 threads per row recompute the chosen tile's 16 logits (4 entries x an eighth of k each)
template <int GPT>
__global__ __launch_bounds__(512) void k_genw_draw(const DrawArgs a) {
    __shared__ double cqb[kDrawRows][8][4 * GPT];  // (max, sum) pairs of the groups, then their fp64 contributions
    __shared__ float mxb[kDrawRows][32];
    __shared__ double sbuf[8][kDrawRows];
    __shared__ int selT[kDrawRows];
    __shared__ double selp[kDrawRows], tgt[kDrawRows];
    __shared__ float Mrow[kDrawRows];
    __shared__ float wbuf[kDrawRows][17];
    const int tid = threadIdx.x, r16 = tid & 15, sg = tid >> 4;
    const int row = blockIdx.x * kDrawRows + r16;
    const bool valid = row < a.A;
    const int rowc = valid ? row : a.A - 1;
    const int NT = a.g.NT, GK = a.g.GK, V = a.g.V;
    auto span_tiles = [&](int s) { return tg_min(tg_max(a.g.NT16 - s * NT, 0), NT); };  // tiles that hold vocabulary
    auto load_group = [&](const float2* rp, int ntv, int gq, float2 (&r)[8]) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int q = gq * GK + tt;
            const float2 v = rp[tg_min(q, tg_max(ntv - 1, 0)) * 16];  // always a mapped address
            r[tt] = (tt < GK && q < ntv) ? v : make_float2(TG_NEG_BIG, 0.f);
        }
    };
    auto group_max = [&](const float2 (&r)[8]) {
        float m = r[0].x;
#pragma unroll
        for (int tt = 1; tt < 8; ++tt) m = __builtin_fmaxf(m, r[tt].x);
        return m;
    };
    // ---- scan
    float mgk[GPT], cgk[GPT];
    {
        const int sp = sg >> 2, gp = sg & 3;
        const int ntv = span_tiles(sp);
        const float2* rp = a.rec + rec_index(rowc, sp * NT, a.ntiles);
        float mloc = TG_NEG_BIG;
#pragma unroll
        for (int bt = 0; bt < GPT / 4; ++bt) {
            float2 r[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) load_group(rp, ntv, gp * GPT + bt * 4 + k, r[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = group_max(r[k]);
                float c = 0.f;
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) c += r[k][tt].y * __builtin_amdgcn_exp2f(r[k][tt].x - m);
                mgk[bt * 4 + k] = m;
                cgk[bt * 4 + k] = c;
                mloc = __builtin_fmaxf(mloc, m);
            }
        }
        mxb[r16][sg] = mloc;
    }
    __syncthreads();
    float M = mxb[r16][0];
#pragma unroll
    for (int i = 1; i < 32; ++i) M = __builtin_fmaxf(M, mxb[r16][i]);
    {
        const int sp = sg >> 2, gp = sg & 3;
#pragma unroll
        for (int k = 0; k < GPT; ++k)  // groups past the span's end have sum 0
            cqb[r16][sp][gp * GPT + k] = (double)cgk[k] * exp2((double)mgk[k] - (double)M);
    }
    __syncthreads();
    // ---- select
    const int sp = sg;  // threads tid < 128: (row r16, span sg)
    const int ntv = tid < 128 ? span_tiles(sp) : 0;
    const int ngv = (ntv + GK - 1) / GK;
    if (tid < 128) {
        double ssum = 0.0;
        for (int gq = 0; gq < ngv; ++gq) ssum += cqb[r16][sp][gq];
        sbuf[sp][r16] = ssum;
    }
    __syncthreads();
    if (tid < 128) {
        double S = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) S += sbuf[s][r16];
        size_t orow, uidx;
        map_row(a.map, rowc, a.t, orow, uidx);
        const double target = (double)a.uniforms[uidx] * S;
        const int last_span = (V - 1) / a.g.Ls;
        int sel = last_span;
        double p = 0.0;
        for (int s = 0; s < 8; ++s) {
            const double ss = sbuf[s][r16];
            if (p + ss > target || s == last_span) { sel = s; break; }
            p += ss;
        }
        if (sel == sp) {
            int gsel = ngv - 1;
            for (int gq = 0; gq < ngv; ++gq) {
                const double c = cqb[r16][sp][gq];
                if (p + c > target || gq == ngv - 1) { gsel = gq; break; }
                p += c;
            }
            float2 r[8];
            load_group(a.rec + rec_index(rowc, sp * NT, a.ntiles), ntv, gsel, r);
            const float m = group_max(r);  // the value the scan used for this group
            const double fac = exp2((double)m - (double)M);
            const int ntg = tg_min(GK, ntv - gsel * GK);
            int tsel = ntg - 1;
            bool found = false;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                if (tt < ntg && !found) {
                    const double ts = (double)(r[tt].y * __builtin_amdgcn_exp2f(r[tt].x - m)) * fac;
                    if (p + ts > target || tt == ntg - 1) { tsel = tt; found = true; }
                    else p += ts;
                }
            }
            selT[r16] = sp * NT + gsel * GK + tsel;
            selp[r16] = p;
            tgt[r16] = target;
            Mrow[r16] = M;
        }
    }
    __syncthreads();
    // ---- the chosen tile's 16 logits again, fp32: thread (row, eq, c) takes entries 4eq..4eq+3 and the 16-B column
    // slices c, c+8, .. of their weight rows (the 8 threads of an entry group read each 128 B of a row as one line)
    // against h = (hi + lo) / 2^13, the very operand of the first pass; partial dots summed across the 8 lanes
    {
        const int rr = tid >> 5, eq = (tid >> 3) & 3, c = tid & 7;
        const int rrow = tg_min(blockIdx.x * kDrawRows + rr, a.A - 1);
        const int v0 = selT[rr] * 16 + 4 * eq;
        const float Mr = Mrow[rr];
        const int H = a.g.H, KBh = a.g.KBh;
        const int rb = rrow >> 5, j = rrow & 31;
        const char* hb = reinterpret_cast<const char*>(a.hfrag);
        auto dpp_add = [](float x, auto ctrl) {
            constexpr int C = decltype(ctrl)::value;
            const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, 0xf, 0xf, true);
            return x + __builtin_bit_cast(float, y);
        };
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int ks = 0; ks < (H >> 5); ++ks) {
            const int k0 = 32 * ks + 4 * c;
            const size_t piece = ((size_t)(rb * KBh + (k0 >> 4)) * 2) * 64 + ((k0 >> 3) & 1) * 32 + j;
            const f16x4 hi = *reinterpret_cast<const f16x4*>(hb + piece * 16 + 2 * (k0 & 7));
            const f16x4 lo = *reinterpret_cast<const f16x4*>(hb + (piece + 64) * 16 + 2 * (k0 & 7));
            float h4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[e] = ((float)hi[e] + (float)lo[e]) * (1.f / kHScale);
            f32x4 wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wv[i] = *reinterpret_cast<const f32x4*>(a.w_out + (size_t)tg_min(v0 + i, V - 1) * H + k0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float p = part[i];
                p = __builtin_fmaf(wv[i][0], h4[0], p);
                p = __builtin_fmaf(wv[i][1], h4[1], p);
                p = __builtin_fmaf(wv[i][2], h4[2], p);
                p = __builtin_fmaf(wv[i][3], h4[3], p);
                part[i] = p;
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float p = part[i];
            p = dpp_add(p, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
            p = dpp_add(p, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
            p = dpp_add(p, std::integral_constant<int, 0x141>{});  // row_half_mirror
            mine = (i == c) ? p : mine;
        }
        if (c < 4) {
            const int v = v0 + c;
            float wgt = 0.f;
            if (v < V) wgt = __builtin_amdgcn_exp2f(__builtin_fmaf(mine, kL2E, a.bias2[v]) - Mr);
            wbuf[rr][4 * eq + c] = wgt;
        }
    }
    __syncthreads();
    if (tid < kDrawRows) {
        const int v0 = selT[r16] * 16;
        const int t_hi = tg_min(V, v0 + 16);
        const double target = tgt[r16];
        const double p = selp[r16];
        int drawn = t_hi - 1;
        float pp = 0.f;
        bool hit = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pp += wbuf[r16][r];
            if (!hit && v0 + r < t_hi && p + (double)pp > target) { drawn = v0 + r; hit = true; }
        }
        if (valid) {
            size_t orow, uidx;
            map_row(a.map, row, a.t, orow, uidx);
            a.out[orow * a.map.L + a.t] = drawn;
            int nt = drawn;
            if (a.forced) nt = tg_min(tg_max((int)a.forced[orow * a.map.L + a.t], 0), V - 1);
            a.tok[row] = nt;
        }
    }
}

__global__ __launch_bounds__(256) void k_genw_fill_tok(int* tok, size_t n, int v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) tok[i] = v;
}

// ------------------------------------------------------------------ host side
#define GW_CHECK_LAUNCH(what)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            tg_set_error("%s launch: %s", what, hipGetErrorString(e_));    \
            return TG_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

// First character of an environment switch -- in `make DIAG=1` builds only; the shipped library reads no environment.
char diag_env(const char* name) {
#if TG_DIAG
    const char* e = getenv(name);
    return e ? e[0] : 0;
#else
    (void)name;
    return 0;
#endif
}

template <bool PROJ, int WM, int WN, int TM, int TN, bool GLDS = false>
int gemm_lds_attr() {
    constexpr int bytes = 2 * (WM * TM + WN * TN) * 256 * 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_genw_gemm<PROJ, WM, WN, TM, TN, GLDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        tg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

int ensure_lds() {
    // per device (the attribute applies to the device that is current when it is set)
    static std::atomic<bool> done_on[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        tg_set_error("tg_genw: hipGetDevice failed or device index out of range");
        return TG_ERR_LAUNCH;
    }
    std::atomic<bool>& done = done_on[dev];
    if (done.load(std::memory_order_relaxed)) return TG_OK;
    int rc = gemm_lds_attr<true, 2, 2, 2, 2>();
    if (!rc) rc = gemm_lds_attr<false, 2, 2, 2, 2>();
    if (!rc) rc = gemm_lds_attr<true, 2, 4, 4, 2>();
    if (!rc) rc = gemm_lds_attr<false, 2, 4, 4, 2>();
    if (!rc) rc = gemm_lds_attr<true, 2, 2, 2, 2, true>();
    if (!rc) rc = gemm_lds_attr<false, 2, 2, 2, 2, true>();
    if (!rc) rc = gemm_lds_attr<true, 2, 4, 4, 2, true>();
    if (!rc) rc = gemm_lds_attr<false, 2, 4, 4, 2, true>();
    if (rc) return rc;
    done.store(true, std::memory_order_relaxed);
    return TG_OK;
}

template <bool PROJ, int WM, int WN, int TM, int TN, bool GLDS = false>
void launch_gemm_t(GemmArgs a, int m_blocks, int rows, hipStream_t st) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int bytes = 2 * (WM * TM + WN * TN) * 256 * 16;
    a.MT = m_blocks * 32 / BM;
    a.RT = (rows + BN - 1) / BN;
    a.NB = (a.RT + 7) / 8 * 8 * a.MT;
    // as many workgroups as the chip holds at once (256 CUs x workgroups per CU by LDS), each walking its share of the
    // tiles (TG_GENW_PERSIST=0 in a DIAG build: one workgroup per tile)
    const int resident = 256 * (160 * 1024 / bytes);
    const int grid = diag_env("TG_GENW_PERSIST") == '0' ? a.NB : (a.NB < resident ? a.NB : resident);
    hipLaunchKernelGGL((k_genw_gemm<PROJ, WM, WN, TM, TN, GLDS>), dim3(grid), dim3(64 * WM * WN), bytes, st, a);
}

int prep_weights(const TgGenWeights& w, const Geom& g, const Carve& c, hipStream_t st) {
    if (hipMemsetAsync(c.absmax, 0, 64, st) != hipSuccess) {
        tg_set_error("hipMemsetAsync failed");
        return TG_ERR_LAUNCH;
    }
    auto amax = [&](const float* x, size_t n, int slot) {
        const size_t blocks = (n + 1023) / 1024;
        hipLaunchKernelGGL(k_genw_absmax, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, x, n,
                           c.absmax + slot);
    };
    amax(w.w_out, (size_t)g.V * g.H, 0);
    amax(w.w_ih, (size_t)4 * g.H * g.E, 1);
    amax(w.w_hh, (size_t)4 * g.H * g.H, 2);
    amax(w.emb, (size_t)g.V * g.E, 3);
    const int Vp = g.MBp * 32, H4 = 4 * g.H;
    const int nmax = Vp > H4 ? Vp : H4;
    hipLaunchKernelGGL(k_genw_prep, dim3((nmax + 255) / 256), dim3(256), 0, st, c.absmax, c.sc, w.b_out, g.V, Vp, c.bias2,
                       w.b_ih, w.b_hh, H4, c.gb);
    hipLaunchKernelGGL(k_genw_make_a<true>, dim3(g.KBh, g.MBp), dim3(64), 0, st, w.w_out, (const float*)nullptr, g.V, g.E,
                       g.H, g.KBh, c.sc, c.a_proj);
    hipLaunchKernelGGL(k_genw_make_a<false>, dim3(g.KBc, 4 * g.H / 32), dim3(64), 0, st, w.w_ih, w.w_hh, g.V, g.E, g.H,
                       g.KBc, c.sc, c.a_cell);
    const size_t n8 = (size_t)g.V * g.E / 8;
    const size_t blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(k_genw_split_emb, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, w.emb, n8, c.sc,
                       reinterpret_cast<f32x4*>(c.emb_hi), reinterpret_cast<f32x4*>(c.emb_lo));
    GW_CHECK_LAUNCH("k_genw weight preparation");
    return TG_OK;
}

// 256 x 256 tiles once the problem fills the chip with them and M divides; 128 x 128 otherwise
bool big_tiles(int m_blocks, int rows) {
    const char force = diag_env("TG_GENW_TILE");  // DIAG build: "128" or "256"
    if (m_blocks % 8) return false;
    if (force == '1') return false;
    if (force == '2') return true;
    return (long)(m_blocks / 8) * ((rows + 255) / 256) >= 256;
}

bool use_glds() { return diag_env("TG_GENW_GLDS") != '0'; }  // DIAG build: "0" = register staging in the 256 x 256 GEMMs

// which = 0: vocabulary projection + sampler records; 1: cell update (W_hh . h on the matrix cores, the W_ih . x part
// and the bias from the gate table); 2: fills the gate table for every vocabulary entry (once per call)
void launch_gemm(int which, const TgGenWeights& /*w*/, const Geom& g, const Carve& c, size_t rowsP, int rows, int hsrc,
                 int hdst, hipStream_t st) {
    GemmArgs a = {};
    a.KBh = g.KBh;
    a.sc = c.sc;
    a.H = g.H;
    a.E = g.E;
    a.V = g.V;
    a.Bh = c.hfrag[hsrc];
    if (which == 0) {
        a.A = c.a_proj;
        a.KB = a.KBA = g.KBh;
        a.bias2 = c.bias2;
        a.rec = c.rec;
        a.ntiles = g.MBp * 2;
        tg_prof_begin(TG_KERNEL_GENW_PROJ, st);
        if (big_tiles(g.MBp, rows) && use_glds()) launch_gemm_t<true, 2, 4, 4, 2, true>(a, g.MBp, rows, st);
        else if (big_tiles(g.MBp, rows)) launch_gemm_t<true, 2, 4, 4, 2>(a, g.MBp, rows, st);
        else if (use_glds()) launch_gemm_t<true, 2, 2, 2, 2, true>(a, g.MBp, rows, st);
        else launch_gemm_t<true, 2, 2, 2, 2>(a, g.MBp, rows, st);
        tg_prof_end(st);
    } else {
        a.A = c.a_cell;
        a.KBA = g.KBc;
        a.emb_hi = c.emb_hi;
        a.emb_lo = c.emb_lo;
        a.tok = c.tok;
        a.gb = c.gb;
        a.gx = c.gx;
        a.cfrag = c.cfrag;
        a.hdst = c.hfrag[hdst];
        if (which == 2) {  // x part: k blocks [0, KBe), every stage gathered from the embedding tables
            a.table = 1;
            a.kbA0 = 0;
            a.KB = a.KBe = g.KBe;
            rows = g.V;
        } else {           // h part: k blocks [KBe, KBe + KBh)
            a.kbA0 = g.KBe;
            a.KB = g.KBh;
            a.KBe = 0;
        }
        if (which == 1) tg_prof_begin(TG_KERNEL_GENW_CELL, st);
        if (big_tiles(4 * g.H / 32, rows) && use_glds()) launch_gemm_t<false, 2, 4, 4, 2, true>(a, 4 * g.H / 32, rows, st);
        else if (big_tiles(4 * g.H / 32, rows)) launch_gemm_t<false, 2, 4, 4, 2>(a, 4 * g.H / 32, rows, st);
        else if (use_glds()) launch_gemm_t<false, 2, 2, 2, 2, true>(a, 4 * g.H / 32, rows, st);
        else launch_gemm_t<false, 2, 2, 2, 2>(a, 4 * g.H / 32, rows, st);
        if (which == 1) tg_prof_end(st);
    }
}

// scales, operand fragments, fp16 embedding tables, then the gate table of every vocabulary entry (one GEMM per call)
int prep_all(const TgGenWeights& w, const Geom& g, const Carve& c, size_t rowsP, hipStream_t st) {
    int rc = prep_weights(w, g, c, st);
    if (rc) return rc;
    launch_gemm(2, w, g, c, rowsP, g.V, 0, 0, st);
    GW_CHECK_LAUNCH("k_genw gate table");
    return TG_OK;
}

void launch_draw(const TgGenWeights& w, const Geom& g, const Carve& c, size_t rowsP, const RowMap& map, int rows, int t,
                 int hsrc, const float* uniforms, const int64_t* forced, int64_t* out, hipStream_t st) {
    DrawArgs d = {};
    d.g = g;
    d.map = map;
    d.rec = c.rec;
    d.ntiles = g.MBp * 2;
    d.A = rows;
    d.t = t;
    d.uniforms = uniforms;
    d.forced = forced;
    d.out = out;
    d.tok = c.tok;
    d.hfrag = c.hfrag[hsrc];
    d.w_out = w.w_out;
    d.bias2 = c.bias2;
    tg_prof_begin(TG_KERNEL_GENW_DRAW, st);
    if (g.NG <= 32) hipLaunchKernelGGL(k_genw_draw<8>, dim3((rows + kDrawRows - 1) / kDrawRows), dim3(512), 0, st, d);
    else hipLaunchKernelGGL(k_genw_draw<16>, dim3((rows + kDrawRows - 1) / kDrawRows), dim3(512), 0, st, d);
    tg_prof_end(st);
}

int check_shape(const TgGenWeights& w, const Geom& g) {
    if (w.H % 32 || w.E % 32 || g.NG > kMaxNG) {
        tg_set_error("wide generator path: needs E, H multiples of 32 and V <= 65536 (V=%d E=%d H=%d)", w.V, w.E, w.H);
        return TG_ERR_ARG;
    }
    return TG_OK;
}

}  // namespace

bool tg_genw_applies(const TgGenWeights* w) {
    return w->precision == TG_GEN_SPLIT_F16 && !(w->E == 32 && w->H == 32) && w->H % 32 == 0 && w->E % 32 == 0 &&
           w->H >= 32 && w->E >= 32 && w->H <= 1024 && w->E <= 1024 && w->V >= 1 && make_geom(*w).NG <= kMaxNG;
}

static size_t chunk_rows_rollout(int B, int L, int n_i, int* nIc_out) {
    const size_t per_i = (size_t)B * L;
    size_t nIc = kChunkRows / per_i;
    if (nIc < 1) nIc = 1;
    if ((size_t)n_i < nIc) nIc = n_i > 0 ? n_i : 1;
    if (nIc_out) *nIc_out = (int)nIc;
    return pad128(nIc * per_i);
}

size_t tg_genw_rollout_workspace_bytes(const TgGenWeights* w, int B, int L, int n_i) {
    const Geom g = make_geom(*w);
    return carve(g, chunk_rows_rollout(B, L, n_i, nullptr), nullptr).bytes;
}

size_t tg_genw_sample_workspace_bytes(const TgGenWeights* w, int N) {
    const Geom g = make_geom(*w);
    const size_t rows = (size_t)N < (size_t)kChunkRows ? (size_t)(N > 0 ? N : 1) : (size_t)kChunkRows;
    return carve(g, pad128(rows), nullptr).bytes;
}

int tg_genw_rollout(const TgGenWeights* w, const int64_t* sentences, int B, int L, int g_lo, int g_hi, int i_lo, int i_hi,
                    const float* uniforms, const int64_t* forced, int64_t* completions, float* states,
                    void* workspace, size_t workspace_bytes, hipStream_t st) {
    const Geom g = make_geom(*w);
    int rc = check_shape(*w, g);
    if (rc) return rc;
    int nIc = 1;
    const size_t rowsP = chunk_rows_rollout(B, L, i_hi - i_lo, &nIc);
    const Carve c = carve(g, rowsP, workspace);
    if (!workspace || workspace_bytes < c.bytes) {
        tg_set_error("wide generator path: workspace too small (%zu < %zu)", workspace_bytes, c.bytes);
        return TG_ERR_WORKSPACE;
    }
    if ((rc = ensure_lds())) return rc;
    if ((rc = prep_all(*w, g, c, rowsP, st))) return rc;
    const int g_hs = g_hi < L - 1 ? g_hi : L - 1;  // given_num = L only copies
    if (g_hs >= g_lo) {
        // states[t] = (h, c) of every sentence after its first t tokens, t = 1..g_hs, with the same cell GEMM
        const size_t n = (size_t)B * (g.H / 4);
        hipLaunchKernelGGL(k_genw_enter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)nullptr, 1, g.H,
                           g.KBh, 0, B, c.hfrag[0], c.cfrag);
        int p = 0;
        for (int t = 1; t <= g_hs; ++t) {
            hipLaunchKernelGGL(k_genw_tok_col, dim3((B + 255) / 256), dim3(256), 0, st, sentences, B, L, t - 1, g.V, c.tok);
            launch_gemm(1, *w, g, c, rowsP, B, p, 1 - p, st);
            p = 1 - p;
            if (t >= g_lo)
                hipLaunchKernelGGL(k_genw_export, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c.hfrag[p], c.cfrag,
                                   B, g.H, g.KBh, states + (size_t)t * B * 2 * g.H);
        }
        GW_CHECK_LAUNCH("k_genw prefix chain");
    }
    for (int ic = i_lo; ic < i_hi; ic += nIc) {
        const int nI = (i_hi - ic) < nIc ? (i_hi - ic) : nIc;
        const int nIB = nI * B;
        {
            const int nG = g_hi - g_lo + 1;
            const size_t total = (size_t)nG * nIB * L;
            hipLaunchKernelGGL(k_genw_prefix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sentences,
                               completions, B, L, nI, ic, g_lo, nG);
        }
        if (g_hs < g_lo) continue;
        RowMap map = {};
        map.rollout = 1;
        map.B = B;
        map.L = L;
        map.nI = nI;
        map.g_lo = g_lo;
        map.i_lo = ic;
        if (hipMemsetAsync(c.tok, 0, rowsP * sizeof(int), st) != hipSuccess) {
            tg_set_error("hipMemsetAsync failed");
            return TG_ERR_LAUNCH;
        }
        int p = 0;
        for (int t = g_lo; t < L; ++t) {
            const int a_prev = t - 1 >= g_lo ? ((t - 1 < g_hs ? t - 1 : g_hs) - g_lo + 1) * nIB : 0;
            const int a_t = ((t < g_hs ? t : g_hs) - g_lo + 1) * nIB;
            // rows that drew position t-1 take a cell step on that token; rows with given_num == t enter with the
            // state after their t prefix tokens (utils/rollout.py:35-38)
            if (a_prev > 0) launch_gemm(1, *w, g, c, rowsP, a_prev, p, 1 - p, st);
            if (t <= g_hs) {
                const size_t n = (size_t)(a_t - a_prev) * (g.H / 4);
                hipLaunchKernelGGL(k_genw_enter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                   states + (size_t)t * B * 2 * g.H, B, g.H, g.KBh, a_prev, a_t, c.hfrag[1 - p], c.cfrag);
            }
            launch_gemm(0, *w, g, c, rowsP, a_t, 1 - p, 0, st);
            launch_draw(*w, g, c, rowsP, map, a_t, t, 1 - p, uniforms, forced, completions, st);
            p = 1 - p;
        }
        GW_CHECK_LAUNCH("k_genw rollout step");
    }
    GW_CHECK_LAUNCH("k_genw rollout");
    return TG_OK;
}

size_t tg_genw_nll_workspace_bytes(const TgGenWeights* w, int N) { return tg_genw_sample_workspace_bytes(w, N); }

// Teacher-forced pass for NLL.cal_nll (metrics/nll.py:49-63): zero state, inp[:, t] fed at step t, log-probability of
// target[:, t].  Same GEMMs as the sampling paths, k_genw_lse in place of the draw.
int tg_genw_nll(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L, float* logp_out,
                float* lse_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const Geom g = make_geom(*w);
    int rc = check_shape(*w, g);
    if (rc) return rc;
    const size_t rows_c = (size_t)N < (size_t)kChunkRows ? (size_t)N : (size_t)kChunkRows;
    const size_t rowsP = pad128(rows_c);
    const Carve c = carve(g, rowsP, workspace);
    if (!workspace || workspace_bytes < c.bytes) {
        tg_set_error("wide generator path: workspace too small (%zu < %zu)", workspace_bytes, c.bytes);
        return TG_ERR_WORKSPACE;
    }
    if ((rc = ensure_lds())) return rc;
    if ((rc = prep_all(*w, g, c, rowsP, st))) return rc;
    for (size_t n0 = 0; n0 < (size_t)N; n0 += rows_c) {
        const int rows = (int)(((size_t)N - n0) < rows_c ? ((size_t)N - n0) : rows_c);
        hipLaunchKernelGGL(k_genw_fill_tok, dim3((unsigned)((rowsP + 255) / 256)), dim3(256), 0, st, c.tok, rowsP, 0);
        const size_t n = (size_t)rows * (g.H / 4);
        hipLaunchKernelGGL(k_genw_enter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)nullptr, 1, g.H,
                           g.KBh, 0, rows, c.hfrag[0], c.cfrag);
        int p = 0;
        for (int t = 0; t < L; ++t) {
            hipLaunchKernelGGL(k_genw_tok_col, dim3((rows + 255) / 256), dim3(256), 0, st, inp + n0 * L, rows, L, t, g.V,
                               c.tok);
            launch_gemm(1, *w, g, c, rowsP, rows, p, 1 - p, st);
            launch_gemm(0, *w, g, c, rowsP, rows, 1 - p, 0, st);
            LseArgs a = {};
            a.g = g;
            a.rec = c.rec;
            a.ntiles = g.MBp * 2;
            a.rows = rows;
            a.t = t;
            a.L = L;
            a.target = target + n0 * L;
            a.logp = logp_out + n0 * L;
            a.lse = lse_out ? lse_out + n0 * L : nullptr;
            a.hfrag = c.hfrag[1 - p];
            a.w_out = w->w_out;
            a.b_out = w->b_out;
            tg_prof_begin(TG_KERNEL_GENW_DRAW, st);
            hipLaunchKernelGGL(k_genw_lse, dim3((rows + kDrawRows - 1) / kDrawRows), dim3(512), 0, st, a);
            tg_prof_end(st);
            p = 1 - p;
        }
        GW_CHECK_LAUNCH("k_genw nll step");
    }
    return TG_OK;
}

int tg_genw_sample(const TgGenWeights* w, int N, int L, int64_t start_letter, const float* uniforms,
                   const int64_t* forced, int64_t* out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const Geom g = make_geom(*w);
    int rc = check_shape(*w, g);
    if (rc) return rc;
    const size_t rows_c = (size_t)N < (size_t)kChunkRows ? (size_t)N : (size_t)kChunkRows;
    const size_t rowsP = pad128(rows_c);
    const Carve c = carve(g, rowsP, workspace);
    if (!workspace || workspace_bytes < c.bytes) {
        tg_set_error("wide generator path: workspace too small (%zu < %zu)", workspace_bytes, c.bytes);
        return TG_ERR_WORKSPACE;
    }
    if ((rc = ensure_lds())) return rc;
    if ((rc = prep_all(*w, g, c, rowsP, st))) return rc;
    for (size_t n0 = 0; n0 < (size_t)N; n0 += rows_c) {
        const int rows = (int)(((size_t)N - n0) < rows_c ? ((size_t)N - n0) : rows_c);
        RowMap map = {};
        map.rollout = 0;
        map.L = L;
        map.N = N;
        map.n0 = (int)n0;
        hipLaunchKernelGGL(k_genw_fill_tok, dim3((unsigned)((rowsP + 255) / 256)), dim3(256), 0, st, c.tok, rowsP,
                           (int)start_letter);
        {   // zero state (models/generator.py:102-109)
            const size_t n = (size_t)rows * (g.H / 4);
            hipLaunchKernelGGL(k_genw_enter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)nullptr, 1,
                               g.H, g.KBh, 0, rows, c.hfrag[0], c.cfrag);
        }
        int p = 0;
        for (int t = 0; t < L; ++t) {
            launch_gemm(1, *w, g, c, rowsP, rows, p, 1 - p, st);
            launch_gemm(0, *w, g, c, rowsP, rows, 1 - p, 0, st);
            launch_draw(*w, g, c, rowsP, map, rows, t, 1 - p, uniforms, forced, out, st);
            p = 1 - p;
        }
        GW_CHECK_LAUNCH("k_genw sample step");
    }
    return TG_OK;
}
