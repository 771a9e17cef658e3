// LeakGAN Monte-Carlo rollout: ROLLOUT.rollout_mc_search_leakgan for every (rollout index, given_num)
// pair of get_reward_leakgan (utils/rollout.py:57-126, 172-175) driving LeakGAN_G.forward (models/LeakGAN_G.py:52-102)
// and the discriminator's get_feature (models/discriminator.py:51-64) -- gfx950 only.
//
// All rows (rollout index i, given_num index t, sentence b) advance one position together; per position:
//   k_leak_draw   token of the position: the sentence's (pos < given_num) or the canonical inverse-CDF draw from the
//                 previous position's outputs; written to the completions and to the zero-padded feature input
//   tg_dis_features  the scorer's get_feature over the rows' tokens so far (tg_dis.hip: conv + max-pool + highway on
//                 the matrix cores, weights prepared once per call)
//   k_leak_cells  worker LSTM cell (token embedding), the projection operand z = temperature * [rg (x) h_w | rg],
//                 manager LSTM cell (feature), the goal bookkeeping
//   k_leak_proj   outputs[n, v] = z[n, :] . [work2goal.weight[v*K..v*K+K, :] | work2goal.bias[v*K..]]  (fp32 MFMA)
//
// Goal algebra.  The reference keeps cur_goal = normalize(mana2goal(h_m)) (G = 1720 values per row and position)
// and uses it only through rg = normalize(goal2goal(real_goal)), real_goal = a sum of four cur_goals (+ goal_init).
// goal2goal is linear and bias-free, so  goal2goal(cur_goal) = (P h_m + q) / ||W h_m + b||  with P = W_g2g W_m2g
// [K,H], q = W_g2g b_m2g and  ||W h_m + b||^2 = h_m^T Q h_m + 2 c^T h_m + s  (Q = W^T W, c = W^T b, s = b^T b): per
// row and position K + 1 numbers instead of G.  P, q, Q, c, s and goal2goal(goal_init[b]) are computed once per call
// and evaluated in fp64 (no cancellation worry, more accurate than the reference's fp32 sums).
#include <mutex>

#include "tg_common.h"
#include "tg_dis.h"

namespace {

constexpr int kLRows = 32;      // rows per workgroup of k_leak_cells
constexpr int kLThreads = 256;
constexpr int kMaxT = 64;       // given_num values per call
constexpr int kNSpan = 8;       // canonical sampler: spans per row
constexpr int kMaxHLeak = 128;

struct LeakArgs {
    TgLeakWeights w;
    const int64_t* sentences;  // [B,L]
    const float* uniforms;     // [R,T,L,B]
    const int64_t* forced;     // [R,T,B,L] or null
    int64_t* out;              // [R,T,B,L]
    int64_t* fed;              // [N,L] tokens fed so far, zero padded (the feature extractor's input)
    int64_t* tok;              // [N]   token fed at this position
    float* feat;               // [N,G]
    float *hw, *cw, *hm, *cm;  // [N,H]
    float* rg;                 // [N,K] normalised goal2goal(real_goal) in force
    double* ring;              // [4][N][K] goal2goal(cur_goal) of the last four positions
    float* z;                  // [N, K*H + K] projection operand
    float* mpre;               // [N, 4H] manager.weight_ih . feature of this position (k_leak_mgates)
    float* logits;             // [N,V]
    const double* prep;        // P [K*H], q [K], Q [H*H], c [H], s [1], gp [B*K]
    int given[kMaxT];
    int B, L, T, i_lo, N, n0;  // N rows of this chunk, first row n0 = (i_chunk - i_lo) * T * B ... see row_ids
    int i0;                    // first rollout index of the chunk
    int pos;
    int Ls, GK, NG;            // canonical sampler geometry
    long long* stamps;         // diagnostic (TG_LEAK_STAMPS, -DTG_DIAG=1): k_leak_cells phase cycles, [workgroup][8]
};

__device__ __forceinline__ void row_ids(const LeakArgs& a, int n, int& i, int& t, int& b) {
    b = n % a.B;
    t = (n / a.B) % a.T;
    i = a.i0 + n / (a.B * a.T);
}

// ---- once per call: the goal algebra's constants, fp64
__global__ __launch_bounds__(64) void k_leak_prep(TgLeakWeights w, int B, double* prep) {
    // one wave per output: lanes stride over the G terms, fixed shuffle tree (the result does not depend on the grid)
    const int K = w.K, H = w.H, G = w.G;
    const int nP = K * H, nq = K, nQ = H * H, nc = H, ns = 1;
    const int idx = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    int j = idx;
    if (j < nP) {
        const int k = j / H, h = j % H;
        for (int o = lane; o < G; o += 64) acc += (double)w.g2g_w[(size_t)k * G + o] * (double)w.m2g_w[(size_t)o * H + h];
    } else if ((j -= nP) < nq) {
        for (int o = lane; o < G; o += 64) acc += (double)w.g2g_w[(size_t)j * G + o] * (double)w.m2g_b[o];
    } else if ((j -= nq) < nQ) {
        const int h1 = j / H, h2 = j % H;
        for (int o = lane; o < G; o += 64) acc += (double)w.m2g_w[(size_t)o * H + h1] * (double)w.m2g_w[(size_t)o * H + h2];
    } else if ((j -= nQ) < nc) {
        for (int o = lane; o < G; o += 64) acc += (double)w.m2g_w[(size_t)o * H + j] * (double)w.m2g_b[o];
    } else if ((j -= nc) < ns) {
        for (int o = lane; o < G; o += 64) acc += (double)w.m2g_b[o] * (double)w.m2g_b[o];
    } else {
        j -= ns;
        const int b = j / K, k = j % K;
        for (int o = lane; o < G; o += 64) acc += (double)w.g2g_w[(size_t)k * G + o] * (double)w.goal_init[(size_t)b * G + o];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) prep[idx] = acc;
}

// ---- per chunk: zero state, zero-padded feature input, rg = normalize(goal2goal(goal_init[b]))
__global__ __launch_bounds__(256) void k_leak_init(LeakArgs a) {
    const int n = blockIdx.x;
    const int H = a.w.H, K = a.w.K;
    int i, t, b;
    row_ids(a, n, i, t, b);
    for (int h = threadIdx.x; h < H; h += 256) {
        a.hw[(size_t)n * H + h] = 0.f; a.cw[(size_t)n * H + h] = 0.f;
        a.hm[(size_t)n * H + h] = 0.f; a.cm[(size_t)n * H + h] = 0.f;
    }
    for (int p = threadIdx.x; p < a.L; p += 256) a.fed[(size_t)n * a.L + p] = 0;
    for (int q = threadIdx.x; q < 4 * K; q += 256) a.ring[((size_t)(q / K) * a.N + n) * K + q % K] = 0.0;
    if (threadIdx.x == 0) {
        const double* gp = a.prep + K * H + K + H * H + H + 1 + (size_t)b * K;
        double ss = 0.0;
        for (int k = 0; k < K; ++k) ss += gp[k] * gp[k];
        double nrm = sqrt(ss);
        nrm = nrm < 1e-12 ? 1e-12 : nrm;
        for (int k = 0; k < K; ++k) a.rg[(size_t)n * K + k] = (float)(gp[k] / nrm);
    }
}

// ---- token of position a.pos for every row (one workgroup per row)
// pos < given_num: the sentence's token.  Otherwise the canonical inverse-CDF draw (DESIGN.md section 4;
// oracle/textgan_oracle.c canonical_draw) from w[v] = exp(logits[n,v] - max): tile sums and group sums fp32
// sequential, span sums / total / target / every prefix comparison fp64.
__global__ __launch_bounds__(256) void k_leak_draw(LeakArgs a) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int V = a.w.V, L = a.L, pos = a.pos;
    int i, t, b;
    row_ids(a, n, i, t, b);
    const size_t orow = (((size_t)i * a.T + t) * a.B + b) * L;
    if (pos < a.given[t]) {
        if (tid == 0) {
            const int64_t tk = a.sentences[(size_t)b * L + pos];
            a.out[orow + pos] = tk;
            a.fed[(size_t)n * L + pos] = tk;
            a.tok[n] = tk;
        }
        return;
    }
    extern __shared__ float lds[];
    const int Ls = a.Ls, GK = a.GK, NG = a.NG, NT = Ls / 16;
    float* ts = lds;                      // [8*NT] tile sums
    float* gs = ts + kNSpan * NT;         // [8*NG] group sums
    double* ss = reinterpret_cast<double*>(gs + kNSpan * NG + ((kNSpan * (NT + NG)) & 1));  // [8] span sums
    float* red = reinterpret_cast<float*>(ss + kNSpan);                                    // [256]
    const float* lg = a.logits + (size_t)n * V;
    float m = TG_NEG_INF;
    for (int v = tid; v < V; v += 256) m = __builtin_fmaxf(m, lg[v]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = __builtin_fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    m = red[0];
    // weight of entry v: exp(logit - max), 0 past the vocabulary; evaluated where it is summed (the row's weights are
    // not kept: 8*Ls floats of LDS were what limited V to 32 768) -- the same expression at every use, so the same bits
    auto wgt = [&](int v) { return v < V ? tg_expf(lg[v] - m) : 0.f; };
    for (int q = tid; q < kNSpan * NT; q += 256) {   // tile q of the row: entries [16q, 16q+16)
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += wgt(q * 16 + r);
        ts[q] = s;
    }
    __syncthreads();
    for (int q = tid; q < kNSpan * NG; q += 256) {   // group (sp, g): tiles g*GK .. of span sp
        const int sp = q / NG, g = q % NG;
        float c = 0.f;
        for (int k = 0; k < GK; ++k) {
            const int tl = g * GK + k;
            if (tl >= NT) break;
            c += ts[sp * NT + tl];
        }
        gs[q] = c;
    }
    __syncthreads();
    if (tid < kNSpan) {
        double s = 0.0;
        for (int g = 0; g < NG; ++g) s += (double)gs[tid * NG + g];
        ss[tid] = s;
    }
    __syncthreads();
    if (tid != 0) return;
    int last_nonempty = 0;
    double S = 0.0;
    for (int sp = 0; sp < kNSpan; ++sp) {
        S += ss[sp];
        if (sp * Ls < V) last_nonempty = sp;
    }
    const float u = a.uniforms[(((size_t)i * a.T + t) * L + pos) * a.B + b];
    const double target = (double)u * S;
    int sel = last_nonempty;
    double p = 0.0;
    for (int sp = 0; sp < kNSpan; ++sp) {
        if (p + ss[sp] > target || sp == last_nonempty) { sel = sp; break; }
        p += ss[sp];
    }
    const int gsz = GK * 16;
    const int span_lo = sel * Ls;
    const int span_hi = tg_min(span_lo + Ls, V);
    const int ng_valid = (span_hi - span_lo + gsz - 1) / gsz;
    int gq = ng_valid - 1;
    for (int g = 0; g < ng_valid; ++g) {
        const double c = (double)gs[sel * NG + g];
        if (p + c > target || g == ng_valid - 1) { gq = g; break; }
        p += c;
    }
    const int g_lo = span_lo + gq * gsz;
    const int g_hi = tg_min(g_lo + gsz, span_hi);
    const int nt_valid = (g_hi - g_lo + 15) / 16;
    int tq = nt_valid - 1;
    for (int k = 0; k < nt_valid; ++k) {
        const double tsum = (double)ts[(g_lo >> 4) + k];
        if (p + tsum > target || k == nt_valid - 1) { tq = k; break; }
        p += tsum;
    }
    const int t_lo = g_lo + tq * 16;
    const int t_hi = tg_min(t_lo + 16, g_hi);
    float pp = 0.f;
    int tk = t_hi - 1;
    for (int j = t_lo; j < t_hi; ++j) {
        pp += wgt(j);
        if (p + (double)pp > target) { tk = j; break; }
    }
    a.out[orow + pos] = tk;
    const int64_t fedtok = a.forced ? a.forced[orow + pos] : (int64_t)tk;   // replay mode feeds the given token on
    a.fed[(size_t)n * L + pos] = fedtok;
    a.tok[n] = fedtok;
}

// acc += A[i][:] . B[j][:] over K (A, B row-major with the contraction index contiguous): lane l feeds
// v_mfma_f32_32x32x2_f32 with four k's per 16-byte load -- k = 8q + 4*(l>>5) + e for e = 0..3 -- on both operands
// alike (the k order inside a product is free as long as A and B agree).  VEC: K % 8 == 0 and both rows 16-B aligned.
template <bool VEC>
__device__ __forceinline__ f32x16 dot_seg(f32x16 acc, const float* __restrict__ arow, const float* __restrict__ brow,
                                          int K, int kk) {
    if constexpr (VEC) {
        // blocks of U k-steps, the next block's 2U loads in flight under this block's 4U MFMAs: with one wave per SIMD
        // and 40 workgroups on the chip (1 280 rows) nothing else hides the L2 latency of the operand rows
        constexpr int U = 4;
        f32x4 a0[U], b0[U], a1[U], b1[U], a2[U], b2[U];
        const float* ap = arow + 4 * kk;
        const float* bp = brow + 4 * kk;
        auto load = [&](f32x4 (&av)[U], f32x4 (&bv)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + 8 * u < K) {
                    av[u] = *reinterpret_cast<const f32x4*>(ap + k0 + 8 * u);
                    bv[u] = *reinterpret_cast<const f32x4*>(bp + k0 + 8 * u);
                }
        };
        auto mfma = [&](const f32x4 (&av)[U], const f32x4 (&bv)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + 8 * u < K) {
                    acc = TG_MFMA32(av[u][0], bv[u][0], acc);
                    acc = TG_MFMA32(av[u][1], bv[u][1], acc);
                    acc = TG_MFMA32(av[u][2], bv[u][2], acc);
                    acc = TG_MFMA32(av[u][3], bv[u][3], acc);
                }
        };
        // three register sets: the loads of the block after next are issued before this block's MFMAs
        constexpr int BK = 8 * U;
        load(a0, b0, 0);
        load(a1, b1, BK);
        for (int k = 0; k < K; k += 3 * BK) {
            load(a2, b2, k + 2 * BK);
            mfma(a0, b0, k);
            if (k + BK < K) {
                load(a0, b0, k + 3 * BK);
                mfma(a1, b1, k + BK);
            }
            if (k + 2 * BK < K) {
                load(a1, b1, k + 4 * BK);
                mfma(a2, b2, k + 2 * BK);
            }
        }
    } else {
        for (int k = 0; k < K; k += 2) {
            const bool ok = k + kk < K;
            const float av = ok ? arow[k + kk] : 0.f;
            const float bv = ok ? brow[k + kk] : 0.f;
            acc = TG_MFMA32(av, bv, acc);
        }
    }
    return acc;
}

__device__ __forceinline__ bool vec_ok(const void* p, int ld, int K) {
    return (K % 8 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
}

// gates [4H][32 rows] = W_ih . x + W_hh . h for the workgroup's 32 rows -> LDS g[4H][33]; wave w takes gate tiles
// w, w+4, ...  x rows: xrow(j) (K = nin, row-major), h rows: hprev + n*H.
// xbase / hbase: the arrays the rows come from (their alignment decides the load width, uniformly for the wave).
// pre != nullptr: W_ih . x was computed by k_leak_mgates (pre[n][4H], row of this lane = prow): only the recurrent
// product runs here.
__device__ void lstm_gates(const float* __restrict__ w_ih, int nin, const float* __restrict__ w_hh, int H,
                           const float* __restrict__ xrow, const float* xbase, const float* __restrict__ hrow,
                           const float* hbase, float* g, int wave, int lane, const float* __restrict__ prow = nullptr) {
    const int i = lane & 31, kk = lane >> 5;
    const bool v1 = vec_ok(w_ih, nin, nin) && vec_ok(xbase, nin, nin);
    const bool v2 = vec_ok(w_hh, H, H) && vec_ok(hbase, H, H);
    for (int tile = wave; tile < (4 * H) / 32; tile += 4) {
        f32x16 acc = {};
        const float* a1 = w_ih + (size_t)(tile * 32 + i) * nin;
        const float* a2 = w_hh + (size_t)(tile * 32 + i) * H;
        if (!prow) acc = v1 ? dot_seg<true>(acc, a1, xrow, nin, kk) : dot_seg<false>(acc, a1, xrow, nin, kk);
        acc = v2 ? dot_seg<true>(acc, a2, hrow, H, kk) : dot_seg<false>(acc, a2, hrow, H, kk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gi = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            g[gi * 33 + i] = prow ? prow[gi] + acc[r] : acc[r];
        }
    }
}

// ---- manager.weight_ih . feature for every row of the position: pre[n][4H].  One workgroup per (32 rows, 32 gate
// rows); its four waves take a quarter of the K = G contraction each (the product is a chain of K/2 dependent MFMAs:
// inside k_leak_cells, 40 workgroups of 4 waves, it was half of that kernel's time) and are summed through LDS in
// segment order.
__global__ __launch_bounds__(256) void k_leak_mgates(LeakArgs a) {
    __shared__ float red[4][32][33];
    const int H = a.w.H, G = a.w.G, N = a.N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, kk = lane >> 5;
    const int n0 = blockIdx.x * 32, tile = blockIdx.y;
    const int nj = tg_min(n0 + i, N - 1);
    const float* arow = a.w.m_wih + (size_t)(tile * 32 + i) * G;
    const float* brow = a.feat + (size_t)nj * G;
    const bool vec = vec_ok(a.w.m_wih, G, G) && vec_ok(a.feat, G, G);
    f32x16 acc = {};
    if (vec) {
        const int seg_len = ((G / 8 + 3) / 4) * 8;
        const int k_lo = tg_min(wave * seg_len, G), k_hi = tg_min(k_lo + seg_len, G);
        acc = dot_seg<true>(acc, arow + k_lo, brow + k_lo, k_hi - k_lo, kk);
    } else if (wave == 0) {
        acc = dot_seg<false>(acc, arow, brow, G, kk);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kk][i] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int j = idx >> 5, gi = idx & 31;          // consecutive threads: consecutive gate rows of one row
        if (n0 + j < N)
            a.mpre[(size_t)(n0 + j) * 4 * H + tile * 32 + gi] = ((red[0][gi][j] + red[1][gi][j]) + red[2][gi][j]) + red[3][gi][j];
    }
}

// nn.LSTM cell from pre-activations g[4H][33] (+ b_ih + b_hh), gate order i,f,g,o: updates h, c (global, [N,H]) and
// keeps the new h of the 32 rows in LDS hl[32][H+1].
__device__ void lstm_update(const float* g, const float* __restrict__ b_ih, const float* __restrict__ b_hh, int H,
                            float* h, float* c, float* hl, int n0, int N, int tid) {
    for (int idx = tid; idx < 32 * H; idx += kLThreads) {
        const int j = idx / H, uu = idx % H;   // consecutive threads: consecutive units of one row (coalesced h / c)
        const int n = n0 + j;
        if (n >= N) { hl[j * (H + 1) + uu] = 0.f; continue; }
        const float ig = tg_sigmoid(g[uu * 33 + j] + (b_ih[uu] + b_hh[uu]));
        const float fg = tg_sigmoid(g[(H + uu) * 33 + j] + (b_ih[H + uu] + b_hh[H + uu]));
        const float gg = tg_tanh(g[(2 * H + uu) * 33 + j] + (b_ih[2 * H + uu] + b_hh[2 * H + uu]));
        const float og = tg_sigmoid(g[(3 * H + uu) * 33 + j] + (b_ih[3 * H + uu] + b_hh[3 * H + uu]));
        const float cn = fg * c[(size_t)n * H + uu] + ig * gg;
        const float hn = og * tg_tanh(cn);
        c[(size_t)n * H + uu] = cn;
        h[(size_t)n * H + uu] = hn;
        hl[j * (H + 1) + uu] = hn;
    }
}

// ---- worker cell, projection operand, manager cell, goal bookkeeping for 32 rows (LeakGAN_G.py:69-84)
__global__ __launch_bounds__(kLThreads) void k_leak_cells(LeakArgs a) {
    extern __shared__ float lds[];
    const int H = a.w.H, K = a.w.K, E = a.w.E, G = a.w.G, N = a.N;
    float* g = lds;                     // [4H][33]
    float* hl = g + 4 * H * 33;         // [32][H+1]
    double* dred = reinterpret_cast<double*>(hl + 32 * (H + 1) + ((32 * (H + 1)) & 1));  // [32][8] partial norms
    double* pk = dred + 32 * 8;         // [32][K] goal2goal(cur_goal) of this position
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blockIdx.x * kLRows;
    const int j = lane & 31;
    const int nj = tg_min(n0 + j, N - 1);   // loads of rows past the end read the last row; stores are guarded

    const long long c0 = a.stamps ? tg_clock() : 0;
    // worker LSTM on the embedding of the token fed at this position
    int64_t tk = a.tok[nj];
    tk = tk < 0 ? 0 : (tk >= a.w.V ? a.w.V - 1 : tk);
    lstm_gates(a.w.w_wih, E, a.w.w_whh, H, a.w.emb + (size_t)tk * E, a.w.emb, a.hw + (size_t)nj * H, a.hw, g, wave, lane);
    __syncthreads();
    lstm_update(g, a.w.w_bih, a.w.w_bhh, H, a.hw, a.cw, hl, n0, N, tid);
    __syncthreads();
    const long long c1 = a.stamps ? tg_clock() : 0;
    // z = temperature * [rg (x) h_w | rg], with the real goal in force BEFORE this position's update; temperature
    // 1.5 for idx <= 1, else 1.0 (train=True; LeakGAN_G.py:87-95)
    const float temp = a.pos > 1 ? 1.0f : a.w.temperature;
    const int ldz = K * H + K;
    // thread = column q of the operand (its goal index and hidden unit fixed once), loop over the 32 rows: no
    // division inside the loop, the stores of a wave are 256 contiguous bytes of one row
    const int rows = tg_min(32, N - n0);
    for (int idx = tid; idx < rows * K; idx += kLThreads) g[idx] = a.rg[(size_t)n0 * K + idx];   // g is free here (<= 32 x 64)
    __syncthreads();
    for (int q = tid; q < ldz; q += kLThreads) {
        const bool tail = q >= K * H;
        const int kq = tail ? q - K * H : q / H, hq = tail ? 0 : q % H;
        for (int r = 0; r < rows; ++r) {
            const float rgk = g[r * K + kq];
            a.z[(size_t)(n0 + r) * ldz + q] = tail ? temp * rgk : temp * rgk * hl[r * (H + 1) + hq];
        }
    }
    __syncthreads();

    const long long c2 = a.stamps ? tg_clock() : 0;
    // manager LSTM on the feature of the tokens so far
    lstm_gates(a.w.m_wih, G, a.w.m_whh, H, a.feat + (size_t)nj * G, a.feat, a.hm + (size_t)nj * H, a.hm, g, wave, lane,
               a.mpre + (size_t)nj * 4 * H);
    __syncthreads();
    lstm_update(g, a.w.m_bih, a.w.m_bhh, H, a.hm, a.cm, hl, n0, N, tid);
    __syncthreads();

    const long long c3 = a.stamps ? tg_clock() : 0;
    // goal2goal(cur_goal) = (P h_m + q) / max(||W h_m + b||, 1e-12), fp64; 8 threads per row
    const double* P = a.prep;
    const double* q = P + K * H;
    const double* Q = q + K;
    const double* c = Q + H * H;
    const double s = c[H];
    const int r = tid >> 3, part = tid & 7;
    const float* hrow = hl + r * (H + 1);
    double acc = 0.0;
    for (int h1 = part; h1 < H; h1 += 8) {
        double row = 2.0 * c[h1];
        for (int h2 = 0; h2 < H; ++h2) row += Q[h1 * H + h2] * (double)hrow[h2];
        acc += row * (double)hrow[h1];
    }
    dred[r * 8 + part] = acc;
    for (int k = part; k < K; k += 8) {
        double v = q[k];
        for (int h = 0; h < H; ++h) v += P[k * H + h] * (double)hrow[h];
        pk[r * K + k] = v;
    }
    __syncthreads();
    const int n = n0 + r;
    if (part == 0 && n < N) {
        double n2 = s;
        for (int e = 0; e < 8; ++e) n2 += dred[r * 8 + e];
        double nrm = sqrt(n2 > 0.0 ? n2 : 0.0);
        nrm = nrm < 1e-12 ? 1e-12 : nrm;
        const int pos = a.pos, step = a.w.step_size;
        double* slot = a.ring + ((size_t)(pos & 3) * N + n) * K;
        for (int k = 0; k < K; ++k) slot[k] = pk[r * K + k] / nrm;
        // goal_array[pos] = cur_goal; if pos > 0 and pos % step_size == 0: real_goal = sum(goal_array[pos-3 .. pos])
        // (an EMPTY slice when pos < 3), + goal_init[b] when pos == step_size (utils/rollout.py:89-93, 117-121)
        if (pos > 0 && pos % step == 0) {
            int i, t, b;
            row_ids(a, n, i, t, b);
            const double* gp = c + H + 1 + (size_t)b * K;
            double ss = 0.0;
            for (int k = 0; k < K; ++k) {
                double v = 0.0;
                if (pos >= 3)
                    for (int jj = pos - 3; jj <= pos; ++jj) v += a.ring[((size_t)(jj & 3) * N + n) * K + k];
                if (pos == step) v += gp[k];
                pk[r * K + k] = v;
                ss += v * v;
            }
            double rn = sqrt(ss);
            rn = rn < 1e-12 ? 1e-12 : rn;
            for (int k = 0; k < K; ++k) a.rg[(size_t)n * K + k] = (float)(pk[r * K + k] / rn);
        }
    }
    if (a.stamps && tid == 0) {
        const long long c4 = tg_clock();
        long long* o = a.stamps + (size_t)blockIdx.x * 8;
        o[0] += c1 - c0; o[1] += c2 - c1; o[2] += c3 - c2; o[3] += c4 - c3; o[4] += 1;
    }
}

// ---- outputs[n, v] = z[n, :] . [work2goal.weight rows v*K .. v*K+K-1 | work2goal.bias[v*K ..]]
// (work2goal(h_w).view(V, K) @ rg, LeakGAN_G.py:80-84, times the temperature folded into z).  Workgroup tile 128
// vocabulary entries x 128 rows, 2 x 2 waves of 64 x 64 on v_mfma_f32_32x32x2_f32; both operands read from L2 with
// 16-byte loads.  Row tiles none of whose rows draws its next token are skipped.
__global__ __launch_bounds__(256) void k_leak_proj(LeakArgs a) {
    const int V = a.w.V, K = a.w.K, H = a.w.H, N = a.N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int v0 = blockIdx.x * 128 + (wave & 1) * 64;
    const int r0 = blockIdx.y * 128 + (wave >> 1) * 64;
    // any row of the workgroup's 128 whose next position is sampled?
    int need = 0;
    if (tid < 128) {
        const int n = blockIdx.y * 128 + tid;
        if (n < N) need = a.pos + 1 >= a.given[(n / a.B) % a.T];
    }
    if (!__syncthreads_or(need)) return;
    const int i = lane & 31, kk = lane >> 5;
    const int KH = K * H, ldz = KH + K;
    const float* arow[2];
    const float* brow[2];
    const float* abias[2];
    for (int q = 0; q < 2; ++q) {
        const int v = tg_min(v0 + 32 * q + i, V - 1);
        const int n = tg_min(r0 + 32 * q + i, N - 1);
        arow[q] = a.w.w2g_w + (size_t)v * KH;
        abias[q] = a.w.w2g_b + (size_t)v * K;
        brow[q] = a.z + (size_t)n * ldz;
    }
    f32x16 acc[2][2] = {};
    const bool vec = (KH % 8 == 0) && (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.w.w2g_w) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.w.w2g_b) & 15) == 0);
    if (vec) {
        // the next k-step's four operand loads in flight under this step's 16 MFMAs
        f32x4 av[2], bv[2], an[2], bn[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            av[q] = *reinterpret_cast<const f32x4*>(arow[q] + 4 * kk);
            bv[q] = *reinterpret_cast<const f32x4*>(brow[q] + 4 * kk);
        }
        for (int k = 0; k < KH; k += 8) {
            const int kn = k + 8 < KH ? k + 8 : k;   // the last step re-reads its own operands (never used)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                an[q] = *reinterpret_cast<const f32x4*>(arow[q] + kn + 4 * kk);
                bn[q] = *reinterpret_cast<const f32x4*>(brow[q] + kn + 4 * kk);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int qa = 0; qa < 2; ++qa)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) acc[qa][qb] = TG_MFMA32(av[qa][e], bv[qb][e], acc[qa][qb]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                av[q] = an[q];
                bv[q] = bn[q];
            }
        }
        for (int k = 0; k < K; k += 8) {
            f32x4 av[2], bv[2];
            for (int q = 0; q < 2; ++q) {
                av[q] = *reinterpret_cast<const f32x4*>(abias[q] + k + 4 * kk);
                bv[q] = *reinterpret_cast<const f32x4*>(brow[q] + KH + k + 4 * kk);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int qa = 0; qa < 2; ++qa)
                    for (int qb = 0; qb < 2; ++qb) acc[qa][qb] = TG_MFMA32(av[qa][e], bv[qb][e], acc[qa][qb]);
        }
    } else {
        for (int qa = 0; qa < 2; ++qa)
            for (int qb = 0; qb < 2; ++qb) {
                acc[qa][qb] = dot_seg<false>(acc[qa][qb], arow[qa], brow[qb], KH, kk);
                acc[qa][qb] = dot_seg<false>(acc[qa][qb], abias[qa], brow[qb] + KH, K, kk);
            }
    }
    // D lane l, reg r -> vocabulary v0 + 32qa + (r&3) + 8(r>>2) + 4kk, row r0 + 32qb + (l&31)
    const bool st4 = (V % 4) == 0;
    for (int qa = 0; qa < 2; ++qa)
        for (int qb = 0; qb < 2; ++qb) {
            const int n = r0 + 32 * qb + i;
            if (n >= N) continue;
            float* orow = a.logits + (size_t)n * V;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int v = v0 + 32 * qa + 8 * rq + 4 * kk;
                if (st4 && v + 3 < V) {
                    const f32x4 o = {acc[qa][qb][4 * rq], acc[qa][qb][4 * rq + 1], acc[qa][qb][4 * rq + 2], acc[qa][qb][4 * rq + 3]};
                    *reinterpret_cast<f32x4*>(orow + v) = o;
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (v + e < V) orow[v + e] = acc[qa][qb][4 * rq + e];
                }
            }
        }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct LeakLayout {
    size_t fed, tok, feat, hw, cw, hm, cm, rg, ring, z, mpre, logits, prep, dis, total;
    int rows;  // rows of one chunk
};

int span_len(int V) { return 16 * ((V + kNSpan * 16 - 1) / (kNSpan * 16)); }

// rows per chunk: whole rollout indices (T*B rows each), logits of a chunk at most ~1 GiB
int chunk_indices(const TgLeakWeights* g, int B, int T, int n_i) {
    const size_t per_i = (size_t)T * B;
    size_t fit = ((size_t)1 << 30) / (per_i * g->V * sizeof(float));
    if (fit < 1) fit = 1;
    return (int)(fit < (size_t)n_i ? fit : (size_t)n_i);
}

LeakLayout layout(const TgLeakWeights* g, const TgDisWeights* d, int B, int L, int T, int n_i) {
    LeakLayout o = {};
    const int ci = chunk_indices(g, B, T, n_i);
    const size_t N = (size_t)ci * T * B;
    const int H = g->H, K = g->K;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += al256(bytes); return at; };
    o.rows = (int)N;
    o.fed = take(N * L * sizeof(int64_t));
    o.tok = take(N * sizeof(int64_t));
    o.feat = take(N * g->G * sizeof(float));
    o.hw = take(N * H * sizeof(float));
    o.cw = take(N * H * sizeof(float));
    o.hm = take(N * H * sizeof(float));
    o.cm = take(N * H * sizeof(float));
    o.rg = take(N * K * sizeof(float));
    o.ring = take(4 * N * K * sizeof(double));
    o.z = take(N * ((size_t)K * H + K) * sizeof(float));
    o.mpre = take(N * (size_t)4 * H * sizeof(float));
    o.logits = take(N * (size_t)g->V * sizeof(float));
    o.prep = take(((size_t)K * H + K + (size_t)H * H + H + 1 + (size_t)B * K) * sizeof(double));
    o.dis = take(tg_dis_workspace_bytes(d, (int)N, L));
    o.total = off;
    return o;
}

int check_leak(const TgLeakWeights* g, const TgDisWeights* d, int B, int L, int T, const int32_t* given) {
    if (!g || !d || !g->emb || !g->w_wih || !g->w_whh || !g->w_bih || !g->w_bhh || !g->m_wih || !g->m_whh || !g->m_bih ||
        !g->m_bhh || !g->w2g_w || !g->w2g_b || !g->m2g_w || !g->m2g_b || !g->g2g_w || !g->goal_init) {
        tg_set_error("tg_leak_rollout: null weight pointer");
        return TG_ERR_ARG;
    }
    {
        const void* ptrs[] = {g->emb, g->w_wih, g->w_whh, g->w_bih, g->w_bhh, g->m_wih, g->m_whh, g->m_bih, g->m_bhh,
                              g->w2g_w, g->w2g_b, g->m2g_w, g->m2g_b, g->g2g_w, g->goal_init};
        uintptr_t al = 0;
        for (const void* q : ptrs) al |= reinterpret_cast<uintptr_t>(q);
        if (al & 15) {   // include/textgan_hip.h, "Alignment"
            tg_set_error("tg_leak_rollout: every weight tensor must be 16-byte aligned");
            return TG_ERR_ARG;
        }
    }
    if (g->H % 32 || g->H < 32 || g->H > kMaxHLeak || g->E < 1 || g->K < 1 || g->K > 64 || g->V < 1 || g->V > 65536 ||
        g->step_size < 1) {
        tg_set_error("tg_leak_rollout: shape V=%d E=%d H=%d K=%d step=%d not supported (H a multiple of 32 up to %d, "
                     "goal_size <= 64, V <= 65536)", g->V, g->E, g->H, g->K, g->step_size, kMaxHLeak);
        return TG_ERR_ARG;
    }
    if (g->G != d->F) {
        tg_set_error("tg_leak_rollout: goal_out_size %d != discriminator feature width %d", g->G, d->F);
        return TG_ERR_ARG;
    }
    if (B < 1 || B > g->B_init) {
        tg_set_error("tg_leak_rollout: batch %d exceeds goal_init's %d rows", B, g->B_init);
        return TG_ERR_ARG;
    }
    if (T < 1 || T > kMaxT || !given) {
        tg_set_error("tg_leak_rollout: between 1 and %d given_num values per call (got %d)", kMaxT, T);
        return TG_ERR_ARG;
    }
    for (int t = 0; t < T; ++t)
        if (given[t] < 1 || given[t] > L) {
            tg_set_error("tg_leak_rollout: given_num %d outside [1, %d]", given[t], L);
            return TG_ERR_ARG;
        }
    return TG_OK;
}

}  // namespace

extern "C" size_t tg_leak_rollout_workspace_bytes(const TgLeakWeights* g, const TgDisWeights* d, int B, int L, int T,
                                                  int n_i) {
    if (!g || !d || B < 1 || L < 1 || T < 1 || n_i < 1) return 0;
    return layout(g, d, B, L, T, n_i).total;
}

extern "C" int tg_leak_rollout(const TgLeakWeights* g, const TgDisWeights* d, const int64_t* sentences, int B, int L,
                               int R, const int32_t* given_nums, int T, int i_lo, int i_hi, const float* uniforms,
                               const int64_t* forced, int64_t* completions, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_leak(g, d, B, L, T, given_nums);
    if (rc) return rc;
    if (!sentences || !uniforms || !completions || L < 2 || R < 1 || i_lo < 0 || i_hi > R || i_lo > i_hi) {
        tg_set_error("tg_leak_rollout: bad arguments (B=%d L=%d R=%d i=[%d,%d))", B, L, R, i_lo, i_hi);
        return TG_ERR_ARG;
    }
    if (i_lo == i_hi) return TG_OK;
    const LeakLayout lo = layout(g, d, B, L, T, i_hi - i_lo);
    if (!workspace || workspace_bytes < lo.total) {
        tg_set_error("tg_leak_rollout: workspace too small (%zu < %zu)", workspace_bytes, lo.total);
        return TG_ERR_WORKSPACE;
    }
    char* ws = reinterpret_cast<char*>(workspace);
    LeakArgs a = {};
    a.w = *g;
    a.sentences = sentences;
    a.uniforms = uniforms;
    a.forced = forced;
    a.out = completions;
    a.fed = reinterpret_cast<int64_t*>(ws + lo.fed);
    a.tok = reinterpret_cast<int64_t*>(ws + lo.tok);
    a.feat = reinterpret_cast<float*>(ws + lo.feat);
    a.hw = reinterpret_cast<float*>(ws + lo.hw);
    a.cw = reinterpret_cast<float*>(ws + lo.cw);
    a.hm = reinterpret_cast<float*>(ws + lo.hm);
    a.cm = reinterpret_cast<float*>(ws + lo.cm);
    a.rg = reinterpret_cast<float*>(ws + lo.rg);
    a.ring = reinterpret_cast<double*>(ws + lo.ring);
    a.z = reinterpret_cast<float*>(ws + lo.z);
    a.mpre = reinterpret_cast<float*>(ws + lo.mpre);
    a.logits = reinterpret_cast<float*>(ws + lo.logits);
    a.prep = reinterpret_cast<const double*>(ws + lo.prep);
    for (int t = 0; t < T; ++t) a.given[t] = given_nums[t];
    a.B = B;
    a.L = L;
    a.T = T;
    a.i_lo = i_lo;
    a.Ls = span_len(g->V);
    const int NT = a.Ls / 16;
    a.GK = 1;
    a.stamps = tg_diag_pointer("TG_LEAK_STAMPS");
    while ((NT + a.GK - 1) / a.GK > 16 && a.GK < 8) a.GK *= 2;
    a.NG = (NT + a.GK - 1) / a.GK;
    const int H = g->H, K = g->K;

    const size_t sm_draw = sizeof(float) * ((size_t)kNSpan * NT + (size_t)kNSpan * a.NG + 2) +
                           sizeof(double) * kNSpan + sizeof(float) * 256;
    const size_t sm_cells = sizeof(float) * ((size_t)4 * H * 33 + (size_t)32 * (H + 1) + 2) +
                            sizeof(double) * ((size_t)32 * 8 + (size_t)32 * K);
    static TgLdsAttr attr_draw, attr_cells;
    if ((rc = tg_lds_attr(attr_draw, reinterpret_cast<const void*>(&k_leak_draw), sm_draw, "k_leak_draw"))) return rc;
    if ((rc = tg_lds_attr(attr_cells, reinterpret_cast<const void*>(&k_leak_cells), sm_cells, "k_leak_cells"))) return rc;
    if (sm_draw > 160 * 1024 || sm_cells > 160 * 1024) {
        tg_set_error("tg_leak_rollout: LDS need %zu / %zu bytes exceeds 160 KiB", sm_draw, sm_cells);
        return TG_ERR_ARG;
    }

    const int n_prep = K * H + K + H * H + H + 1 + B * K;
    hipLaunchKernelGGL(k_leak_prep, dim3(n_prep), dim3(64), 0, stream, *g, B,
                       reinterpret_cast<double*>(ws + lo.prep));
    const int ci = chunk_indices(g, B, T, i_hi - i_lo);
    int prepared_n = -1;     // row count the scorer's prepared weights in ws + lo.dis were laid out for
    for (int i0 = i_lo; i0 < i_hi; i0 += ci) {
        const int ni = (i_hi - i0 < ci) ? i_hi - i0 : ci;
        a.i0 = i0;
        a.N = ni * T * B;
        hipLaunchKernelGGL(k_leak_init, dim3(a.N), dim3(256), 0, stream, a);
        for (int pos = 0; pos < L; ++pos) {
            a.pos = pos;
            tg_prof_begin(TG_KERNEL_LEAK_DRAW, stream);
            hipLaunchKernelGGL(k_leak_draw, dim3(a.N), dim3(256), sm_draw, stream, a);
            tg_prof_end(stream);
            if (pos == L - 1) break;   // the reference's forward after the last token is never read
            rc = tg_dis_features(d, a.fed, a.N, L, a.feat, ws + lo.dis, tg_dis_workspace_bytes(d, a.N, L),
                                 prepared_n == a.N, stream);
            if (rc) return rc;
            prepared_n = a.N;
            tg_prof_begin(TG_KERNEL_LEAK_CELLS, stream);
            hipLaunchKernelGGL(k_leak_mgates, dim3((a.N + 31) / 32, (4 * g->H) / 32), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(k_leak_cells, dim3((a.N + kLRows - 1) / kLRows), dim3(kLThreads), sm_cells, stream, a);
            tg_prof_end(stream);
            tg_prof_begin(TG_KERNEL_LEAK_PROJ, stream);
            hipLaunchKernelGGL(k_leak_proj, dim3((g->V + 127) / 128, (a.N + 127) / 128), dim3(256), 0, stream, a);
            tg_prof_end(stream);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("tg_leak_rollout launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}
