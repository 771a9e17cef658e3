// LeakGAN Monte-Carlo rollout: ROLLOUT.rollout_mc_search_leakgan for every (rollout index, given_num)
// pair of get_reward_leakgan (utils/rollout.py:57-126, 172-175) driving LeakGAN_G.forward (models/LeakGAN_G.py:52-102)
// and the discriminator's get_feature (models/discriminator.py:51-64) -- gfx950 only.
//
// All rows advance one position together (row_ids: one leader row per sentence until its rows reach their given_num,
// then the (given_num, rollout index, sentence) rows); per position:
//   k_leak_draw    token of the position: the sentence's (a leader) or the canonical inverse-CDF draw from the previous
//                  position's outputs; written to the completions and to the feature extractor's input.  With small
//                  cells the worker's pass below runs in the same kernel (k_leak_draw<true>).
//   k_leak_worker  worker LSTM cell (token embedding) and the projection operand z = temperature * [rg (x) h_w | rg]
//   projection     outputs[n, v] = z[n, :] . [work2goal.weight[v*K..v*K+K, :] | work2goal.bias[v*K..]]: k_leak_proj
//                  (exact fp32 MFMA) or, TG_GEN_SPLIT_F16, tg_pgw.hip's split-fp16 fragment GEMM on z written as pieces
// and, beside that chain:
//   tg_dis_features  the scorer's get_feature over the rows' tokens so far (tg_dis.hip: conv + max-pool + highway on
//                  the matrix cores; weights prepared once per call)
//   k_leak_mgates, k_leak_manager  manager LSTM cell (feature), the goal bookkeeping
//
// Streams.  The manager's side reaches the tokens only through the real goal rg, which is rebuilt every step_size
// positions (utils/rollout.py:89-93); between two rebuilds the tokens' chain draw -> worker -> projection -> draw needs
// nothing from it, and the feature passes of different positions need nothing from each other (only the manager's
// cells are a chain).  So the call spreads over four streams: the caller's (tokens' chain), two of the device's own
// for the feature passes of even and odd positions, one for the manager's cells; events order them (draw(pos) ->
// feature(pos) -> manager(pos) -> worker(pos + 1) where pos rebuilt the goal).  Feature buffers and scorer call
// workspaces come in sets of kFeatSets, used round-robin by position (manager(pos - kFeatSets) -> feature(pos) for
// their reuse); the scorer's prepared weights are shared by all sets and written once per call.  rg is a pair of
// buffers (the one in force, the one being rebuilt) and the feature pass reads positions > pos as padding by
// argument, so no stream reads what another is writing.  Manager passes behind the last rebuild that a worker pass
// still reads are not run.  Results do not depend on the spread: TG_LEAK_ONE_STREAM=1 puts everything on the caller's
// stream (same bits, for timing comparisons).
//
// Goal algebra.  The reference keeps cur_goal = normalize(mana2goal(h_m)) (G = 1720 values per row and position)
// and uses it only through rg = normalize(goal2goal(real_goal)), real_goal = a sum of four cur_goals (+ goal_init).
// goal2goal is linear and bias-free, so  goal2goal(cur_goal) = (P h_m + q) / ||W h_m + b||  with P = W_g2g W_m2g
// [K,H], q = W_g2g b_m2g and  ||W h_m + b||^2 = h_m^T Q h_m + 2 c^T h_m + s  (Q = W^T W, c = W^T b, s = b^T b): per
// row and position K + 1 numbers instead of G.  P, q, Q, c, s and goal2goal(goal_init[b]) are computed once per call
// and evaluated in fp64 (no cancellation worry, more accurate than the reference's fp32 sums).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "tg_common.h"
#include "tg_dis.h"
#include "tg_pgw.h"

namespace {

constexpr int kLRows = 32;      // rows per workgroup of k_leak_worker / k_leak_manager
constexpr int kLThreads = 256;
constexpr int kMaxT = 64;       // given_num values per call
constexpr int kNSpan = 8;       // canonical sampler: spans per row
constexpr int kMaxHLeak = 128;
constexpr int kMaxDevices = 64;
constexpr int kFeatSets = 4;     // feature buffers and scorer call workspaces: positions p, p+1, .. p+3 in flight

// per device: the three side streams of a call (feature passes of even / odd positions, manager cells) and the events
// that order them against each other and the caller's stream
struct LeakSide {
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev;
};
std::mutex g_side_mu;
LeakSide g_side[kMaxDevices];

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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
    const float* rg_rd;        // [N,K] normalised goal2goal(real_goal) in force at this position (k_leak_worker reads)
    float* rg_wr;              // [N,K] the other half of the pair: k_leak_manager writes the next one there
    double* ring;              // [4][N][K] goal2goal(cur_goal) of the last four positions
    float* z;                  // [N, K*H + K] projection operand (exact-fp32 projection)
    f16x8* zfrag;              // split projection: the same operand as the fragment GEMM's A pieces (tg_pgw.h), scaled
    int zKB;                   //   by sZ; 16-wide k blocks per 32-row block
    float sZ;
    int split;                 // 1: TG_GEN_SPLIT_F16
    float* mpre;               // [N, 4H] manager.weight_ih . feature of this position (k_leak_mgates)
    float* logits;             // [N,V]
    const double* prep;        // P [K*H], q [K], Q [H*H], c [H], s [1], gp [B*K]
    int given[kMaxT];
    int order[kMaxT];          // the given_num indices in ascending order of given_num: order[rank] = t
    int B, L, T, i_lo;
    int N;                     // rows that take part at this position (a prefix of the chunk's rows, see row_ids)
    int Nmax;                  // rows of the chunk: B + ni * T * B
    int i0, ni;                // first rollout index of the chunk, number of them
    int pos;
    int Ls, GK, NG;            // canonical sampler geometry
    long long* stamps;         // diagnostic (TG_LEAK_STAMPS, -DTG_DIAG=1): k_leak_cells phase cycles, [workgroup][8]
};

// Rows of a chunk.  Until its given_num a row is fed its sentence and nothing it computes depends on its rollout index
// or on which given_num it belongs to, so that part is computed once per sentence: rows 0 .. B-1 are the LEADERS, one
// per sentence, fed the sentence at every position (t = -1).  Behind them the (given_num, rollout index, sentence) rows
// with the given_nums in ascending order: a row joins at pos == given_num -- its state copied from its leader
// (k_leak_fork), its first draw taken from the leader's outputs -- so the rows taking part at a position are a
// prefix of the array (run_leakgan.py's shape: 13 900 row-steps per call instead of 24 300).
__device__ __forceinline__ void row_ids(const LeakArgs& a, int n, int& i, int& t, int& b) {
    if (n < a.B) { b = n; t = -1; i = a.i0; return; }
    const int m = n - a.B;
    b = m % a.B;
    i = a.i0 + (m / a.B) % a.ni;
    t = a.order[m / (a.B * a.ni)];
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

// Piece p = (k block, half) of row n of the projection operand z = temperature * [rg (x) h_w | rg] (zero past K*H + K),
// scaled, as fp16 hi / lo halves at their place in the fragment GEMM's A operand: the operand never exists in fp32.
__device__ __forceinline__ void z_store_frag(const LeakArgs& a, int n, int p, const float* rs, const float* hs, float temp) {
    const int H = a.w.H, K = a.w.K, KH = K * H;
    const int kb = p >> 1, half = p & 1, k0 = 16 * kb + 8 * half;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int q = k0 + e;
        float v = 0.f;
        if (q < KH) v = temp * rs[q / H] * hs[q % H];
        else if (q < KH + K) v = temp * rs[q - KH];
        const float x = v * a.sZ;
        const _Float16 xh = (_Float16)x;
        h[e] = xh;
        l[e] = (_Float16)(x - (float)xh);
    }
    f16x8* dst = a.zfrag + (((size_t)(n >> 5) * a.zKB + kb) * 2) * 64 + (n & 31) + 32 * half;
    dst[0] = h;
    dst[64] = l;
}

// work2goal as one [V, K*H + K] matrix (row v: its K rows of the weight, then its K biases): the projection's B operand
__global__ __launch_bounds__(256) void k_leak_wcat(TgLeakWeights w, float* wcat) {
    const int KH = w.K * w.H, ld = KH + w.K;
    const size_t total = (size_t)w.V * ld;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t v = i / ld;
        const int q = (int)(i - v * ld);
        wcat[i] = q < KH ? w.w2g_w[v * KH + q] : w.w2g_b[v * w.K + (q - KH)];
    }
}

// scal[0] = sW (power of two putting |max| of wcat near 2^14), [1] = sZ, [2] = 1 / (sW sZ)
__global__ void k_leak_scales(const unsigned* absmax, float sZ, float* scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float m = __uint_as_float(absmax[0]);
    int ex = 0;
    float sW = 1.f;
    if (m > 0.f && m < __builtin_inff()) {
        (void)frexpf(m, &ex);
        sW = ldexpf(1.f, 14 - ex);
    }
    scal[0] = sW;
    scal[1] = sZ;
    scal[2] = 1.f / (sW * sZ);
}

// ---- per chunk.  Leaders: zero state, zero-padded feature input, rg = normalize(goal2goal(goal_init[b])).  The other
// rows: the sentence's first given_num tokens in the completions and in the feature input (their state arrives with
// k_leak_fork).
__global__ __launch_bounds__(256) void k_leak_init(LeakArgs a) {
    const int n = blockIdx.x;
    const int H = a.w.H, K = a.w.K, L = a.L;
    int i, t, b;
    row_ids(a, n, i, t, b);
    if (t >= 0) {
        const size_t orow = (((size_t)i * a.T + t) * a.B + b) * L;
        for (int p = threadIdx.x; p < L; p += 256) {
            const bool given = p < a.given[t];
            const int64_t tk = given ? a.sentences[(size_t)b * L + p] : 0;
            if (given) a.out[orow + p] = tk;
            a.fed[(size_t)n * L + p] = tk;
        }
        return;
    }
    for (int h = threadIdx.x; h < H; h += 256) {
        a.hw[(size_t)n * H + h] = 0.f; a.cw[(size_t)n * H + h] = 0.f;
        a.hm[(size_t)n * H + h] = 0.f; a.cm[(size_t)n * H + h] = 0.f;
    }
    for (int p = threadIdx.x; p < L; p += 256) a.fed[(size_t)n * L + p] = 0;
    for (int q = threadIdx.x; q < 4 * K; q += 256) a.ring[((size_t)(q / K) * a.Nmax + n) * K + q % K] = 0.0;
    if (threadIdx.x == 0) {
        const double* gp = a.prep + K * H + K + H * H + H + 1 + (size_t)b * K;
        double ss = 0.0;
        for (int k = 0; k < K; ++k) ss += gp[k] * gp[k];
        double nrm = sqrt(ss);
        nrm = nrm < 1e-12 ? 1e-12 : nrm;
        for (int k = 0; k < K; ++k) a.rg_wr[(size_t)n * K + k] = (float)(gp[k] / nrm);
    }
}

// ---- rows [first, first + gridDim.x) reach their given_num at this position: the state their leaders hold after the
// position before.  which 0 (caller's stream, before the draw): worker h, c and the goal in force (rg_wr = the buffer
// rg_rd points into, for this launch); which 1 (manager's stream, before the manager pass): manager h, c, goal ring.
__global__ __launch_bounds__(256) void k_leak_fork(LeakArgs a, int first, int which) {
    const int n = first + blockIdx.x, b = (n - a.B) % a.B;
    const int H = a.w.H, K = a.w.K, tid = threadIdx.x;
    if (which == 0) {
        for (int h = tid; h < H; h += 256) {
            a.hw[(size_t)n * H + h] = a.hw[(size_t)b * H + h];
            a.cw[(size_t)n * H + h] = a.cw[(size_t)b * H + h];
        }
        for (int k = tid; k < K; k += 256) a.rg_wr[(size_t)n * K + k] = a.rg_rd[(size_t)b * K + k];
    } else {
        for (int h = tid; h < H; h += 256) {
            a.hm[(size_t)n * H + h] = a.hm[(size_t)b * H + h];
            a.cm[(size_t)n * H + h] = a.cm[(size_t)b * H + h];
        }
        for (int q = tid; q < 4 * K; q += 256) {
            double* slot = a.ring + (size_t)(q / K) * a.Nmax * K + q % K;
            slot[(size_t)n * K] = slot[(size_t)b * K];
        }
    }
}

// ---- token of position a.pos for every row (one workgroup per row)
// A leader: the sentence's token.  Any other row (pos >= given_num, or it would not take part): the canonical inverse-CDF draw (DESIGN.md section 4;
// oracle/textgan_oracle.c canonical_draw) from w[v] = exp(logits[n,v] - max): tile sums and group sums fp32
// sequential, span sums / total / target / every prefix comparison fp64.
//
// CELL: the worker's pass of the position for the same row follows in the same workgroup (k_leak_worker's work on the
// vector units: a launch less per position on the chain the tokens wait for; chosen by the host when the cell's weights
// are small, see tg_leak_rollout).  Gate pre-activations in the oracle's order: b_ih + b_hh, then the input terms,
// then the recurrent terms, one fused multiply-add each (oracle/textgan_oracle.c lstm_cell_vec).
__device__ __forceinline__ void leak_draw_row(const LeakArgs& a, int n, int i, int t, int b, float* lds, int64_t* fed_tok);
__device__ __forceinline__ void leak_cell_row(const LeakArgs& a, int n, int64_t tk, float* cell);

template <bool CELL>
__global__ __launch_bounds__(256) void k_leak_draw(LeakArgs a) {
    extern __shared__ float lds[];
    __shared__ int64_t fed_tok;
    const int n = blockIdx.x;
    int i, t, b;
    row_ids(a, n, i, t, b);
    leak_draw_row(a, n, i, t, b, lds, &fed_tok);
    if constexpr (CELL) {
        if (a.pos == a.L - 1) return;   // the reference's forward after the last token is never read
        __syncthreads();
        leak_cell_row(a, n, fed_tok, lds);
    }
}

__device__ __forceinline__ void leak_cell_row(const LeakArgs& a, int n, int64_t tk, float* cell) {
    const int H = a.w.H, K = a.w.K, E = a.w.E, tid = threadIdx.x;
    float* xs = cell;          // [E] embedding of the token fed
    float* hs = xs + E;        // [H] h before, then after
    float* gs = hs + H;        // [4H] gate pre-activations
    float* rs = gs + 4 * H;    // [K] the real goal in force
    tk = tk < 0 ? 0 : (tk >= a.w.V ? a.w.V - 1 : tk);
    for (int e = tid; e < E; e += 256) xs[e] = a.w.emb[(size_t)tk * E + e];
    for (int h = tid; h < H; h += 256) hs[h] = a.hw[(size_t)n * H + h];
    for (int k = tid; k < K; k += 256) rs[k] = a.rg_rd[(size_t)n * K + k];
    __syncthreads();
    for (int r = tid; r < 4 * H; r += 256) {
        float acc = a.w.w_bih[r] + a.w.w_bhh[r];
        const float* wi = a.w.w_wih + (size_t)r * E;
        const float* wh = a.w.w_whh + (size_t)r * H;
        for (int k = 0; k < E; ++k) acc = __builtin_fmaf(wi[k], xs[k], acc);
        for (int k = 0; k < H; ++k) acc = __builtin_fmaf(wh[k], hs[k], acc);
        gs[r] = acc;
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {
        const float ig = tg_sigmoid(gs[j]);
        const float fg = tg_sigmoid(gs[H + j]);
        const float gg = tg_tanh(gs[2 * H + j]);
        const float og = tg_sigmoid(gs[3 * H + j]);
        const float cn = fg * a.cw[(size_t)n * H + j] + ig * gg;
        const float hn = og * tg_tanh(cn);
        a.cw[(size_t)n * H + j] = cn;
        a.hw[(size_t)n * H + j] = hn;
        hs[j] = hn;
    }
    __syncthreads();
    // z = temperature * [rg (x) h_w | rg] (k_leak_worker's expression)
    const float temp = a.pos > 1 ? 1.0f : a.w.temperature;
    const int ldz = K * H + K;
    if (a.split) {
        for (int p = tid; p < 2 * a.zKB; p += 256) z_store_frag(a, n, p, rs, hs, temp);
        return;
    }
    float* z = a.z + (size_t)n * ldz;
    for (int q = tid; q < ldz; q += 256) {
        const bool tail = q >= K * H;
        const int kq = tail ? q - K * H : q / H, hq = tail ? 0 : q % H;
        const float rgk = rs[kq];
        z[q] = tail ? temp * rgk : temp * rgk * hs[hq];
    }
}

__device__ __forceinline__ void leak_draw_row(const LeakArgs& a, int n, int i, int t, int b, float* lds, int64_t* fed_tok) {
    const int tid = threadIdx.x;
    const int V = a.w.V, L = a.L, pos = a.pos;
    if (t < 0) {
        if (tid == 0) {
            const int64_t tk = a.sentences[(size_t)b * L + pos];
            a.fed[(size_t)n * L + pos] = tk;
            a.tok[n] = tk;
            *fed_tok = tk;
        }
        return;
    }
    const size_t orow = (((size_t)i * a.T + t) * a.B + b) * L;
    const int Ls = a.Ls, GK = a.GK, NG = a.NG, NT = Ls / 16;
    float* ts = lds;                      // [8*NT] tile sums
    float* gs = ts + kNSpan * NT;         // [8*NG] group sums
    double* ss = reinterpret_cast<double*>(gs + kNSpan * NG + ((kNSpan * (NT + NG)) & 1));  // [8] span sums
    float* red = reinterpret_cast<float*>(ss + kNSpan);                                    // [256]
    // the row's own outputs of the position before -- its leader's when it joins here
    const float* lg = a.logits + (size_t)(pos == a.given[t] ? b : n) * V;
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
    *fed_tok = fedtok;
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

// ---- worker cell and projection operand for 32 rows (LeakGAN_G.py:69-71, 80-95).  Needs the token of the position and
// the real goal in force; nothing of this position's feature.
__global__ __launch_bounds__(kLThreads) void k_leak_worker(LeakArgs a) {
    extern __shared__ float lds[];
    const int H = a.w.H, K = a.w.K, E = a.w.E, N = a.N;
    float* g = lds;                     // [4H][33]
    float* hl = g + 4 * H * 33;         // [32][H+1]
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
    for (int idx = tid; idx < rows * K; idx += kLThreads) g[idx] = a.rg_rd[(size_t)n0 * K + idx];   // g is free here (<= 32 x 64)
    __syncthreads();
    if (a.split) {
        const int np = 2 * a.zKB;
        for (int idx = tid; idx < rows * np; idx += kLThreads) {
            const int r = idx / np;
            z_store_frag(a, n0 + r, idx - r * np, g + r * K, hl + r * (H + 1), temp);
        }
    }
    for (int q = tid; q < (a.split ? 0 : ldz); q += kLThreads) {
        const bool tail = q >= K * H;
        const int kq = tail ? q - K * H : q / H, hq = tail ? 0 : q % H;
        for (int r = 0; r < rows; ++r) {
            const float rgk = g[r * K + kq];
            a.z[(size_t)(n0 + r) * ldz + q] = tail ? temp * rgk : temp * rgk * hl[r * (H + 1) + hq];
        }
    }
    if (a.stamps && tid == 0) {
        const long long c2 = tg_clock();
        long long* o = a.stamps + (size_t)blockIdx.x * 8;
        o[0] += c1 - c0; o[1] += c2 - c1; o[4] += 1;
    }
}

// ---- manager cell and goal bookkeeping for 32 rows (LeakGAN_G.py:73-78; utils/rollout.py:89-93, 117-121).  Needs the
// feature of the tokens up to this position (k_leak_mgates' product); feeds the worker only through the real goal, at
// the positions where that is rebuilt.
__global__ __launch_bounds__(kLThreads) void k_leak_manager(LeakArgs a) {
    extern __shared__ float lds[];
    const int H = a.w.H, K = a.w.K, G = a.w.G, N = a.N;
    float* g = lds;                     // [4H][33]
    float* hl = g + 4 * H * 33;         // [32][H+1]
    double* dred = reinterpret_cast<double*>(hl + 32 * (H + 1) + ((32 * (H + 1)) & 1));  // [32][8] partial norms
    double* pk = dred + 32 * 8;         // [32][K] goal2goal(cur_goal) of this position
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blockIdx.x * kLRows;
    const int j = lane & 31;
    const int nj = tg_min(n0 + j, N - 1);

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
        double* slot = a.ring + ((size_t)(pos & 3) * a.Nmax + n) * K;
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
                    for (int jj = pos - 3; jj <= pos; ++jj) v += a.ring[((size_t)(jj & 3) * a.Nmax + n) * K + k];
                if (pos == step) v += gp[k];
                pk[r * K + k] = v;
                ss += v * v;
            }
            double rn = sqrt(ss);
            rn = rn < 1e-12 ? 1e-12 : rn;
            for (int k = 0; k < K; ++k) a.rg_wr[(size_t)n * K + k] = (float)(pk[r * K + k] / rn);
        }
    }
    if (a.stamps && tid == 0) {
        const long long c4 = tg_clock();
        long long* o = a.stamps + (size_t)blockIdx.x * 8;
        o[2] += c3 - c2; o[3] += c4 - c3;
    }
}

// ---- outputs[n, v] = z[n, :] . [work2goal.weight rows v*K .. v*K+K-1 | work2goal.bias[v*K ..]]
// (work2goal(h_w).view(V, K) @ rg, LeakGAN_G.py:80-84, times the temperature folded into z).  Workgroup tile 128
// vocabulary entries x 128 rows, 2 x 2 waves of 64 x 64 on v_mfma_f32_32x32x2_f32; both operands read from L2 with
// 16-byte loads.  (Every row that takes part draws its next token from these, or lends them to the rows that join.)
__global__ __launch_bounds__(256) void k_leak_proj(LeakArgs a) {
    const int V = a.w.V, K = a.w.K, H = a.w.H, N = a.N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int v0 = blockIdx.x * 128 + (wave & 1) * 64;
    const int r0 = blockIdx.y * 128 + (wave >> 1) * 64;
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
    size_t fed, tok, feat, hw, cw, hm, cm, rg, ring, z, mpre, logits, prep, dis_w, dis, wcat, wfrag, zfrag, scal, total;
    size_t dis_call;   // bytes of one scorer call workspace
    int rows;  // rows of one chunk
};

int span_len(int V) { return 16 * ((V + kNSpan * 16 - 1) / (kNSpan * 16)); }

// rows per chunk: the B leaders and whole rollout indices (T*B rows each), logits of a chunk at most ~1 GiB
int chunk_indices(const TgLeakWeights* g, int B, int T, int n_i) {
    const size_t per_i = (size_t)T * B, row = g->V * sizeof(float);
    size_t budget = (size_t)1 << 30;
    if (const char* e = getenv("TG_LEAK_LOGITS_BYTES")) {   // diagnostic: a small budget forces several chunks
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v > 0) budget = (size_t)v;
    }
    const size_t room = budget / row;
    size_t fit = room > (size_t)B ? (room - B) / per_i : 0;
    if (fit < 1) fit = 1;
    return (int)(fit < (size_t)n_i ? fit : (size_t)n_i);
}

LeakLayout layout(const TgLeakWeights* g, const TgDisWeights* d, int B, int L, int T, int n_i) {
    LeakLayout o = {};
    const int ci = chunk_indices(g, B, T, n_i);
    const size_t N = (size_t)B + (size_t)ci * T * B;
    const int H = g->H, K = g->K;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += al256(bytes); return at; };
    o.rows = (int)N;
    o.fed = take(N * L * sizeof(int64_t));
    o.tok = take(N * sizeof(int64_t));
    o.feat = take(kFeatSets * N * g->G * sizeof(float));   // one per position in flight
    o.hw = take(N * H * sizeof(float));
    o.cw = take(N * H * sizeof(float));
    o.hm = take(N * H * sizeof(float));
    o.cm = take(N * H * sizeof(float));
    o.rg = take(2 * N * K * sizeof(float));   // the goal in force and the one being rebuilt
    o.ring = take(4 * N * K * sizeof(double));
    const bool split = g->precision == TG_GEN_SPLIT_F16;
    const long ldz = (long)K * H + K;
    o.z = take(split ? 0 : N * (size_t)ldz * sizeof(float));
    o.wcat = take(split ? (size_t)g->V * ldz * sizeof(float) : 0);
    o.wfrag = take(split ? tg_fg_bytes(g->V, ldz) : 0);
    o.zfrag = take(split ? tg_fg_bytes((long)N, ldz) : 0);
    o.scal = take(split ? 256 : 0);
    o.mpre = take(N * (size_t)4 * H * sizeof(float));
    o.logits = take(N * (size_t)g->V * sizeof(float));
    o.prep = take(((size_t)K * H + K + (size_t)H * H + H + 1 + (size_t)B * K) * sizeof(double));
    o.dis_w = take(tg_dis_features_weights_bytes(d, L));
    o.dis_call = al256(tg_dis_features_call_bytes(d, (int)N, L));
    o.dis = take(kFeatSets * o.dis_call);
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
    if (g->precision != TG_GEN_FP32 && g->precision != TG_GEN_SPLIT_F16) {
        tg_set_error("tg_leak_rollout: unknown precision %d", g->precision);
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
    a.ring = reinterpret_cast<double*>(ws + lo.ring);
    a.z = reinterpret_cast<float*>(ws + lo.z);
    a.split = g->precision == TG_GEN_SPLIT_F16;
    a.zfrag = reinterpret_cast<f16x8*>(ws + lo.zfrag);
    a.zKB = tg_fg_kb((long)g->K * g->H + g->K);
    {
        // |z| <= temperature (|rg| <= 1, |h_w| < 1): the power of two that keeps temperature * sZ below 2^15
        float t = g->temperature > 1.f ? g->temperature : 1.f;
        a.sZ = 16384.f;
        while (t > 1.f && a.sZ > 1.f) { t *= 0.5f; a.sZ *= 0.5f; }
    }
    a.mpre = reinterpret_cast<float*>(ws + lo.mpre);
    a.logits = reinterpret_cast<float*>(ws + lo.logits);
    a.prep = reinterpret_cast<const double*>(ws + lo.prep);
    for (int t = 0; t < T; ++t) a.given[t] = given_nums[t];
    for (int t = 0; t < T; ++t) {      // insertion sort of the indices by given_num (stable)
        int r = t;
        while (r > 0 && given_nums[a.order[r - 1]] > given_nums[t]) { a.order[r] = a.order[r - 1]; --r; }
        a.order[r] = t;
    }
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
    const size_t sm_worker = sizeof(float) * ((size_t)4 * H * 33 + (size_t)32 * (H + 1) + 2);
    const size_t sm_manager = sm_worker + sizeof(double) * ((size_t)32 * 8 + (size_t)32 * K);
    // the worker's cell rides in the draw kernel (one workgroup per row, vector units) while its weights are small
    // enough to be re-read by every row from L2; wider cells keep k_leak_worker (32 rows per workgroup on the MFMA)
    const char* sc = getenv("TG_LEAK_SPLIT_CELL");        // diagnostic: always k_leak_worker
    const bool fused_cell = sizeof(float) * 4 * H * (size_t)(g->E + H) <= 64 * 1024 && !(sc && sc[0] == '1');
    const size_t sm_cell = sizeof(float) * ((size_t)g->E + 5 * (size_t)H + K);
    const size_t sm_step = sm_draw > sm_cell ? sm_draw : sm_cell;
    static TgLdsAttr attr_draw, attr_step, attr_worker, attr_manager;
    if ((rc = tg_lds_attr(attr_draw, reinterpret_cast<const void*>(&k_leak_draw<false>), sm_draw, "k_leak_draw"))) return rc;
    if ((rc = tg_lds_attr(attr_step, reinterpret_cast<const void*>(&k_leak_draw<true>), sm_step, "k_leak_draw"))) return rc;
    if ((rc = tg_lds_attr(attr_worker, reinterpret_cast<const void*>(&k_leak_worker), sm_worker, "k_leak_worker"))) return rc;
    if ((rc = tg_lds_attr(attr_manager, reinterpret_cast<const void*>(&k_leak_manager), sm_manager, "k_leak_manager"))) return rc;
    if (sm_draw > 160 * 1024 || sm_manager > 160 * 1024) {
        tg_set_error("tg_leak_rollout: LDS need %zu / %zu bytes exceeds 160 KiB", sm_draw, sm_manager);
        return TG_ERR_ARG;
    }

    // Streams (file header, "Streams"): the caller's carries the tokens' chain; three of the device's own carry the
    // feature passes of even and of odd positions and the manager's cells.  Those streams and the events between them
    // belong to the device, so calls on one device take turns here (the lock covers the enqueueing only).
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        tg_set_error("tg_leak_rollout: hipGetDevice failed or device index %d >= %d", dev, kMaxDevices);
        return TG_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_side_mu);
    LeakSide& sd = g_side[dev];
    const char* one = getenv("TG_LEAK_ONE_STREAM");       // diagnostic: everything on the caller's stream
    const bool side = !(one && one[0] == '1');
    // (plain streams on purpose: with low-priority streams for the feature passes, two processes sharing one GPU
    // stopped making progress -- the caller's stream waiting at a barrier for a queue the scheduler no longer mapped)
    for (int q = 0; side && q < 3; ++q)
        if (!sd.stream[q] && hipStreamCreateWithFlags(&sd.stream[q], hipStreamNonBlocking) != hipSuccess) {
            tg_set_error("tg_leak_rollout: cannot create a side stream");
            return TG_ERR_LAUNCH;
        }
    hipStream_t fstream[2] = {side ? sd.stream[0] : stream, side ? sd.stream[1] : stream};
    hipStream_t mstream = side ? sd.stream[2] : stream;
    bool ev_fail = false;
    auto event = [&](size_t k) -> hipEvent_t {
        while (sd.ev.size() <= k) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ev_fail = true; return nullptr; }
            sd.ev.push_back(e);
        }
        return sd.ev[k];
    };
    // record(k, s): event k = everything s has been given so far; wait(s, k): what s is given next starts after it
    auto record = [&](size_t k, hipStream_t s) {
        hipEvent_t e = side ? event(k) : nullptr;
        if (side && (!e || hipEventRecord(e, s) != hipSuccess)) ev_fail = true;
    };
    auto wait = [&](hipStream_t s, size_t k) {
        hipEvent_t e = side ? event(k) : nullptr;
        if (side && (!e || hipStreamWaitEvent(s, e, 0) != hipSuccess)) ev_fail = true;
    };
    const size_t ev_draw = 0, ev_feat = (size_t)L, ev_mgr = 2 * (size_t)L;   // one event per position each
    const size_t ev_fork = 3 * (size_t)L, ev_prep = ev_fork + 1;

    // where the side streams may start: behind whatever the caller's stream did with this memory before the call, not
    // behind this call's own kernels on that stream (they write other parts of the workspace)
    record(ev_fork, stream);
    bool forked = true;
    const int n_prep = K * H + K + H * H + H + 1 + B * K;
    hipLaunchKernelGGL(k_leak_prep, dim3(n_prep), dim3(64), 0, stream, *g, B,
                       reinterpret_cast<double*>(ws + lo.prep));
    float* pscal = reinterpret_cast<float*>(ws + lo.scal);
    if (a.split) {
        // work2goal once per call as the projection GEMM's B pieces: [V, K*H + K] fp32, its |max|, the scales, the pieces
        const long ldz = (long)K * H + K;
        float* wcat = reinterpret_cast<float*>(ws + lo.wcat);
        unsigned* absmax = reinterpret_cast<unsigned*>(pscal + 16);
        if (hipMemsetAsync(absmax, 0, 16, stream) != hipSuccess) {
            tg_set_error("hipMemsetAsync failed");
            return TG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(k_leak_wcat, dim3(2048), dim3(256), 0, stream, *g, wcat);
        tg_fg_absmax(stream, wcat, (size_t)g->V * ldz, absmax);
        hipLaunchKernelGGL(k_leak_scales, dim3(1), dim3(64), 0, stream, absmax, a.sZ, pscal);
        tg_fg_to_frag(stream, wcat, ldz, 1, g->V, ldz, pscal, 0, ws + lo.wfrag);
    }
    const int ci = chunk_indices(g, B, T, i_hi - i_lo);
    // the manager's output reaches the tokens only through the real goal, rebuilt at pos > 0 with pos % step_size == 0
    // and read by the worker from pos + 1 on; the last worker pass is at L - 2, so the manager's chain ends at the
    // last such pos <= L - 3 (none: it never runs)
    int last_update = -1;
    for (int p = 1; p <= L - 3; ++p)
        if (p % g->step_size == 0) last_update = p;
    int prepared_n = -1;     // row count the scorer's workspaces were prepared for
    void* call_ws[kFeatSets];
    for (int q = 0; q < kFeatSets; ++q) call_ws[q] = ws + lo.dis + q * lo.dis_call;
    for (int i0 = i_lo; i0 < i_hi; i0 += ci) {
        const int ni = (i_hi - i0 < ci) ? i_hi - i0 : ci;
        a.i0 = i0;
        a.ni = ni;
        a.Nmax = B + ni * T * B;
        a.N = a.Nmax;
        float* rgbuf[2] = {reinterpret_cast<float*>(ws + lo.rg), reinterpret_cast<float*>(ws + lo.rg) + (size_t)a.Nmax * K};
        float* featbuf[kFeatSets];
        for (int q = 0; q < kFeatSets; ++q) featbuf[q] = reinterpret_cast<float*>(ws + lo.feat) + (size_t)q * a.Nmax * g->G;
        a.rg_rd = rgbuf[0];
        a.rg_wr = rgbuf[0];
        a.feat = featbuf[0];
        hipLaunchKernelGGL(k_leak_init, dim3(a.Nmax), dim3(256), 0, stream, a);
        if (last_update >= 0 && prepared_n != a.Nmax) {
            // the scorer's weights, once for all the call workspaces, on the first feature stream -- behind whatever
            // the caller's stream did with this memory before; the other feature stream waits for it once
            if (!forked) record(ev_fork, stream);   // a later chunk of another size: behind the chunks before it
            forked = false;
            wait(fstream[0], ev_fork);
            rc = tg_dis_features_prepare(d, a.Nmax, L, featbuf[0], call_ws, kFeatSets, lo.dis_call, ws + lo.dis_w, fstream[0]);
            if (rc) return rc;
            record(ev_prep, fstream[0]);
            wait(fstream[1], ev_prep);
            prepared_n = a.Nmax;
        }
        // rows taking part at a position: the leaders and the rows whose given_num it has reached (row_ids)
        auto n_live = [&](int pos) {
            int groups = 0;
            for (int t = 0; t < T; ++t) groups += given_nums[t] <= pos;
            return B + groups * ni * B;
        };
        const int step = g->step_size;
        // real goals rebuilt before position pos: the one in force there is rgbuf[epoch & 1]
        auto epoch_at = [&](int pos) { const int hi = pos - 1 < last_update ? pos - 1 : last_update; return hi >= 1 ? hi / step : 0; };
        auto rebuilds = [&](int pos) { return pos > 0 && pos <= last_update && pos % step == 0; };
        auto at = [&](int pos) {
            a.pos = pos;
            a.N = n_live(pos);
            a.rg_rd = rgbuf[epoch_at(pos) & 1];
            a.rg_wr = rgbuf[(epoch_at(pos) + 1) & 1];
            a.feat = featbuf[pos % kFeatSets];
        };
        int proj_rc = TG_OK;
        auto tokens_pass = [&](int pos) {
            at(pos);
            // the worker -- and the rows that join here -- read the goal the manager pass of the position before rebuilt
            if (rebuilds(pos - 1)) wait(stream, ev_mgr + pos - 1);
            const int joined = pos > 0 ? n_live(pos - 1) : B;
            if (a.N > joined) {
                LeakArgs f = a;
                f.rg_wr = const_cast<float*>(a.rg_rd);
                hipLaunchKernelGGL(k_leak_fork, dim3(a.N - joined), dim3(256), 0, stream, f, joined, 0);
            }
            tg_prof_begin(TG_KERNEL_LEAK_DRAW, stream);
            if (fused_cell)
                hipLaunchKernelGGL(k_leak_draw<true>, dim3(a.N), dim3(256), sm_step, stream, a);
            else
                hipLaunchKernelGGL(k_leak_draw<false>, dim3(a.N), dim3(256), sm_draw, stream, a);
            tg_prof_end(stream);
            if (pos == L - 1) return;   // the reference's forward after the last token is never read
            if (pos <= last_update) record(ev_draw + pos, stream);
            if (!fused_cell) {
                tg_prof_begin(TG_KERNEL_LEAK_CELLS, stream);
                hipLaunchKernelGGL(k_leak_worker, dim3((a.N + kLRows - 1) / kLRows), dim3(kLThreads), sm_worker, stream, a);
                tg_prof_end(stream);
            }
            tg_prof_begin(TG_KERNEL_LEAK_PROJ, stream);
            if (a.split) {
                const int rcp = tg_fg_gemm_rows(stream, a.zfrag, a.N, ws + lo.wfrag, g->V, (long)K * H + K, a.logits, g->V,
                                                pscal, 2);
                if (rcp) proj_rc = rcp;
            } else {
                hipLaunchKernelGGL(k_leak_proj, dim3((g->V + 127) / 128, (a.N + 127) / 128), dim3(256), 0, stream, a);
            }
            tg_prof_end(stream);
        };
        auto manager_pass = [&](int pos) -> int {
            at(pos);
            // the feature of the tokens up to pos: needs this position's draw; its output buffer and workspace were
            // last used by position pos - kFeatSets, whose manager pass must be through with them.  Tokens behind pos
            // are read as padding whatever the draws running ahead have written there by now.
            hipStream_t fs = fstream[pos & 1];
            wait(fs, ev_draw + pos);
            if (pos >= kFeatSets) wait(fs, ev_mgr + pos - kFeatSets);
            const int rc2 = tg_dis_features(d, a.fed, a.N, a.Nmax, L, pos + 1, a.feat, call_ws[pos % kFeatSets],
                                            lo.dis_call, ws + lo.dis_w, fs);
            if (rc2) return rc2;
            record(ev_feat + pos, fs);
            wait(mstream, ev_feat + pos);
            const int joined = pos > 0 ? n_live(pos - 1) : B;
            if (a.N > joined) hipLaunchKernelGGL(k_leak_fork, dim3(a.N - joined), dim3(256), 0, mstream, a, joined, 1);
            tg_prof_begin(TG_KERNEL_LEAK_CELLS, mstream);
            hipLaunchKernelGGL(k_leak_mgates, dim3((a.N + 31) / 32, (4 * g->H) / 32), dim3(256), 0, mstream, a);
            hipLaunchKernelGGL(k_leak_manager, dim3((a.N + kLRows - 1) / kLRows), dim3(kLThreads), sm_manager, mstream, a);
            tg_prof_end(mstream);
            record(ev_mgr + pos, mstream);
            return TG_OK;
        };
        // Enqueued in segments that end where a goal is rebuilt (the tokens' chain needs nothing from the side streams
        // inside one): the segment's token passes first, so that the caller's stream never waits for the host to get
        // through the side streams' launches, then its manager passes; the next segment's first worker waits for the
        // last of them.
        for (int pos = 0; pos < L;) {
            int seg_end = L - 1;
            for (int p = pos; p <= last_update; ++p)
                if (rebuilds(p)) { seg_end = p; break; }
            for (int p = pos; p <= seg_end; ++p) tokens_pass(p);
            if (proj_rc) return proj_rc;
            for (int p = pos; p <= seg_end && p <= last_update; ++p)
                if ((rc = manager_pass(p))) return rc;
            pos = seg_end + 1;
        }
        // whatever follows on the caller's stream (the next chunk's reset, the caller's own work, freeing the
        // workspace) comes after the side streams' last work: the manager pass at last_update, behind every feature pass
        if (last_update >= 0) wait(stream, ev_mgr + last_update);
    }
    if (ev_fail) {
        (void)hipGetLastError();
        tg_set_error("tg_leak_rollout: event record / wait between the streams failed");
        return TG_ERR_LAUNCH;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("tg_leak_rollout launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}
