// The generator's teacher-forced pass and the policy-gradient backward for shapes beyond E = H = 32 (gfx950).
//
// tg_gen.hip / tg_pg.hip keep one workgroup's weights and state on chip, which works for the reference's own
// generator (E = H = 32) and for H in {32..512} in the forward.  Every other shape (E, H multiples of 32 up to 1024:
// H = 96, 160, 768, ..., and the scaled config's backward) runs here as plain fp32 GEMMs on one LDS-tiled
// v_mfma_f32_32x32x2_f32 kernel (exact fp32 products, fp32 accumulation) plus element-wise kernels:
//
//   forward   per position t: gates = x_t W_ih^T + h_{t-1} W_hh^T, cell -> states[t+1]        (models/generator.py:47-51)
//             per chunk of (t, n) rows: logits = h W_out^T + b_out; lse, log p(target)         (:53-55, metrics/nll.py:59-61)
//   backward  per chunk of rows: logits again, dlogits = coeff (1[v = target] - exp(logits - lse)) in place,
//             dh_proj = dlogits W_out, dW_out += dlogits^T h, db_out += column sums
//             t = L-1 .. 0: pre-activations again, da (the 4H gate gradients) -> DA[t], dh_next = da W_hh
//             dW_ih = DA^T X, dW_hh = DA^T H_prev, db = column sums of DA, dX = DA W_ih scattered into d(emb)
//   (what autograd does behind SeqGAN_G.batchPGLoss models/SeqGAN_G.py:21-39 without the [B*L, V] log-softmax matrix
//   and the [B, L, V] one-hot both being alive at once: the logits exist one chunk of rows (<= 1 GiB) at a time.)
#include "tg_common.h"
#include "tg_pgw.h"

namespace {

constexpr int kBM = 128, kBN = 128, kBK = 32, kLd = kBM + 4;

// C[m][n] (+)= sum_k a(m,k) b(n,k) (+ bias[n]).  An operand is stored either with k contiguous ("rows": element
// (m,k) at X[row(m)*ld + k]) or with its own index contiguous (KMAJOR: element (m,k) at X[row(k)*ld + m]); row() goes
// through an optional int64 gather (embedding rows by token).
struct GemmP {
    const float* A; const float* B; float* C; const float* bias;
    const int64_t* a_gather; const int64_t* b_gather;
    long lda, ldb, ldc;
    int M, N, K, accumulate, gather_max;
    int kslice;       // k range of one blockIdx.z (a multiple of kBK; K when the contraction is not cut)
    long zstride;     // C offset per blockIdx.z (slices are summed afterwards in fixed order: k_sum_slices)
};

template <bool KMAJOR>
__device__ __forceinline__ void g_fetch(f32x4 (&v)[4], const float* X, long ld, const int64_t* gather, int gmax, int i0,
                                        int ni, int k0, int K, int tid) {
    // tile: 128 own-indices x 32 k.  Own index i0.., guarded against ni; k guarded against K; zero fill.
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (!KMAJOR) {
            const int i = i0 + (tid >> 3) + 32 * p, k = k0 + 4 * (tid & 7);
            if (i < ni && k < K) {
                long row = i;
                if (gather) { row = gather[i]; row = row < 0 ? 0 : (row >= gmax ? gmax - 1 : row); }
                const float* src = X + row * ld + k;
                if (k + 3 < K && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) r = *reinterpret_cast<const f32x4*>(src);
                else
                    for (int e = 0; e < 4; ++e) r[e] = (k + e < K) ? src[e] : 0.f;
            }
        } else {
            const int k = k0 + (tid >> 5) + 8 * p, i = i0 + 4 * (tid & 31);
            if (k < K && i < ni) {
                long row = k;
                if (gather) { row = gather[k]; row = row < 0 ? 0 : (row >= gmax ? gmax - 1 : row); }
                const float* src = X + row * ld + i;
                if (i + 3 < ni && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) r = *reinterpret_cast<const f32x4*>(src);
                else
                    for (int e = 0; e < 4; ++e) r[e] = (i + e < ni) ? src[e] : 0.f;
            }
        }
        v[p] = r;
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void g_commit(const f32x4 (&v)[4], float* S, int tid) {  // S[k][kLd], own index contiguous
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (!KMAJOR) {
            const int i = (tid >> 3) + 32 * p, k = 4 * (tid & 7);
#pragma unroll
            for (int e = 0; e < 4; ++e) S[(k + e) * kLd + i] = v[p][e];
        } else {
            const int k = (tid >> 5) + 8 * p, i = 4 * (tid & 31);
            *reinterpret_cast<f32x4*>(S + k * kLd + i) = v[p];
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void k_gemm32(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float As[kBK * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[kBK * kLd];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, kk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
    const int kbeg = (int)blockIdx.z * p.kslice, kend = kbeg + p.kslice < p.K ? kbeg + p.kslice : p.K;
    f32x16 acc[2][2] = {};
    f32x4 va[4], vb[4];
    g_fetch<A_KMAJOR>(va, p.A, p.lda, p.a_gather, p.gather_max, m0, p.M, kbeg, kend, tid);
    g_fetch<B_KMAJOR>(vb, p.B, p.ldb, p.b_gather, p.gather_max, n0, p.N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += kBK) {
        __syncthreads();
        g_commit<A_KMAJOR>(va, As, tid);
        g_commit<B_KMAJOR>(vb, Bs, tid);
        __syncthreads();
        if (k0 + kBK < kend) {
            g_fetch<A_KMAJOR>(va, p.A, p.lda, p.a_gather, p.gather_max, m0, p.M, k0 + kBK, kend, tid);
            g_fetch<B_KMAJOR>(vb, p.B, p.ldb, p.b_gather, p.gather_max, n0, p.N, k0 + kBK, kend, tid);
        }
#pragma unroll
        for (int kp = 0; kp < kBK / 2; ++kp) {
            const float* ar = As + (2 * kp + kk) * kLd + wm * 64 + i;
            const float* br = Bs + (2 * kp + kk) * kLd + wn * 64 + i;
            const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
            acc[0][0] = TG_MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = TG_MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = TG_MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = TG_MFMA32(a1, b1, acc[1][1]);
        }
    }
    // D lane (j = n, kk), reg r -> m = (r&3) + 8(r>>2) + 4kk
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int n = n0 + wn * 64 + qb * 32 + i;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + qa * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m >= p.M) continue;
                float* c = p.C + (long)blockIdx.z * p.zstride + (long)m * p.ldc + n;
                const float v = acc[qa][qb][r] + bv;
                *c = p.accumulate ? *c + v : v;
            }
        }
}

template <bool AK, bool BK>
void gemm(hipStream_t s, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
          const float* bias = nullptr, bool accumulate = false, const int64_t* a_gather = nullptr,
          const int64_t* b_gather = nullptr, int gather_max = 0) {
    if (M <= 0 || N <= 0) return;
    GemmP p = {A, B, C, bias, a_gather, b_gather, lda, ldb, ldc, M, N, K, accumulate ? 1 : 0, gather_max, K, 0};
    hipLaunchKernelGGL((k_gemm32<AK, BK>), dim3((N + kBN - 1) / kBN, (M + kBM - 1) / kBM), dim3(256), 0, s, p);
}

// ---- element-wise pieces
// idx[t*N + n] = inp[n*L + t]
__global__ __launch_bounds__(256) void k_tok_tn(const int64_t* inp, int N, int L, int64_t* idx) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)N * L) return;
    const int t = (int)(r / N), n = (int)(r % N);
    idx[r] = inp[(long)n * L + t];
}

__global__ __launch_bounds__(256) void k_zero(float* p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}


// ---- the recurrence, one launch per position ------------------------------------------------------------------
// An adversarial batch is 64 sequences: a [64 x 4H x H] product per position on 128 x 128 GEMM tiles is 16 workgroups
// each walking K behind two barriers per 32 -- ~40 us per launch, three launches per position, and the whole step was
// launch latency (27 of 29 ms at E = H = 512).  Here the input side of every position is one batched GEMM
// (X W_ih^T, in the backward also H_prev W_hh^T: all h_{t-1} are known there), and a position costs ONE launch:
// workgroup (unit tile of 32, row tile of 32), 8 waves, each wave one slice of the contraction with all its operand
// loads in flight at once (16-byte loads straight from L2: a lane's 16 consecutive k of its row, the MFMA's k order
// being free as long as A and B agree), partial tiles summed through LDS in wave order, then the element-wise part
// with threads remapped so that consecutive threads own consecutive units (coalesced state / gradient rows).
constexpr int kStepThreads = 512;
#define TG_MFMA16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The tile of one workgroup is T units x T rows, T = 32 (v_mfma_f32_32x32x2_f32) or, when 32 x 32 tiles would leave
// most of the chip idle (an adversarial batch: 64 rows), T = 16 (v_mfma_f32_16x16x4_f32: four times the workgroups, a
// quarter of the MFMA cycles each).  Lane (i = lane % T, kg = lane / T) holds T / 2 consecutive k of row i per chunk
// of 32 k and feeds the q-th of them to the q-th MFMA on both operands (the MFMA's k order is free as long as A and B agree).
template <int T> struct StepAcc { typedef f32x16 type; };
template <> struct StepAcc<16> { typedef f32x4v type; };

template <int T>
__device__ __forceinline__ void step_dot(typename StepAcc<T>::type& acc, const float* __restrict__ arow,
                                         const float* __restrict__ brow, int kg, int c0, int c1) {
    constexpr int KP = T / 2;            // k per lane and chunk of 32: 16 (T = 32, two k groups) or 8 (T = 16, four)
    constexpr int NQ = KP / 4;           // 16-byte loads per operand and chunk
    for (int c = c0; c < c1; c += 4) {
        f32x4 av[4][NQ], bv[4][NQ];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = c + j < c1 ? c + j : c1 - 1;
            const f32x4* pa = reinterpret_cast<const f32x4*>(arow + 32 * cc + KP * kg);
            const f32x4* pb = reinterpret_cast<const f32x4*>(brow + 32 * cc + KP * kg);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { av[j][q] = pa[q]; bv[j][q] = pb[q]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c + j < c1) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (T == 32) acc = TG_MFMA32(av[j][q][e], bv[j][q][e], acc);
                        else acc = TG_MFMA16x4(av[j][q][e], bv[j][q][e], acc);
                    }
            }
        }
    }
}

// this lane's accumulator -> part[unit][row] of its wave (D layouts of the two MFMAs)
template <int T>
__device__ __forceinline__ void step_spill(const typename StepAcc<T>::type& acc, float (*part)[T + 1], int i, int kg) {
    if constexpr (T == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(r & 3) + 8 * (r >> 2) + 4 * kg][i] = acc[r];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[4 * kg + r][i] = acc[r];
    }
}

// Forward position t: gates = preX[t] + b_ih + b_hh + h_{t-1} W_hh^T, cell -> states[t+1]   (models/generator.py:51)
// wave w: gate w & 3, k half w >> 2.  preX [L*N, 4H] (row t*N + n), states [L+1, N, 2, H].
template <int T>
__global__ __launch_bounds__(kStepThreads) void k_lstm_step_fwd(const float* __restrict__ preX, const float* __restrict__ w_hh,
                                                                const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                                float* __restrict__ states, int N, int H, int t) {
    __shared__ float part[8][T][T + 1];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane % T, kg = lane / T;
    const int u0 = T * blockIdx.x, n0 = T * blockIdx.y;
    const int g = w & 3, kh = w >> 2;
    const int nc = H >> 5, half = (nc + 1) >> 1;
    const int nrow = n0 + i < N ? n0 + i : N - 1;
    // the element-wise part's own operands are fetched first, so their round trip runs under the contraction
    constexpr int NP = (T * T + kStepThreads - 1) / kStepThreads;   // (unit, row) pairs per thread: 2 or 1
    float e_pre[NP][4], e_c[NP];
    bool e_ok[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = tid + kStepThreads * q, u = p % T, nl = p / T, n = n0 + nl;
        e_ok[q] = p < T * T && n < N;
        e_c[q] = 0.f;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) e_pre[q][gg] = 0.f;
        if (e_ok[q]) {
            const size_t row = (size_t)t * N + n;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int col = gg * H + u0 + u;
                e_pre[q][gg] = preX[row * 4 * H + col] + (b_ih[col] + b_hh[col]);
            }
            e_c[q] = states[row * 2 * H + H + u0 + u];
        }
    }
    typename StepAcc<T>::type acc = {};
    step_dot<T>(acc, w_hh + (size_t)(g * H + u0 + i) * H, states + ((size_t)t * N + nrow) * 2 * H, kg,
                kh ? half : 0, kh ? nc : half);
    step_spill<T>(acc, part[w], i, kg);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (!e_ok[q]) continue;
        const int p = tid + kStepThreads * q, u = p % T, nl = p / T, n = n0 + nl;
        float gate[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) gate[gg] = (part[gg][u][nl] + part[4 + gg][u][nl]) + e_pre[q][gg];
        const float ig = tg_sigmoid(gate[0]), fg = tg_sigmoid(gate[1]), gc = tg_tanh(gate[2]), og = tg_sigmoid(gate[3]);
        float* cur = states + ((size_t)(t + 1) * N + n) * 2 * H;
        const float cn = fg * e_c[q] + ig * gc;
        cur[H + u0 + u] = cn;
        cur[u0 + u] = og * tg_tanh(cn);
    }
}

// Backward position t: dh = dhproj[t] + da_{t+1} W_hh, then the cell's gate gradients da_t -> DA[t], dc carried in dcn.
// P [L*N, 4H]: pre-activations without bias (X W_ih^T + H_prev W_hh^T); w_hhT [H, 4H]; wave w: k slice w of 8.
template <int T>
__global__ __launch_bounds__(kStepThreads) void k_lstm_step_bwd(const float* __restrict__ P, const float* __restrict__ w_hhT,
                                                                const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                                const float* __restrict__ states, const float* __restrict__ dhproj,
                                                                float* __restrict__ dcn, float* __restrict__ DA, int N, int H,
                                                                int t, int last) {
    __shared__ float part[8][T][T + 1];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane % T, kg = lane / T;
    const int u0 = T * blockIdx.x, n0 = T * blockIdx.y;
    // the element-wise part's own operands first: their round trip runs under the contraction
    constexpr int NP = (T * T + kStepThreads - 1) / kStepThreads;
    float e_g[NP][4], e_dh[NP], e_cp[NP], e_ct[NP], e_dc[NP];
    bool e_ok[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = tid + kStepThreads * q, u = p % T, nl = p / T, n = n0 + nl;
        e_ok[q] = p < T * T && n < N;
        e_dh[q] = e_cp[q] = e_ct[q] = e_dc[q] = 0.f;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) e_g[q][gg] = 0.f;
        if (e_ok[q]) {
            const size_t row = (size_t)t * N + n;
            const int uu = u0 + u;
            const float* gp = P + row * 4 * H;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) e_g[q][gg] = gp[gg * H + uu] + (b_ih[gg * H + uu] + b_hh[gg * H + uu]);
            e_dh[q] = dhproj[row * H + uu];
            e_cp[q] = states[row * 2 * H + H + uu];
            e_ct[q] = states[((size_t)(t + 1) * N + n) * 2 * H + H + uu];
            e_dc[q] = last ? 0.f : dcn[(size_t)n * H + uu];
        }
    }
    if (!last) {
        const int nc = (4 * H) >> 5;                    // a multiple of 4
        const int c0 = (int)((long)nc * w / 8), c1 = (int)((long)nc * (w + 1) / 8);
        const int nrow = n0 + i < N ? n0 + i : N - 1;
        typename StepAcc<T>::type acc = {};
        step_dot<T>(acc, w_hhT + (size_t)(u0 + i) * 4 * H, DA + ((size_t)(t + 1) * N + nrow) * 4 * H, kg, c0, c1);
        step_spill<T>(acc, part[w], i, kg);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (!e_ok[q]) continue;
        const int p = tid + kStepThreads * q, u = p % T, nl = p / T, n = n0 + nl;
        const size_t row = (size_t)t * N + n;
        const int uu = u0 + u;
        float dh = e_dh[q];
        if (!last) {
            float sdot = part[0][u][nl];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) sdot += part[ww][u][nl];
            dh += sdot;
        }
        const float ig = tg_sigmoid(e_g[q][0]), fg = tg_sigmoid(e_g[q][1]), gg = tg_tanh(e_g[q][2]), og = tg_sigmoid(e_g[q][3]);
        const float tc = tg_tanh(e_ct[q]);
        const float dc = e_dc[q] + dh * og * (1.f - tc * tc);
        float* da = DA + row * 4 * H;
        da[uu] = dc * gg * ig * (1.f - ig);
        da[H + uu] = dc * e_cp[q] * fg * (1.f - fg);
        da[2 * H + uu] = dc * ig * (1.f - gg * gg);
        da[3 * H + uu] = dh * tc * og * (1.f - og);
        dcn[(size_t)n * H + uu] = dc * fg;
    }
}

// dst[c][r] = src[r][c]  (W_hh [4H, H] -> [H, 4H], once per backward call)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, int R, int Cn, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cn) tile[j][tx] = src[(size_t)(r0 + j) * Cn + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cn && r0 + tx < R) dst[(size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// one workgroup per row r = t*N + n of a logits chunk [rows, ldl]: lse (fp64 merge of per-thread (max, sum) pairs in a
// fixed order), log p(target); optionally turns the row into dlogits = coeff * (1[v == target] - exp(logit - lse)).
// row0 = first (t, n) row of the chunk.  lse_in != null: backward (the saved lse is used, nothing is written out).
__global__ __launch_bounds__(256) void k_row_softmax(float* logits, long ldl, int V, long row0, int N, int L,
                                                      const int64_t* target, float* lse_out, float* logp_out,
                                                      const float* lse_in, const float* coeff) {
    __shared__ double sm[256], ss[256];
    const long r = row0 + blockIdx.x;
    const int t = (int)(r / N), n = (int)(r % N);
    const long o = (long)n * L + t;
    float* lg = logits + (long)blockIdx.x * ldl;
    const int tid = threadIdx.x;
    if (lse_in) {
        const float lse = lse_in[o], cf = coeff[o];
        const int tg = (int)target[o];
        for (int v = tid; v < V; v += 256) lg[v] = cf * ((v == tg ? 1.f : 0.f) - tg_expf(lg[v] - lse));
        return;
    }
    float m = TG_NEG_INF;
    for (int v = tid; v < V; v += 256) m = __builtin_fmaxf(m, lg[v]);
    float s = 0.f;
    for (int v = tid; v < V; v += 256) s += tg_expf(lg[v] - m);
    sm[tid] = (double)m;
    ss[tid] = (double)s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            const double ma = sm[tid], mb = sm[tid + st];
            const double M = ma > mb ? ma : mb;
            double acc = 0.0;
            if (ss[tid] > 0.0) acc += ss[tid] * exp(ma - M);
            if (ss[tid + st] > 0.0) acc += ss[tid + st] * exp(mb - M);
            sm[tid] = M;
            ss[tid] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float lse = (float)(sm[0] + log(ss[0]));
        if (lse_out) lse_out[o] = lse;
        if (logp_out && target) {
            int64_t tg = target[o];
            tg = tg < 0 ? 0 : (tg >= V ? V - 1 : tg);
            logp_out[o] = lg[tg] - lse;
        }
    }
}

// d(emb)[tok[r]][e] += dX[r][e] (float atomics, as torch's embedding backward on a GPU); the padding_idx row stays 0
__global__ __launch_bounds__(256) void k_scatter_emb(const float* dX, const int64_t* idx, long rows, int E, int V,
                                                      int64_t padding_idx, float* demb) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= rows * E) return;
    const long r = q / E;
    const int e = (int)(q % E);
    const int64_t tk = idx[r];
    if (tk == padding_idx || tk < 0 || tk >= V) return;
    atomicAdd(demb + tk * E + e, dX[q]);
}

__global__ __launch_bounds__(256) void k_copy(const float* src, float* dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}


// ============================================================================================================
// Split-fp16 GEMMs for the three vocabulary products of the policy-gradient step (TG_GEN_SPLIT_F16):
//   logits = h W_out^T + b,   dh_proj = dlogits W_out,   dW_out = dlogits^T h
// -- 8 H V flops per token row, everything else in this file is noise beside them at V = 32 000.  Operands are scaled by
// a per-tensor power of two and written as (hi, lo) fp16 halves IN MFMA FRAGMENT ORDER (k_to_frag: piece
// ((rb*KB + kb)*2 + part)*64 + lane = 8 halves of row 32 rb + lane%32, k = 16 kb + 8 (lane/32) + 0..7), so the GEMM
// stages them with LDS-DMA as they lie and every ds_read_b128 is a conflict-free 1-KiB row: the structure of the
// scorer's highway GEMM (tg_dis.hip) and of the generator's vocabulary GEMM (tg_genw.hip).  A product is
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (22 mantissa bits per operand).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define TG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void fg_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// X (element (r, k) at X[r*rs + k*cs]) -> pieces [RBp][KB][hi|lo][64] x 8 halves of X * scale[si]; zero outside R x K.
// cs == 1: a lane reads 32 contiguous bytes; rs == 1 (a transposed view): the 32 rows of a piece are contiguous, so
// each of the 8 loads of a lane is one coalesced segment across the wave.
__global__ __launch_bounds__(256) void k_to_frag(const float* __restrict__ X, long rs, long cs, int R, int K,
                                                  const float* __restrict__ scale, int si, int RBp, int KB,
                                                  f16x8* __restrict__ dst) {
    const float sc = scale[si];
    const long total = (long)RBp * KB * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long q = i >> 6;
        const int kb = (int)(q % KB);
        const long rb = q / KB;
        const long r = 32 * rb + (lane & 31);
        const int k0 = 16 * kb + 8 * (lane >> 5);
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < R && k0 < K) {
            const float* p = X + r * rs + (long)k0 * cs;
            float v[8];
            if (cs == 1 && k0 + 8 <= K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (k0 + e < K) ? p[(long)e * cs] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[e] * sc;
                const _Float16 hh = (_Float16)x;
                h[e] = hh;
                l[e] = (_Float16)(x - (float)hh);
            }
        }
        dst[(q * 2) * 64 + lane] = h;
        dst[(q * 2 + 1) * 64 + lane] = l;
    }
}

__global__ __launch_bounds__(256) void k_absmax_u(const float* x, size_t n, unsigned* out) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);   // |x| as ordered bits
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m) atomicMax(out, m);
    }
}

// scal: [0] sW (W_out), [1] sH = 2^14 (|h| < 1), [2] 1/(sW sH), [3] sD (dlogits: |d| <= max |coeff|), [4] 1/(sD sW),
// [5] 1/(sD sH); absmax[0] = |max| W_out, absmax[1] = |max| coeff (0 in the forward)
__global__ void k_pgw_scales(const unsigned* absmax, float* scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto pow2_for = [](float m) {
        int ex = 0;
        float sc = 1.f;
        if (m > 0.f && m < __builtin_inff()) {
            (void)frexpf(m, &ex);          // m < 2^ex
            sc = ldexpf(1.f, 14 - ex);     // m * scale in [2^13, 2^14)
        }
        return sc;
    };
    const float sW = pow2_for(__uint_as_float(absmax[0])), sH = 16384.f, sD = pow2_for(__uint_as_float(absmax[1]));
    scal[0] = sW; scal[1] = sH; scal[2] = 1.f / (sW * sH);
    scal[3] = sD; scal[4] = 1.f / (sD * sW); scal[5] = 1.f / (sD * sH);
}

// out[n*ldo + m] (+)= inv * sum_k A(m,k) B(n,k) (+ bias[m]); A pieces [MBp][KB][2][64], B pieces [RBp][KB][2][64].
// Persistent workgroups, XCD-aware tile walk (the 8 workgroups of a group of 8 logical ids share an A tile and take 8
// row tiles: 8 consecutive ids go to the 8 XCDs, so ids that share an XCD walk the same row tiles), two LDS stages of
// one k step (32) filled by LDS-DMA, one barrier per step.  nz > 1: the k steps are cut into nz slices, slice z
// writes out + z*zstride (summed afterwards in fixed order by k_sum_slices).
struct FgArgs {
    const f32x4* Af; const f32x4* Bf;
    int KB, MT, RT, NB, nz, gx;
    float* out; long ldo, zstride;
    int M, N, vec, accumulate;
    const float* scal; int inv_idx;
    const float* bias;
    int tstore;   // 1: out[m*ldo + n] (+ bias[n]) -- the 32 lanes of a store are 32 consecutive n: whole 128-byte lines per
                  // instruction where the 16-byte form (4 consecutive m per lane, rows ldo apart) writes 32-byte pieces
};

template <int WM, int WN, int TM, int TN, bool TSTORE>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_fgemm(const FgArgs a) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int MBT = WM * TM, RBT = WN * TN;
    constexpr int NLA = MBT * 256 / NTHR, NLB = RBT * 256 / NTHR;
    constexpr int STAGE = (MBT + RBT) * 256;
    extern __shared__ __attribute__((aligned(16))) f32x4 flds[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wm = w / WN, wn = w % WN;
    const int z = (int)blockIdx.x / a.gx, bx = (int)blockIdx.x % a.gx;
    const int per_xcd = a.NB >> 3, stride = a.gx >> 3;
    const int lb_end = (bx & 7) * per_xcd + per_xcd;
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
    int lb = next_valid((bx & 7) * per_xcd + (bx >> 3), mb0, rb0);
    if (lb >= lb_end) return;
    const int S = a.KB >> 1;
    const int s_lo = (int)((long)S * z / a.nz), s_hi = (int)((long)S * (z + 1) / a.nz);
    if (s_lo >= s_hi) return;
    const int pw = tid & 255;
    auto load_stage = [&](int mb, int rb, int s, int buf) {
        f32x4* d = flds + buf * STAGE + tid;
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            fg_glds16(a.Af + ((size_t)(mb + ((tid + i * NTHR) >> 8)) * a.KB + 2 * s) * 128 + pw, d + i * NTHR);
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            fg_glds16(a.Bf + ((size_t)(rb + ((tid + i * NTHR) >> 8)) * a.KB + 2 * s) * 128 + pw, d + MBT * 256 + i * NTHR);
    };
    f32x16 acc[TM][TN];
    load_stage(mb0, rb0, s_lo, 0);
    __syncthreads();
    int gs = 0;
    float* outz = a.out + (size_t)z * a.zstride;
    for (;;) {
        int nmb0 = 0, nrb0 = 0;
        const int nlb = next_valid(lb + stride, nmb0, nrb0);
        const bool has_next = nlb < lb_end;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int s = s_lo; s < s_hi; ++s, ++gs) {
            if (s + 1 < s_hi) load_stage(mb0, rb0, s + 1, (gs + 1) & 1);
            else if (has_next) load_stage(nmb0, nrb0, s_lo, (gs + 1) & 1);
            const f32x4* la = flds + (gs & 1) * STAGE + (TM * wm) * 256 + lane;
            const f32x4* lbp = flds + (gs & 1) * STAGE + MBT * 256 + (TN * wn) * 256 + lane;
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
            __syncthreads();
        }
        // ---- epilogue: D reg r of lane (n, hh) -> m = 32 mb + (r&3) + 8 (r>>2) + 4 hh, row 32 rb + n
        if constexpr (TSTORE) {
            const float inv = a.scal[a.inv_idx];
            const int mbw = mb0 + TM * wm, rbw = rb0 + TN * wn;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = 32 * (rbw + ni) + n;
                const bool col_ok = col < a.N;
                const float bv = (a.bias && col_ok) ? a.bias[col] : 0.f;
                // one running pointer, advanced row by row (128 precomputed 64-bit addresses would not fit the registers)
                float* d = outz + (size_t)(32 * mbw + 4 * hh) * a.ldo + col;
                int m = 32 * mbw + 4 * hh;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (col_ok && m < a.M) {
                            float v = acc[mi][ni][r] * inv + bv;
                            if (a.accumulate) v += *d;
                            *d = v;
                        }
                        const int adv = (r & 3) == 3 ? 5 : 1;   // rows (r&3) + 8 (r>>2): +1 three times, then +5
                        d += (size_t)adv * a.ldo;
                        m += adv;
                        asm volatile("" : "+v"(d), "+v"(m));
                    }
            }
        } else {
            const float inv = a.scal[a.inv_idx];
            const int mbw = mb0 + TM * wm, rbw = rb0 + TN * wn;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int row = 32 * (rbw + ni) + n;
                if (row >= a.N) continue;
                float* orow = outz + (size_t)row * a.ldo;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int m = 32 * (mbw + mi) + 8 * rq + 4 * hh;
                        if (m >= a.M) continue;
                        if (a.vec && m + 3 < a.M) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * rq + e] * inv;
                            if (a.bias) {
                                const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + m);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += b[e];
                            }
                            f32x4* d = reinterpret_cast<f32x4*>(orow + m);
                            if (a.accumulate) {
                                const f32x4 o = *d;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += o[e];
                            }
                            *d = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (m + e < a.M) {
                                    float v = acc[mi][ni][4 * rq + e] * inv + (a.bias ? a.bias[m + e] : 0.f);
                                    if (a.accumulate) v += orow[m + e];
                                    orow[m + e] = v;
                                }
                        }
                    }
            }
        }
        if (!has_next) break;
        lb = nlb;
        mb0 = nmb0;
        rb0 = nrb0;
    }
}


// Backward, split mode: one pass over a chunk of logits [rows, V] does what were four -- dlogits in place, two
// fragment conversions, the bias gradient's column sums (2.9 GB of traffic per 3 264 x 32 000 chunk; this pass reads the
// logits once and writes the fragments: 1.25 GB).  Workgroup = 32 rows x 128 vocabulary entries:
//   d = coeff (1[v = target] - exp(logit - lse)) -> LDS (fp32), then
//   B pieces of dlogits   (n = row, k = v):  [rb][kb = v/16][hi|lo][lane = row%32 + 32 (v/8 % 2)]      -> DLb3
//   B pieces of dlogits^T (n = v, k = row):  [vb = v/32][kb = row/16][hi|lo][lane = v%32 + 32 (row/8 % 2)] -> DLb4
//   part[rb][v] = sum of the block's 32 rows (summed over the blocks in order by k_colsum_final)
constexpr int kDlV = 128;
__global__ __launch_bounds__(256) void k_dl_frag(const float* __restrict__ logits, int V, int rows, long row0, int N, int L,
                                                 const int64_t* __restrict__ target, const float* __restrict__ lse,
                                                 const float* __restrict__ coeff, const float* __restrict__ scal,
                                                 f16x8* __restrict__ DLb3, int KB3, f16x8* __restrict__ DLb4, int KB4,
                                                 float* __restrict__ part) {
    __shared__ float tile[32][kDlV + 1];
    const int tid = threadIdx.x;
    const int v0 = blockIdx.x * kDlV, rb = blockIdx.y;
    __shared__ float r_lse[32], r_cf[32];
    __shared__ int r_tg[32];
    if (tid < 32) {
        const int rl = rb * 32 + tid;
        float ls = 0.f, cf = 0.f;
        int tg = -1;
        if (rl < rows) {
            const long r = row0 + rl;
            const long o = (long)(r % N) * L + (r / N);
            ls = lse[o];
            cf = coeff[o];
            tg = (int)target[o];
        }
        r_lse[tid] = ls; r_cf[tid] = cf; r_tg[tid] = tg;
    }
    __syncthreads();
    // element idx = tid + 256 i -> (row idx / 128, column idx % 128): a wave reads 256 contiguous bytes of one row
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i, row = idx >> 7, col = idx & (kDlV - 1);
        const int rl = rb * 32 + row, v = v0 + col;
        float d = 0.f;
        if (rl < rows && v < V)
            d = r_cf[row] * ((v == r_tg[row] ? 1.f : 0.f) - tg_expf(logits[(long)rl * V + v] - r_lse[row]));
        tile[row][col] = d;
    }
    __syncthreads();
    const float sD = scal[3];
    auto split8 = [&](const float (&x)[8], f16x8& h, f16x8& l) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = x[e] * sD;
            const _Float16 hh = (_Float16)y;
            h[e] = hh;
            l[e] = (_Float16)(y - (float)hh);
        }
    };
    // dlogits pieces: 8 k blocks x 64 lanes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + 256 * it, lane = q & 63, kbl = q >> 6;
        const int kb = (v0 >> 4) + kbl;
        if (kb < KB3) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = tile[lane & 31][16 * kbl + 8 * (lane >> 5) + e];
            f16x8 h, l;
            split8(x, h, l);
            f16x8* d = DLb3 + (((size_t)rb * KB3 + kb) * 2) * 64 + lane;
            d[0] = h;
            d[64] = l;
        }
    }
    // dlogits^T pieces: 4 vocabulary blocks x 2 k blocks (16 rows each) x 64 lanes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + 256 * it, lane = q & 63, kbl = (q >> 6) & 1, vbl = q >> 7;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[16 * kbl + 8 * (lane >> 5) + e][32 * vbl + (lane & 31)];
        f16x8 h, l;
        split8(x, h, l);
        const size_t vb = (size_t)(v0 >> 5) + vbl;
        f16x8* d = DLb4 + ((vb * KB4 + 2 * rb + kbl) * 2) * 64 + lane;
        d[0] = h;
        d[64] = l;
    }
    if (tid < kDlV && v0 + tid < V) {
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) sacc += tile[r][tid];
        part[(size_t)rb * V + v0 + tid] = sacc;
    }
}

// out[i] = sum_z part[z*zstride + i], z ascending (fixed order)
__global__ __launch_bounds__(256) void k_sum_slices(const float* part, int nz, long zstride, long n, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = part[i];
        for (int z = 1; z < nz; ++z) s += part[(long)z * zstride + i];
        out[i] = s;
    }
}

// column sums in two deterministic passes: slab y of kSlab rows -> part[y][c]; then the slabs in order
constexpr int kSlab = 256;
__global__ __launch_bounds__(256) void k_colsum_slab(const float* X, long ld, long rows, int ncol, float* part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncol) return;
    const long r0 = (long)blockIdx.y * kSlab, r1 = r0 + kSlab < rows ? r0 + kSlab : rows;
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += X[r * ld + c];
    part[(long)blockIdx.y * ncol + c] = s;
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* part, int nslab, int ncol, float* out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (int y = 0; y < nslab; ++y) s += part[(long)y * ncol + c];
    out[c] = accumulate ? out[c] + s : s;
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline int blocks(long n) { return (int)((n + 255) / 256); }

// rows of logits alive at once: <= 1 GiB, a multiple of 128, at least 128
long chunk_rows(long NL, int V) {
    long rc = ((long)1 << 30) / ((long)V * 4);
    rc = rc / 128 * 128;
    if (rc < 128) rc = 128;
    return rc < NL ? rc : NL;
}

int launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("%s launch: %s", what, hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}


// ---- fragment operands: 32-row blocks padded to whole 256-row tiles, 16-wide k blocks in pairs
inline int fg_rb(long R) { return (int)((R + 255) / 256 * 8); }
inline int fg_kb(long K) { return (int)(2 * ((K + 31) / 32)); }
inline size_t fg_bytes(long R, long K, int extra_rb = 0) { return (size_t)(fg_rb(R) + extra_rb) * fg_kb(K) * 2 * 64 * 16; }

void to_frag(hipStream_t s, const float* X, long rs, long cs, long R, long K, const float* scal, int si, int RBp, int KB,
             void* dst) {
    const long total = (long)RBp * KB * 64;
    const long want = (total + 255) / 256;
    hipLaunchKernelGGL(k_to_frag, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, s, X, rs, cs, (int)R, (int)K,
                       scal, si, RBp, KB, reinterpret_cast<f16x8*>(dst));
}

template <int WM, int WN, int TM, int TN>
int launch_fgemm_t(FgArgs& a, int MBp, hipStream_t s) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr size_t sm = (size_t)2 * (WM * TM + WN * TN) * 256 * 16;
    static TgLdsAttr attr_n, attr_t;
    if (int rc = a.tstore ? tg_lds_attr(attr_t, reinterpret_cast<const void*>(&k_fgemm<WM, WN, TM, TN, true>), sm, "k_fgemm")
                          : tg_lds_attr(attr_n, reinterpret_cast<const void*>(&k_fgemm<WM, WN, TM, TN, false>), sm, "k_fgemm"))
        return rc;
    a.MT = MBp * 32 / BM;
    a.RT = (a.N + BN - 1) / BN;
    a.NB = (a.RT + 7) / 8 * 8 * a.MT;
    const int resident = 256 * (int)((160 * 1024) / sm);
    int gx = resident / a.nz / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > a.NB) gx = a.NB;
    a.gx = gx;
    if (a.tstore) hipLaunchKernelGGL((k_fgemm<WM, WN, TM, TN, true>), dim3(gx * a.nz), dim3(64 * WM * WN), sm, s, a);
    else hipLaunchKernelGGL((k_fgemm<WM, WN, TM, TN, false>), dim3(gx * a.nz), dim3(64 * WM * WN), sm, s, a);
    return TG_OK;
}

// out[n*ldo + m] (+)= scal[inv_idx] * sum_k A(m,k) B(n,k) (+ bias[m])  -- or, tstore, out[m*ldo + n] (+ bias[n]).
// MBp = A's 32-row blocks (a multiple of 8).  zbuf (>= kMaxSlices*N*M floats; dense output, no accumulate): a problem
// with few tiles and a long contraction is cut over k into slices, one workgroup each, summed afterwards in order.
constexpr int kMaxSlices = 16;
int fgemm(hipStream_t s, const void* Af, int MBp, const void* Bf, int KB, float* out, long ldo, int M, int N,
          const float* scal, int inv_idx, const float* bias, bool accumulate, float* zbuf = nullptr, bool tstore = false) {
    if (M <= 0 || N <= 0) return TG_OK;
    FgArgs a = {};
    a.Af = reinterpret_cast<const f32x4*>(Af);
    a.Bf = reinterpret_cast<const f32x4*>(Bf);
    a.KB = KB;
    a.out = out; a.ldo = ldo; a.zstride = 0; a.nz = 1;
    a.M = M; a.N = N;
    a.scal = scal; a.inv_idx = inv_idx; a.bias = bias; a.accumulate = accumulate ? 1 : 0;
    a.tstore = tstore ? 1 : 0;
    a.vec = !tstore && (ldo % 4 == 0) && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
            (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
    // shape and slices: passes over the chip x k steps per workgroup x cost of a step (a 256 x 256 tile moves half the
    // operand bytes per flop: 4 x 0.93 of a 128 x 128 step on a CU of its own; two small workgroups share a CU: 2)
    const int S = KB / 2;
    const bool can_slice = zbuf && !tstore && ldo == M && !accumulate;
    double best = 1e300;
    bool use_big = true;
    int best_nz = 1;
    for (int shape = 0; shape < 2; ++shape) {
        // as launch_fgemm_t lays the grid out: NB logical tiles (row tiles rounded up to 8), gx workgroups per slice
        const int BM = shape == 0 ? 256 : 128;
        const int MT = MBp * 32 / BM, RT = (N + BM - 1) / BM, NB = (RT + 7) / 8 * 8 * MT;
        const int slots = shape == 0 ? 256 : 512;
        const double step = shape == 0 ? 4.0 * 0.93 : 2.0;
        for (int nz = 1; nz <= (can_slice ? kMaxSlices : 1); ++nz) {
            if (nz > 1 && S / nz < 16) break;
            int gx = slots / nz / 8 * 8;
            if (gx < 8) gx = 8;
            if (gx > NB) gx = NB;
            const int passes = (NB / 8 + gx / 8 - 1) / (gx / 8);
            const double cost = (double)passes * ((S + nz - 1) / nz + 4) * step;
            if (cost < best * (shape == 1 ? 0.9 : 1.0)) { best = cost; use_big = shape == 0; best_nz = nz; }   // ties go to the big tile
        }
    }
    if (best_nz > 1) {
        a.nz = best_nz;
        a.out = zbuf;
        a.ldo = M;
        a.zstride = (long)N * M;
        a.vec = (M % 4 == 0) && (reinterpret_cast<uintptr_t>(zbuf) & 15) == 0;
    }
    int rc = use_big ? launch_fgemm_t<2, 4, 4, 2>(a, MBp, s) : launch_fgemm_t<2, 2, 2, 2>(a, MBp, s);
    if (rc) return rc;
    if (a.nz > 1) {
        const long n = (long)N * M;
        hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s,
                           zbuf, a.nz, a.zstride, n, out);
    }
    return TG_OK;
}

// k_gemm32 with the contraction cut over blockIdx.z when the output has too few tiles to fill the chip (the weight
// gradients: [4H, E] from thousands of token rows): slices into zbuf [nz][M][N], summed in order into C (ldc == N).
// k slices gemm_sliced cuts an M x N x K product into (1 = not cut); its zbuf needs sliced_nz * M * N floats
int sliced_nz(long M, long N, long K) {
    const long tiles = ((N + kBN - 1) / kBN) * ((M + kBM - 1) / kBM);
    if (tiles <= 0) return 1;
    long nz = 512 / tiles;
    if (nz > kMaxSlices) nz = kMaxSlices;
    if (nz > K / (8 * kBK)) nz = K / (8 * kBK);
    return nz < 2 ? 1 : (int)nz;
}

template <bool AK, bool BK>
void gemm_sliced(hipStream_t s, const float* A, long lda, const float* B, long ldb, float* C, int M, int N, int K,
                 float* zbuf, const int64_t* a_gather = nullptr, const int64_t* b_gather = nullptr, int gather_max = 0) {
    if (M <= 0 || N <= 0) return;
    int nz = sliced_nz(M, N, K);
    if (nz < 2) { gemm<AK, BK>(s, A, lda, B, ldb, C, N, M, N, K, nullptr, false, a_gather, b_gather, gather_max); return; }
    const int kslice = ((K + nz - 1) / nz + kBK - 1) / kBK * kBK;
    nz = (K + kslice - 1) / kslice;
    GemmP p = {A, B, zbuf, nullptr, a_gather, b_gather, lda, ldb, (long)N, M, N, K, 0, gather_max, kslice, (long)M * N};
    hipLaunchKernelGGL((k_gemm32<AK, BK>), dim3((N + kBN - 1) / kBN, (M + kBM - 1) / kBM, nz), dim3(256), 0, s, p);
    const long n = (long)M * N;
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, zbuf,
                       nz, (long)M * N, n, C);
}

// deterministic column sums of X [rows, ncol] (ld): slabs of kSlab rows, then the slabs in order.  part: >= nslab*ncol
void colsum(hipStream_t s, const float* X, long ld, long rows, int ncol, float* part, float* out, bool accumulate) {
    const int nslab = (int)((rows + kSlab - 1) / kSlab);
    hipLaunchKernelGGL(k_colsum_slab, dim3(blocks(ncol), nslab), dim3(256), 0, s, X, ld, rows, ncol, part);
    hipLaunchKernelGGL(k_colsum_final, dim3(blocks(ncol)), dim3(256), 0, s, part, nslab, ncol, out, accumulate ? 1 : 0);
}

// split-fp16 projection: scratch shared by forward and backward
struct SplitLayout { size_t hdr, Wa1, Hb, total; };
SplitLayout split_fwd_layout(const TgGenWeights* w, long NL) {
    SplitLayout o = {};
    size_t off = 0;
    o.hdr = off; off += al(256);
    o.Wa1 = off; off += al(fg_bytes(w->V, w->H));
    o.Hb = off; off += al(fg_bytes(NL, w->H, 8));
    o.total = off;
    return o;
}
int split_prepare(const TgGenWeights* w, const float* hs, long NL, const float* coeff, char* base, const SplitLayout& lo,
                  hipStream_t s) {
    unsigned* absmax = reinterpret_cast<unsigned*>(base + lo.hdr);
    float* scal = reinterpret_cast<float*>(base + lo.hdr + 64);
    if (hipMemsetAsync(absmax, 0, 64, s) != hipSuccess) {
        tg_set_error("hipMemsetAsync failed");
        return TG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k_absmax_u, dim3(1024), dim3(256), 0, s, w->w_out, (size_t)w->V * w->H, absmax);
    if (coeff) hipLaunchKernelGGL(k_absmax_u, dim3(64), dim3(256), 0, s, coeff, (size_t)NL, absmax + 1);
    hipLaunchKernelGGL(k_pgw_scales, dim3(1), dim3(64), 0, s, absmax, scal);
    to_frag(s, w->w_out, w->H, 1, w->V, w->H, scal, 0, fg_rb(w->V), fg_kb(w->H), base + lo.Wa1);     // A(m = v, k = h)
    to_frag(s, hs, 2 * w->H, 1, NL, w->H, scal, 1, fg_rb(NL) + 8, fg_kb(w->H), base + lo.Hb);           // B(n = row, k = h)
    return TG_OK;
}

}  // namespace

bool tg_pgw_shape_ok(const TgGenWeights* w) {
    return w && w->V >= 1 && w->E >= 32 && w->H >= 32 && w->E <= 1024 && w->H <= 1024 && w->E % 32 == 0 && w->H % 32 == 0;
}

size_t tg_pgw_forward_workspace_bytes(const TgGenWeights* w, int N, int L, bool have_logits, bool have_states, bool split) {
    const long NL = (long)N * L;
    size_t b = al(sizeof(int64_t) * NL) + al(sizeof(float) * (size_t)NL * 4 * w->H);   // token ids, X W_ih^T of every position
    if (!have_logits) b += al(sizeof(float) * (size_t)chunk_rows(NL, w->V) * w->V);
    if (!have_states) b += al(sizeof(float) * (size_t)(L + 1) * N * 2 * w->H);
    if (split && !have_logits) b += split_fwd_layout(w, NL).total;
    return b;
}

int tg_pgw_forward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L, float* logp_out,
                   float* lse_out, float* logits_out, float* states_out, void* workspace, size_t workspace_bytes,
                   hipStream_t s, const float* states_in, bool split) {
    const int V = w->V, E = w->E, H = w->H;
    const long NL = (long)N * L;
    split = split && !logits_out;   // full logits (parity surface) are always exact fp32
    const size_t need = tg_pgw_forward_workspace_bytes(w, N, L, logits_out != nullptr, states_out != nullptr, split);
    if (!workspace || workspace_bytes < need) {
        tg_set_error("generator forward (GEMM path): workspace too small (%zu < %zu)", workspace_bytes, need);
        return TG_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    int64_t* idx = (int64_t*)ws; ws += al(sizeof(int64_t) * NL);
    float* preX = (float*)ws; ws += al(sizeof(float) * (size_t)NL * 4 * H);
    float* lchunk = nullptr;
    if (!logits_out) { lchunk = (float*)ws; ws += al(sizeof(float) * (size_t)chunk_rows(NL, V) * V); }
    float* states = states_out ? states_out : (float*)ws;
    if (!states_out) ws += al(sizeof(float) * (size_t)(L + 1) * N * 2 * H);
    char* split_base = ws;
    tg_prof_begin(TG_KERNEL_PGW_FORWARD, s);
    hipLaunchKernelGGL(k_tok_tn, dim3(blocks(NL)), dim3(256), 0, s, inp, N, L, idx);
    // slot 0: the initial state -- zeros (init_hidden, generator.py:102-109) or the caller's carried (h,c)
    if (states_in)
        hipLaunchKernelGGL(k_copy, dim3(blocks((long)N * 2 * H) < 1024 ? blocks((long)N * 2 * H) : 1024), dim3(256), 0, s,
                           states_in, states, (long)N * 2 * H);
    else
        hipLaunchKernelGGL(k_zero, dim3(blocks((long)N * 2 * H) < 1024 ? blocks((long)N * 2 * H) : 1024), dim3(256), 0, s,
                           states, (long)N * 2 * H);
    // gates = emb[inp[:, t]] . W_ih^T + h_{t-1} . W_hh^T (models/generator.py:47-51): the input side of every position
    // as one GEMM, then one launch per position for the recurrent side and the cell
    gemm<false, false>(s, w->emb, E, w->w_ih, E, preX, 4 * H, (int)NL, 4 * H, E, nullptr, false, idx, nullptr, V);
    const bool small_tiles = (long)(H / 32) * ((N + 31) / 32) < 256;   // 32 x 32 tiles would not fill the chip
    for (int t = 0; t < L; ++t) {
        if (small_tiles)
            hipLaunchKernelGGL(k_lstm_step_fwd<16>, dim3(H / 16, (N + 15) / 16), dim3(kStepThreads), 0, s, preX, w->w_hh,
                               w->b_ih, w->b_hh, states, N, H, t);
        else
            hipLaunchKernelGGL(k_lstm_step_fwd<32>, dim3(H / 32, (N + 31) / 32), dim3(kStepThreads), 0, s, preX, w->w_hh,
                               w->b_ih, w->b_hh, states, N, H, t);
    }
    const bool need_proj = (target && logp_out) || lse_out || logits_out;
    if (need_proj) {
        const long rc = chunk_rows(NL, V);
        const float* hs = states + (long)N * 2 * H;   // row r = t*N + n: h after position t
        const SplitLayout slo = split_fwd_layout(w, NL);
        if (split) {
            if (int rc2 = split_prepare(w, hs, NL, nullptr, split_base, slo, s)) return rc2;
        }
        const float* scal = reinterpret_cast<const float*>(split_base + slo.hdr + 64);
        for (long r0 = 0; r0 < NL; r0 += rc) {
            const int rows = (int)((NL - r0 < rc) ? NL - r0 : rc);
            // logits_out is [N, L, V] (row n*L + t); the chunk rows are (t, n): with logits_out the chunk is one
            // position t at a time so that the rows of a launch are a strided slice of the output
            if (logits_out) break;
            if (split) {
                tg_prof_end(s);
                tg_prof_begin(TG_KERNEL_PGW_PROJ, s);
                // operands swapped (M = rows, N = vocabulary) with the transposed store: a lane's column is a vocabulary entry
                if (int rc2 = fgemm(s, split_base + slo.Hb + (size_t)(r0 / 32) * fg_kb(H) * 2048, fg_rb(rows),
                                    split_base + slo.Wa1, fg_kb(H), lchunk, V, rows, V, scal, 2, w->b_out, false, nullptr,
                                    true)) return rc2;
                tg_prof_end(s);
                tg_prof_begin(TG_KERNEL_PGW_FORWARD, s);
            } else {
                tg_prof_end(s);
                tg_prof_begin(TG_KERNEL_PGW_PROJ, s);
                gemm<false, false>(s, hs + r0 * 2 * H, 2 * H, w->w_out, H, lchunk, V, rows, V, H, w->b_out);
                tg_prof_end(s);
                tg_prof_begin(TG_KERNEL_PGW_FORWARD, s);
            }
            hipLaunchKernelGGL(k_row_softmax, dim3(rows), dim3(256), 0, s, lchunk, (long)V, V, r0, N, L, target, lse_out,
                               logp_out, nullptr, nullptr);
        }
        if (logits_out) {
            for (int t = 0; t < L; ++t) {
                float* out_t = logits_out + (long)t * V;   // row n of this launch lands at logits_out[n, t, :]
                gemm<false, false>(s, hs + (long)t * N * 2 * H, 2 * H, w->w_out, H, out_t, (long)L * V, N, V, H, w->b_out);
                hipLaunchKernelGGL(k_row_softmax, dim3(N), dim3(256), 0, s, out_t, (long)L * V, V, (long)t * N, N, L, target,
                                   lse_out, logp_out, nullptr, nullptr);
            }
        }
    }
    tg_prof_end(s);
    return launch_ok("generator forward (GEMM path)");
}

namespace {
struct BwdLayout { size_t idx, DL, dhproj, DA, P, whhT, dcn, dX, part, gz, split, Wa3, Ha4, DLb3, DLb4, zbuf, total; };
BwdLayout bwd_layout(const TgGenWeights* w, int N, int L, bool split) {
    const long NL = (long)N * L;
    const int H = w->H, E = w->E, V = w->V;
    const long rc = chunk_rows(NL, V);
    BwdLayout o = {};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += al(bytes); return at; };
    o.idx = take(sizeof(int64_t) * NL);
    o.DL = take(sizeof(float) * (size_t)rc * V);
    o.dhproj = take(sizeof(float) * (size_t)NL * H);
    o.DA = take(sizeof(float) * (size_t)NL * 4 * H);
    o.P = take(sizeof(float) * (size_t)NL * 4 * H);      // pre-activations of every position (no bias)
    o.whhT = take(sizeof(float) * (size_t)4 * H * H);    // W_hh^T
    o.dcn = take(sizeof(float) * (size_t)N * H);
    o.dX = take(sizeof(float) * (size_t)NL * E);
    const size_t slabs_dl = (size_t)((rc + (split ? 32 : kSlab) - 1) / (split ? 32 : kSlab)) * V,
                 slabs_da = (size_t)((NL + kSlab - 1) / kSlab) * 4 * H;
    o.part = take(sizeof(float) * (slabs_dl > slabs_da ? slabs_dl : slabs_da));
    {   // k slices of the weight-gradient GEMMs ([4H, E], [4H, H] over NL rows) and of dX [NL, E] (over 4H) when cut
        size_t need = (size_t)sliced_nz(4 * H, E, NL) * 4 * H * E;
        const size_t b = (size_t)sliced_nz(4 * H, H, NL) * 4 * H * H, c = (size_t)sliced_nz(NL, E, 4 * H) * NL * E;
        need = need > b ? need : b;
        need = need > c ? need : c;
        o.gz = take(sizeof(float) * need);
    }
    if (split) {
        o.split = take(split_fwd_layout(w, NL).total);
        o.Wa3 = take(fg_bytes(H, V));      // A(m = h, k = v) = W_out^T
        o.Ha4 = take(fg_bytes(H, rc));     // A(m = h, k = row) = h^T of a chunk
        o.DLb3 = take(fg_bytes(rc, V));    // B(n = row, k = v) = dlogits
        o.DLb4 = take(fg_bytes(V, rc));    // B(n = v, k = row) = dlogits^T
        o.zbuf = take(sizeof(float) * (size_t)kMaxSlices * rc * H);
    }
    o.total = off;
    return o;
}
}  // namespace

size_t tg_pgw_backward_workspace_bytes(const TgGenWeights* w, int N, int L, bool split) {
    return bwd_layout(w, N, L, split).total;
}

int tg_pgw_backward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, const float* coeff,
                    const float* lse, const float* states, int N, int L, int64_t padding_idx, const TgGenGrads* g,
                    void* workspace, size_t workspace_bytes, hipStream_t s, bool split) {
    const int V = w->V, E = w->E, H = w->H;
    const long NL = (long)N * L;
    const BwdLayout lo = bwd_layout(w, N, L, split);
    if (!workspace || workspace_bytes < lo.total) {
        tg_set_error("tg_gen_pg_backward (GEMM path): workspace of %zu bytes needed, %zu given", lo.total, workspace_bytes);
        return TG_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    const long rc = chunk_rows(NL, V);
    int64_t* idx = (int64_t*)(ws + lo.idx);
    float* DL = (float*)(ws + lo.DL);
    float* dhproj = (float*)(ws + lo.dhproj);
    float* DA = (float*)(ws + lo.DA);
    float* P = (float*)(ws + lo.P);
    float* whhT = (float*)(ws + lo.whhT);
    float* dcn = (float*)(ws + lo.dcn);
    float* dX = (float*)(ws + lo.dX);
    float* part = (float*)(ws + lo.part);
    tg_prof_begin(TG_KERNEL_PGW_BACKWARD, s);
    hipLaunchKernelGGL(k_tok_tn, dim3(blocks(NL)), dim3(256), 0, s, inp, N, L, idx);
    const float* hs = states + (long)N * 2 * H;   // row r = t*N + n: h after position t (ld 2H)
    const SplitLayout slo = split_fwd_layout(w, NL);
    char* sbase = ws + lo.split;
    const float* scal = reinterpret_cast<const float*>(sbase + slo.hdr + 64);
    if (split) {
        if (int rc2 = split_prepare(w, hs, NL, coeff, sbase, slo, s)) return rc2;
        to_frag(s, w->w_out, 1, H, H, V, scal, 0, fg_rb(H), fg_kb(V), ws + lo.Wa3);
    }
    // the three vocabulary products run under their own profile id: close / reopen the bracket around them
    auto proj_begin = [&](int id = TG_KERNEL_PGW_PROJ) { tg_prof_end(s); tg_prof_begin(id, s); };
    auto proj_end = [&]() { tg_prof_end(s); tg_prof_begin(TG_KERNEL_PGW_BACKWARD, s); };
    // ---- projection: dh_proj, dW_out, db_out, one chunk of rows at a time
    for (long r0 = 0; r0 < NL; r0 += rc) {
        const int rows = (int)((NL - r0 < rc) ? NL - r0 : rc);
        if (split) {
            proj_begin();
            if (int rc2 = fgemm(s, sbase + slo.Hb + (size_t)(r0 / 32) * fg_kb(H) * 2048, fg_rb(rows), sbase + slo.Wa1, fg_kb(H),
                                DL, V, rows, V, scal, 2, w->b_out, false, nullptr, true)) return rc2;
            proj_end();
        } else {
            proj_begin();
            gemm<false, false>(s, hs + r0 * 2 * H, 2 * H, w->w_out, H, DL, V, rows, V, H, w->b_out);
            proj_end();
        }
        if (split) {
            // dlogits, both fragment layouts and the bias gradient's partial sums in one pass over the logits
            const int nrb = (rows + 31) / 32;
            hipLaunchKernelGGL(k_dl_frag, dim3((V + kDlV - 1) / kDlV, nrb), dim3(256), 0, s, DL, V, rows, r0, N, L, target, lse,
                               coeff, scal, reinterpret_cast<f16x8*>(ws + lo.DLb3), fg_kb(V),
                               reinterpret_cast<f16x8*>(ws + lo.DLb4), fg_kb(rows), part);
            hipLaunchKernelGGL(k_colsum_final, dim3(blocks(V)), dim3(256), 0, s, part, nrb, V, g->b_out, r0 > 0 ? 1 : 0);
            to_frag(s, hs + r0 * 2 * H, 1, 2 * H, H, rows, scal, 1, fg_rb(H), fg_kb(rows), ws + lo.Ha4);
            proj_begin(TG_KERNEL_PGW_PROJ_DH);
            if (int rc2 = fgemm(s, ws + lo.Wa3, fg_rb(H), ws + lo.DLb3, fg_kb(V), dhproj + r0 * H, H, H, rows, scal, 4, nullptr,
                                false, (float*)(ws + lo.zbuf))) return rc2;                                   // dlogits . W_out
            proj_begin(TG_KERNEL_PGW_PROJ_DW);
            if (int rc2 = fgemm(s, ws + lo.Ha4, fg_rb(H), ws + lo.DLb4, fg_kb(rows), g->w_out, H, H, V, scal, 5, nullptr,
                                r0 > 0)) return rc2;                                                           // dlogits^T . h
            proj_end();
        } else {
            hipLaunchKernelGGL(k_row_softmax, dim3(rows), dim3(256), 0, s, DL, (long)V, V, r0, N, L, target, nullptr, nullptr,
                               lse, coeff);
            proj_begin(TG_KERNEL_PGW_PROJ_DH);
            gemm<false, true>(s, DL, V, w->w_out, H, dhproj + r0 * H, H, rows, H, V);                 // dlogits . W_out
            proj_begin(TG_KERNEL_PGW_PROJ_DW);
            gemm<true, true>(s, DL, V, hs + r0 * 2 * H, 2 * H, g->w_out, H, V, H, rows, nullptr, r0 > 0);  // dlogits^T . h
            proj_end();
            colsum(s, DL, (long)V, (long)rows, V, part, g->b_out, r0 > 0);
        }
    }
    // ---- back through time: the pre-activations of every position from two batched GEMMs (all h_{t-1} are saved
    // states), then one launch per position: dh = dh_proj + da_{t+1} W_hh and the cell's gate gradients
    gemm<false, false>(s, w->emb, E, w->w_ih, E, P, 4 * H, (int)NL, 4 * H, E, nullptr, false, idx, nullptr, V);
    gemm<false, false>(s, states, 2 * H, w->w_hh, H, P, 4 * H, (int)NL, 4 * H, H, nullptr, true);
    hipLaunchKernelGGL(k_transpose, dim3(H / 32, 4 * H / 32), dim3(256), 0, s, w->w_hh, 4 * H, H, whhT);
    const bool small_tiles = (long)(H / 32) * ((N + 31) / 32) < 256;
    for (int t = L - 1; t >= 0; --t) {
        if (small_tiles)
            hipLaunchKernelGGL(k_lstm_step_bwd<16>, dim3(H / 16, (N + 15) / 16), dim3(kStepThreads), 0, s, P, whhT, w->b_ih,
                               w->b_hh, states, dhproj, dcn, DA, N, H, t, t == L - 1 ? 1 : 0);
        else
            hipLaunchKernelGGL(k_lstm_step_bwd<32>, dim3(H / 32, (N + 31) / 32), dim3(kStepThreads), 0, s, P, whhT, w->b_ih,
                               w->b_hh, states, dhproj, dcn, DA, N, H, t, t == L - 1 ? 1 : 0);
    }
    // ---- weight gradients from all steps at once
    float* gz = (float*)(ws + lo.gz);
    gemm_sliced<true, true>(s, DA, 4 * H, w->emb, E, g->w_ih, 4 * H, E, (int)NL, gz, nullptr, idx, V);             // DA^T . X
    gemm_sliced<true, true>(s, DA, 4 * H, states, 2 * H, g->w_hh, 4 * H, H, (int)NL, gz);                         // DA^T . H_prev
    colsum(s, DA, (long)4 * H, NL, 4 * H, part, g->b_ih, false);
    hipLaunchKernelGGL(k_copy, dim3(blocks(4 * H)), dim3(256), 0, s, g->b_ih, g->b_hh, (long)4 * H);
    gemm_sliced<false, true>(s, DA, 4 * H, w->w_ih, E, dX, (int)NL, E, 4 * H, gz);                                // DA . W_ih
    hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, s, g->emb, (long)V * E);
    hipLaunchKernelGGL(k_scatter_emb, dim3(blocks(NL * E)), dim3(256), 0, s, dX, idx, NL, E, V, padding_idx, g->emb);
    tg_prof_end(s);
    return launch_ok("tg_gen_pg_backward (GEMM path)");
}

// ---- exports of the fragment GEMM (tg_pgw.h)
int tg_fg_rb(long R) { return fg_rb(R); }
int tg_fg_kb(long K) { return fg_kb(K); }
size_t tg_fg_bytes(long R, long K) { return fg_bytes(R, K); }
void tg_fg_absmax(hipStream_t s, const float* x, size_t n, unsigned* out) {
    hipLaunchKernelGGL(k_absmax_u, dim3(1024), dim3(256), 0, s, x, n, out);
}
void tg_fg_to_frag(hipStream_t s, const float* X, long rs, long cs, long R, long K, const float* scal, int si, void* dst) {
    to_frag(s, X, rs, cs, R, K, scal, si, fg_rb(R), fg_kb(K), dst);
}
int tg_fg_gemm_rows(hipStream_t s, const void* Af, long M, const void* Bf, long N, long K, float* out, long ldo,
                    const float* scal, int inv_idx) {
    return fgemm(s, Af, fg_rb(M), Bf, fg_kb(K), out, ldo, (int)M, (int)N, scal, inv_idx, nullptr, false, nullptr, true);
}

