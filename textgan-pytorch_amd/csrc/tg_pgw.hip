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
    f32x16 acc[2][2] = {};
    f32x4 va[4], vb[4];
    g_fetch<A_KMAJOR>(va, p.A, p.lda, p.a_gather, p.gather_max, m0, p.M, 0, p.K, tid);
    g_fetch<B_KMAJOR>(vb, p.B, p.ldb, p.b_gather, p.gather_max, n0, p.N, 0, p.K, tid);
    for (int k0 = 0; k0 < p.K; k0 += kBK) {
        __syncthreads();
        g_commit<A_KMAJOR>(va, As, tid);
        g_commit<B_KMAJOR>(vb, Bs, tid);
        __syncthreads();
        if (k0 + kBK < p.K) {
            g_fetch<A_KMAJOR>(va, p.A, p.lda, p.a_gather, p.gather_max, m0, p.M, k0 + kBK, p.K, tid);
            g_fetch<B_KMAJOR>(vb, p.B, p.ldb, p.b_gather, p.gather_max, n0, p.N, k0 + kBK, p.K, tid);
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
                float* c = p.C + (long)m * p.ldc + n;
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
    GemmP p = {A, B, C, bias, a_gather, b_gather, lda, ldb, ldc, M, N, K, accumulate ? 1 : 0, gather_max};
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

// nn.LSTM cell (gates i,f,g,o): pre [N,4H] (no bias) + b_ih + b_hh, state (h,c) of slot t -> slot t+1 of
// states [L+1,N,2,H]
__global__ __launch_bounds__(256) void k_cell_fwd(const float* pre, const float* b_ih, const float* b_hh, float* states,
                                                   int N, int H, int t) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)N * H) return;
    const int n = (int)(q / H), u = (int)(q % H);
    const float* g = pre + (long)n * 4 * H;
    const float ig = tg_sigmoid(g[u] + (b_ih[u] + b_hh[u]));
    const float fg = tg_sigmoid(g[H + u] + (b_ih[H + u] + b_hh[H + u]));
    const float gg = tg_tanh(g[2 * H + u] + (b_ih[2 * H + u] + b_hh[2 * H + u]));
    const float og = tg_sigmoid(g[3 * H + u] + (b_ih[3 * H + u] + b_hh[3 * H + u]));
    const float* prev = states + ((long)t * N + n) * 2 * H;
    float* cur = states + ((long)(t + 1) * N + n) * 2 * H;
    const float cn = fg * prev[H + u] + ig * gg;
    cur[H + u] = cn;
    cur[u] = og * tg_tanh(cn);
}

__global__ __launch_bounds__(256) void k_zero(float* p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
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

// out[n] (+)= sum_r X[r*ld + n]: one thread per column, rows split over gridDim.y partial sums are avoided by
// keeping the row loop whole (fixed order; columns are coalesced across the lanes)
__global__ __launch_bounds__(256) void k_colsum(const float* X, long ld, long rows, int ncol, float* out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (long r = 0; r < rows; ++r) s += X[r * ld + c];
    out[c] = accumulate ? out[c] + s : s;
}

// one BPTT step, element-wise part: pre [N,4H] (recomputed, no bias); states slots t (previous) and t+1 (this
// step's); dh = dhproj[t*N + n] + dhn[n]; writes DA[t*N + n][4H], dcn <- dc * f.
__global__ __launch_bounds__(256) void k_bptt_elem(const float* pre, const float* b_ih, const float* b_hh,
                                                    const float* states, const float* dhproj, const float* dhn, float* dcn,
                                                    float* DA, int N, int H, int t, int last) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)N * H) return;
    const int n = (int)(q / H), u = (int)(q % H);
    const float* g = pre + (long)n * 4 * H;
    const float ig = tg_sigmoid(g[u] + (b_ih[u] + b_hh[u]));
    const float fg = tg_sigmoid(g[H + u] + (b_ih[H + u] + b_hh[H + u]));
    const float gg = tg_tanh(g[2 * H + u] + (b_ih[2 * H + u] + b_hh[2 * H + u]));
    const float og = tg_sigmoid(g[3 * H + u] + (b_ih[3 * H + u] + b_hh[3 * H + u]));
    const float cprev = states[((long)t * N + n) * 2 * H + H + u];
    const float ct = states[((long)(t + 1) * N + n) * 2 * H + H + u];
    const float tc = tg_tanh(ct);
    const float dh = dhproj[((long)t * N + n) * H + u] + (last ? 0.f : dhn[(long)n * H + u]);
    const float dc = (last ? 0.f : dcn[(long)n * H + u]) + dh * og * (1.f - tc * tc);
    float* da = DA + ((long)t * N + n) * 4 * H;
    da[u] = dc * gg * ig * (1.f - ig);
    da[H + u] = dc * cprev * fg * (1.f - fg);
    da[2 * H + u] = dc * ig * (1.f - gg * gg);
    da[3 * H + u] = dh * tc * og * (1.f - og);
    dcn[(long)n * H + u] = dc * fg;
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

}  // namespace

bool tg_pgw_shape_ok(const TgGenWeights* w) {
    return w && w->V >= 1 && w->E >= 32 && w->H >= 32 && w->E <= 1024 && w->H <= 1024 && w->E % 32 == 0 && w->H % 32 == 0;
}

size_t tg_pgw_forward_workspace_bytes(const TgGenWeights* w, int N, int L, bool have_logits, bool have_states) {
    const long NL = (long)N * L;
    size_t b = al(sizeof(int64_t) * NL) + al(sizeof(float) * (size_t)N * 4 * w->H);
    if (!have_logits) b += al(sizeof(float) * (size_t)chunk_rows(NL, w->V) * w->V);
    if (!have_states) b += al(sizeof(float) * (size_t)(L + 1) * N * 2 * w->H);
    return b;
}

int tg_pgw_forward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L, float* logp_out,
                   float* lse_out, float* logits_out, float* states_out, void* workspace, size_t workspace_bytes,
                   hipStream_t s, const float* states_in) {
    const int V = w->V, E = w->E, H = w->H;
    const long NL = (long)N * L;
    const size_t need = tg_pgw_forward_workspace_bytes(w, N, L, logits_out != nullptr, states_out != nullptr);
    if (!workspace || workspace_bytes < need) {
        tg_set_error("generator forward (GEMM path): workspace too small (%zu < %zu)", workspace_bytes, need);
        return TG_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    int64_t* idx = (int64_t*)ws; ws += al(sizeof(int64_t) * NL);
    float* pre = (float*)ws; ws += al(sizeof(float) * (size_t)N * 4 * H);
    float* lchunk = nullptr;
    if (!logits_out) { lchunk = (float*)ws; ws += al(sizeof(float) * (size_t)chunk_rows(NL, V) * V); }
    float* states = states_out ? states_out : (float*)ws;
    tg_prof_begin(TG_KERNEL_PGW_FORWARD, s);
    hipLaunchKernelGGL(k_tok_tn, dim3(blocks(NL)), dim3(256), 0, s, inp, N, L, idx);
    // slot 0: the initial state -- zeros (init_hidden, generator.py:102-109) or the caller's carried (h,c)
    if (states_in)
        hipLaunchKernelGGL(k_copy, dim3(blocks((long)N * 2 * H) < 1024 ? blocks((long)N * 2 * H) : 1024), dim3(256), 0, s,
                           states_in, states, (long)N * 2 * H);
    else
        hipLaunchKernelGGL(k_zero, dim3(blocks((long)N * 2 * H) < 1024 ? blocks((long)N * 2 * H) : 1024), dim3(256), 0, s,
                           states, (long)N * 2 * H);
    for (int t = 0; t < L; ++t) {
        // gates = emb[inp[:, t]] . W_ih^T + h_{t-1} . W_hh^T   (models/generator.py:47-51)
        gemm<false, false>(s, w->emb, E, w->w_ih, E, pre, 4 * H, N, 4 * H, E, nullptr, false, idx + (long)t * N, nullptr, V);
        gemm<false, false>(s, states + (long)t * N * 2 * H, 2 * H, w->w_hh, H, pre, 4 * H, N, 4 * H, H, nullptr, true);
        hipLaunchKernelGGL(k_cell_fwd, dim3(blocks((long)N * H)), dim3(256), 0, s, pre, w->b_ih, w->b_hh, states, N, H, t);
    }
    const bool need_proj = (target && logp_out) || lse_out || logits_out;
    if (need_proj) {
        const long rc = chunk_rows(NL, V);
        const float* hs = states + (long)N * 2 * H;   // row r = t*N + n: h after position t
        for (long r0 = 0; r0 < NL; r0 += rc) {
            const int rows = (int)((NL - r0 < rc) ? NL - r0 : rc);
            // logits_out is [N, L, V] (row n*L + t); the chunk rows are (t, n): with logits_out the chunk is one
            // position t at a time so that the rows of a launch are a strided slice of the output
            if (logits_out) break;
            gemm<false, false>(s, hs + r0 * 2 * H, 2 * H, w->w_out, H, lchunk, V, rows, V, H, w->b_out);
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

size_t tg_pgw_backward_workspace_bytes(const TgGenWeights* w, int N, int L) {
    const long NL = (long)N * L;
    const int H = w->H, E = w->E;
    return al(sizeof(int64_t) * NL) + al(sizeof(float) * (size_t)chunk_rows(NL, w->V) * w->V) +
           al(sizeof(float) * (size_t)NL * H) + al(sizeof(float) * (size_t)NL * 4 * H) +
           al(sizeof(float) * (size_t)N * 4 * H) + 2 * al(sizeof(float) * (size_t)N * H) +
           al(sizeof(float) * (size_t)NL * E);
}

int tg_pgw_backward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, const float* coeff,
                    const float* lse, const float* states, int N, int L, int64_t padding_idx, const TgGenGrads* g,
                    void* workspace, size_t workspace_bytes, hipStream_t s) {
    const int V = w->V, E = w->E, H = w->H;
    const long NL = (long)N * L;
    const size_t need = tg_pgw_backward_workspace_bytes(w, N, L);
    if (!workspace || workspace_bytes < need) {
        tg_set_error("tg_gen_pg_backward (GEMM path): workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        return TG_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    const long rc = chunk_rows(NL, V);
    int64_t* idx = (int64_t*)ws; ws += al(sizeof(int64_t) * NL);
    float* DL = (float*)ws; ws += al(sizeof(float) * (size_t)rc * V);
    float* dhproj = (float*)ws; ws += al(sizeof(float) * (size_t)NL * H);
    float* DA = (float*)ws; ws += al(sizeof(float) * (size_t)NL * 4 * H);
    float* pre = (float*)ws; ws += al(sizeof(float) * (size_t)N * 4 * H);
    float* dhn = (float*)ws; ws += al(sizeof(float) * (size_t)N * H);
    float* dcn = (float*)ws; ws += al(sizeof(float) * (size_t)N * H);
    float* dX = (float*)ws;
    tg_prof_begin(TG_KERNEL_PGW_BACKWARD, s);
    hipLaunchKernelGGL(k_tok_tn, dim3(blocks(NL)), dim3(256), 0, s, inp, N, L, idx);
    const float* hs = states + (long)N * 2 * H;   // row r = t*N + n: h after position t (ld 2H)
    // ---- projection: dh_proj, dW_out, db_out, one chunk of rows at a time
    for (long r0 = 0; r0 < NL; r0 += rc) {
        const int rows = (int)((NL - r0 < rc) ? NL - r0 : rc);
        gemm<false, false>(s, hs + r0 * 2 * H, 2 * H, w->w_out, H, DL, V, rows, V, H, w->b_out);
        hipLaunchKernelGGL(k_row_softmax, dim3(rows), dim3(256), 0, s, DL, (long)V, V, r0, N, L, target, nullptr, nullptr,
                           lse, coeff);
        gemm<false, true>(s, DL, V, w->w_out, H, dhproj + r0 * H, H, rows, H, V);                 // dlogits . W_out
        gemm<true, true>(s, DL, V, hs + r0 * 2 * H, 2 * H, g->w_out, H, V, H, rows, nullptr, r0 > 0);  // dlogits^T . h
        hipLaunchKernelGGL(k_colsum, dim3(blocks(V)), dim3(256), 0, s, DL, (long)V, (long)rows, V, g->b_out, r0 > 0 ? 1 : 0);
    }
    // ---- back through time
    for (int t = L - 1; t >= 0; --t) {
        gemm<false, false>(s, w->emb, E, w->w_ih, E, pre, 4 * H, N, 4 * H, E, nullptr, false, idx + (long)t * N, nullptr, V);
        gemm<false, false>(s, states + (long)t * N * 2 * H, 2 * H, w->w_hh, H, pre, 4 * H, N, 4 * H, H, nullptr, true);
        hipLaunchKernelGGL(k_bptt_elem, dim3(blocks((long)N * H)), dim3(256), 0, s, pre, w->b_ih, w->b_hh, states, dhproj,
                           dhn, dcn, DA, N, H, t, t == L - 1 ? 1 : 0);
        if (t > 0) gemm<false, true>(s, DA + (long)t * N * 4 * H, 4 * H, w->w_hh, H, dhn, H, N, H, 4 * H);  // da . W_hh
    }
    // ---- weight gradients from all steps at once
    gemm<true, true>(s, DA, 4 * H, w->emb, E, g->w_ih, E, 4 * H, E, (int)NL, nullptr, false, nullptr, idx, V);   // DA^T . X
    gemm<true, true>(s, DA, 4 * H, states, 2 * H, g->w_hh, H, 4 * H, H, (int)NL);                                // DA^T . H_prev
    hipLaunchKernelGGL(k_colsum, dim3(blocks(4 * H)), dim3(256), 0, s, DA, (long)4 * H, NL, 4 * H, g->b_ih, 0);
    hipLaunchKernelGGL(k_copy, dim3(blocks(4 * H)), dim3(256), 0, s, g->b_ih, g->b_hh, (long)4 * H);
    gemm<false, true>(s, DA, 4 * H, w->w_ih, E, dX, E, (int)NL, E, 4 * H);                                       // DA . W_ih
    hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, s, g->emb, (long)V * E);
    hipLaunchKernelGGL(k_scatter_emb, dim3(blocks(NL * E)), dim3(256), 0, s, dX, idx, NL, E, V, padding_idx, g->emb);
    tg_prof_end(s);
    return launch_ok("tg_gen_pg_backward (GEMM path)");
}
