/*
 * textgan_oracle.c -- CPU restatement of TextGAN-PyTorch's Monte-Carlo rollout
 * reward path.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * the timed CPU baseline.  The product path (textgan-pytorch_amd/) never links,
 * loads or calls anything in this directory.
 *
 * Parity status: PINNED.  tools/make_goldens.py imports the real reference
 * (/root/reference, in the build container only), runs it on seeded inputs with
 * torch.multinomial replaced by orc_sample_from_probs (the canonical sampler,
 * below) and commits inputs' seeds + outputs under tests/golden/;
 * tests/test_oracle_golden.py checks every function here against them.
 *
 * Every function cites the reference file:line (relative to the reference
 * tree) it restates.  All arithmetic is fp32, tokens are int64, arrays are
 * row-major.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp, x86-64-v3).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_FILTERS 16
#define ORC_NSPAN 8   /* canonical sampler: vocab spans per row */
#define ORC_TILE 16   /* canonical sampler: elements per tile   */

typedef struct OrcGenWeights {
    const float* emb;   /* [V,E]  */
    const float* w_ih;  /* [4H,E] */
    const float* w_hh;  /* [4H,H] */
    const float* b_ih;  /* [4H]   */
    const float* b_hh;  /* [4H]   */
    const float* w_out; /* [V,H]  */
    const float* b_out; /* [V]    */
    int32_t V, E, H;
    int32_t _pad;
} OrcGenWeights;

typedef struct OrcDisWeights {
    const float* emb;
    const float* conv_w[ORC_MAX_FILTERS];
    const float* conv_b[ORC_MAX_FILTERS];
    const float* hw_w;
    const float* hw_b;
    const float* out_w;
    const float* out_b;
    int32_t f[ORC_MAX_FILTERS];
    int32_t n[ORC_MAX_FILTERS];
    int32_t n_filters;
    int32_t V, Ed, F, n_out;
    int32_t _pad;
} OrcDisWeights;

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* Canonical sampler geometry (shared definition with the HIP kernels,
 * DESIGN.md "canonical sampler"): the vocabulary is cut into ORC_NSPAN
 * contiguous spans of span_len = 16*ceil(V/128) entries; each span into groups
 * of group_tiles tiles of 16 entries, group_tiles = smallest power of two (at
 * most 8) with ceil(span_len/16/group_tiles) <= 16. */
int orc_span_len(int V) { return ORC_TILE * ((V + ORC_NSPAN * ORC_TILE - 1) / (ORC_NSPAN * ORC_TILE)); }
int orc_group_tiles(int V) {
    int nt = orc_span_len(V) / ORC_TILE, gk = 1;
    while ((nt + gk - 1) / gk > 16 && gk < 8) gk *= 2;
    return gk;
}

/* fp32 sequential sum of one 16-entry tile starting at `base` (entries >= hi
 * count as 0). */
static inline float tile_sum(const float* w, int base, int hi) {
    float ts = 0.f;
    for (int r = 0; r < ORC_TILE; ++r) ts += (base + r < hi) ? w[base + r] : 0.f;
    return ts;
}
/* fp32 sequential sum of the tile sums of group q of span sp. */
static inline float group_sum(const float* w, int V, int Ls, int gk, int sp, int q) {
    float c = 0.f;
    for (int t = 0; t < gk; ++t) {
        const int off = (q * gk + t) * ORC_TILE;
        if (off >= Ls) break;
        c += tile_sum(w, sp * Ls + off, V);
    }
    return c;
}

/* Canonical inverse-CDF draw from non-negative weights w[0..V) (need not be
 * normalised) and a uniform u in [0,1).  Replaces
 * torch.multinomial(torch.exp(out), 1) (utils/rollout.py:49,
 * models/generator.py:79).  Hierarchical:
 *   tile sum   = fp32 sequential sum of 16 entries
 *   group sum  = fp32 sequential sum of its tile sums
 *   span sum   = fp64 sequential sum of its group sums
 *   total S    = fp64 sequential sum of the 8 span sums;  target = (double)u*S
 *   span   = first span whose inclusive fp64 prefix exceeds target (else the
 *            last non-empty span);
 *   group  = first group of it whose inclusive fp64 prefix exceeds target
 *            (else its last non-empty group);
 *   tile   = first tile of the group whose inclusive fp64 prefix exceeds target
 *            (else its last non-empty tile);
 *   token  = first entry of the tile with prefix + (double)(fp32 running sum
 *            inside the tile) > target (else the tile's last valid entry).
 * Everything above tile level is fp64 so that the only source of disagreement
 * between two implementations is the rounding of the weights themselves. */
static int64_t canonical_draw(const float* w, int V, float u) {
    const int Ls = orc_span_len(V), gk = orc_group_tiles(V);
    const int gs = gk * ORC_TILE;             /* entries per group */
    const int ng = (Ls + gs - 1) / gs;        /* groups per span   */
    double span_sum[ORC_NSPAN];
    int last_nonempty = 0;
    for (int sp = 0; sp < ORC_NSPAN; ++sp) {
        double s = 0.0;
        for (int q = 0; q < ng; ++q) s += (double)group_sum(w, V, Ls, gk, sp, q);
        span_sum[sp] = s;
        if (sp * Ls < V) last_nonempty = sp;
    }
    double S = 0.0;
    for (int sp = 0; sp < ORC_NSPAN; ++sp) S += span_sum[sp];
    const double target = (double)u * S;
    /* span select */
    int sel = last_nonempty;
    double p = 0.0;
    for (int sp = 0; sp < ORC_NSPAN; ++sp) {
        if (p + span_sum[sp] > target || sp == last_nonempty) { sel = sp; break; }
        p += span_sum[sp];
    }
    /* group select */
    const int span_lo = sel * Ls;
    int span_hi = span_lo + Ls; if (span_hi > V) span_hi = V;
    const int ng_valid = (span_hi - span_lo + gs - 1) / gs;
    int gq = ng_valid - 1;
    for (int q = 0; q < ng_valid; ++q) {
        const double c = (double)group_sum(w, V, Ls, gk, sel, q);
        if (p + c > target || q == ng_valid - 1) { gq = q; break; }
        p += c;
    }
    /* tile select */
    const int g_lo = span_lo + gq * gs;
    int g_hi = g_lo + gs; if (g_hi > span_hi) g_hi = span_hi;
    const int nt_valid = (g_hi - g_lo + ORC_TILE - 1) / ORC_TILE;
    int tq = nt_valid - 1;
    for (int t = 0; t < nt_valid; ++t) {
        const double ts = (double)tile_sum(w, g_lo + t * ORC_TILE, V);
        if (p + ts > target || t == nt_valid - 1) { tq = t; break; }
        p += ts;
    }
    /* element select */
    const int t_lo = g_lo + tq * ORC_TILE;
    int t_hi = t_lo + ORC_TILE; if (t_hi > g_hi) t_hi = g_hi;
    float pp = 0.f;
    for (int j = t_lo; j < t_hi; ++j) {
        pp += w[j];
        if (p + (double)pp > target) return j;
    }
    return t_hi - 1;
}

/* Batched canonical draw on a probability matrix: the function the golden
 * harness installs in place of torch.multinomial(p, 1). */
void orc_sample_from_probs(const float* p, const float* u, int N, int V, int64_t* tok) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) tok[n] = canonical_draw(p + (size_t)n * V, V, u[n]);
}

/* ------------------------------------------------------------------------ */
/* Generator                                                                 */

static inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

/* Transposed copies so the inner loops run over the output index (vectorises
 * without reassociating the k-sums). */
typedef struct GenT {
    float* wih_t;  /* [E][4H] */
    float* whh_t;  /* [H][4H] */
    float* bsum;   /* [4H]  b_ih + b_hh */
    float* wout_t; /* [H][V] */
} GenT;

static void gen_t_init(GenT* t, const OrcGenWeights* w) {
    const int V = w->V, E = w->E, H = w->H, G = 4 * H;
    t->wih_t = (float*)malloc(sizeof(float) * E * G);
    t->whh_t = (float*)malloc(sizeof(float) * H * G);
    t->bsum = (float*)malloc(sizeof(float) * G);
    t->wout_t = (float*)malloc(sizeof(float) * (size_t)H * V);
    for (int r = 0; r < G; ++r) {
        for (int k = 0; k < E; ++k) t->wih_t[k * G + r] = w->w_ih[r * E + k];
        for (int k = 0; k < H; ++k) t->whh_t[k * G + r] = w->w_hh[r * H + k];
        t->bsum[r] = w->b_ih[r] + w->b_hh[r];
    }
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < H; ++k) t->wout_t[(size_t)k * V + v] = w->w_out[(size_t)v * H + k];
}
static void gen_t_free(GenT* t) { free(t->wih_t); free(t->whh_t); free(t->bsum); free(t->wout_t); }

/* One LSTM step for one row: nn.Embedding gather + nn.LSTM cell
 * (models/generator.py:47,51; gates i,f,g,o; b_ih + b_hh).  gates = scratch[4H]. */
static void lstm_cell(const OrcGenWeights* w, const GenT* t, int64_t tok, float* h, float* c, float* gates) {
    const int E = w->E, H = w->H, G = 4 * H;
    const float* x = w->emb + (size_t)tok * E;
    for (int r = 0; r < G; ++r) gates[r] = t->bsum[r];
    for (int k = 0; k < E; ++k) {
        const float xv = x[k];
        const float* wr = t->wih_t + (size_t)k * G;
        for (int r = 0; r < G; ++r) gates[r] += wr[r] * xv;
    }
    for (int k = 0; k < H; ++k) {
        const float hv = h[k];
        const float* wr = t->whh_t + (size_t)k * G;
        for (int r = 0; r < G; ++r) gates[r] += wr[r] * hv;
    }
    for (int j = 0; j < H; ++j) {
        const float ig = sigmoidf_(gates[j]);
        const float fg = sigmoidf_(gates[H + j]);
        const float gg = tanhf(gates[2 * H + j]);
        const float og = sigmoidf_(gates[3 * H + j]);
        const float cn = fg * c[j] + ig * gg;
        c[j] = cn;
        h[j] = og * tanhf(cn);
    }
}

/* nn.Linear(H,V) (models/generator.py:53): logits[v] = b_out[v] + sum_k h[k] W_out[v][k]. */
static void project(const OrcGenWeights* w, const GenT* t, const float* h, float* logits) {
    const int V = w->V, H = w->H;
    for (int v = 0; v < V; ++v) logits[v] = w->b_out[v];
    for (int k = 0; k < H; ++k) {
        const float hv = h[k];
        const float* wr = t->wout_t + (size_t)k * V;
        for (int v = 0; v < V; ++v) logits[v] += wr[v] * hv;
    }
}

/* softmax weights relative to the row max (LogSoftmax + exp,
 * models/generator.py:55 + utils/rollout.py:49): wts[v] = expf(l[v]-M);
 * returns M. */
static float exp_weights(const float* logits, int V, float* wts) {
    float M = logits[0];
    for (int v = 1; v < V; ++v) M = logits[v] > M ? logits[v] : M;
    for (int v = 0; v < V; ++v) wts[v] = expf(logits[v] - M);
    return M;
}

/* Canonical total of the weights (same summation tree as canonical_draw). */
static double canonical_total(const float* w, int V) {
    const int Ls = orc_span_len(V), gk = orc_group_tiles(V);
    const int gs = gk * ORC_TILE, ng = (Ls + gs - 1) / gs;
    double S = 0.0;
    for (int sp = 0; sp < ORC_NSPAN; ++sp) {
        double s = 0.0;
        for (int q = 0; q < ng; ++q) s += (double)group_sum(w, V, Ls, gk, sp, q);
        S += s;
    }
    return S;
}

/* LSTMGenerator.forward over [N,L] with zero initial state
 * (models/generator.py:40-60, init_hidden :102-109), the NLL pick
 * (metrics/nll.py:59-61) and the rollout prefix state (utils/rollout.py:35-38).
 * Any of logp_out [N,L], lse_out [N,L], logits_out [N,L,V],
 * states_out [L+1,N,2,H] may be NULL. */
void orc_gen_teacher_forced(const OrcGenWeights* w, const int64_t* inp, const int64_t* target,
                            int N, int L, float* logp_out, float* lse_out, float* logits_out,
                            float* states_out) {
    GenT t; gen_t_init(&t, w);
    const int V = w->V, H = w->H;
    const int need_proj = (target && logp_out) || lse_out || logits_out;
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * H);
        float* c = (float*)malloc(sizeof(float) * H);
        float* gates = (float*)malloc(sizeof(float) * 4 * H);
        float* logits = (float*)malloc(sizeof(float) * V);
        float* wts = (float*)malloc(sizeof(float) * V);
#pragma omp for schedule(dynamic, 4)
        for (int n = 0; n < N; ++n) {
            memset(h, 0, sizeof(float) * H);
            memset(c, 0, sizeof(float) * H);
            if (states_out) {
                float* s0 = states_out + ((size_t)0 * N + n) * 2 * H;
                memset(s0, 0, sizeof(float) * 2 * H);
            }
            for (int s = 0; s < L; ++s) {
                lstm_cell(w, &t, inp[(size_t)n * L + s], h, c, gates);
                if (states_out) {
                    float* st = states_out + ((size_t)(s + 1) * N + n) * 2 * H;
                    memcpy(st, h, sizeof(float) * H);
                    memcpy(st + H, c, sizeof(float) * H);
                }
                if (!need_proj) continue;
                project(w, &t, h, logits);
                if (logits_out) memcpy(logits_out + ((size_t)n * L + s) * V, logits, sizeof(float) * V);
                const float M = exp_weights(logits, V, wts);
                const float lse = (float)((double)M + log(canonical_total(wts, V)));
                if (lse_out) lse_out[(size_t)n * L + s] = lse;
                if (target && logp_out) logp_out[(size_t)n * L + s] = logits[target[(size_t)n * L + s]] - lse;
            }
        }
        free(h); free(c); free(gates); free(logits); free(wts);
    }
    gen_t_free(&t);
}

/* ROLLOUT.rollout_mc_search for all (i, given_num) of get_reward's loop
 * (utils/rollout.py:25-55, 143-145).  The prefix state after g tokens is
 * computed once per sentence (the reference recomputes it per call with the
 * same deterministic arithmetic).  uniforms [R,L,L,B]; completions [R,L,B,L];
 * forced [R,L,B,L] or NULL = replay mode: the drawn token is still what is
 * written, but the token fed to the next step is forced[...] (step-wise
 * agreement measurement that does not cascade). */
void orc_gen_rollout(const OrcGenWeights* w, const int64_t* sentences, int B, int L, int R,
                     int i_lo, int i_hi, const float* uniforms, const int64_t* forced,
                     int64_t* completions) {
    GenT t; gen_t_init(&t, w);
    const int V = w->V, H = w->H;
    float* states = (float*)malloc(sizeof(float) * (size_t)(L + 1) * B * 2 * H);
    /* prefix states: rollout.py:35-38 (no start letter) */
#pragma omp parallel
    {
        float* gates = (float*)malloc(sizeof(float) * 4 * H);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            float* s0 = states + ((size_t)0 * B + b) * 2 * H;
            memset(s0, 0, sizeof(float) * 2 * H);
            for (int s = 0; s < L; ++s) {
                float* prev = states + ((size_t)s * B + b) * 2 * H;
                float* cur = states + ((size_t)(s + 1) * B + b) * 2 * H;
                memcpy(cur, prev, sizeof(float) * 2 * H);
                lstm_cell(w, &t, sentences[(size_t)b * L + s], cur, cur + H, gates);
            }
        }
        free(gates);
    }
    const long n_items = (long)(i_hi - i_lo) * L * B;
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * H);
        float* c = (float*)malloc(sizeof(float) * H);
        float* gates = (float*)malloc(sizeof(float) * 4 * H);
        float* logits = (float*)malloc(sizeof(float) * V);
        float* wts = (float*)malloc(sizeof(float) * V);
#pragma omp for schedule(dynamic, 8)
        for (long it = 0; it < n_items; ++it) {
            const int b = (int)(it % B);
            const int g = (int)((it / B) % L) + 1;
            const int i = (int)(it / ((long)B * L)) + i_lo;
            int64_t* out = completions + (((size_t)i * L + (g - 1)) * B + b) * L;
            for (int s = 0; s < g; ++s) out[s] = sentences[(size_t)b * L + s];
            if (g == L) continue;
            const float* st = states + ((size_t)g * B + b) * 2 * H;
            memcpy(h, st, sizeof(float) * H);
            memcpy(c, st + H, sizeof(float) * H);
            for (int s = g; s < L; ++s) {
                project(w, &t, h, logits);
                exp_weights(logits, V, wts);
                const float u = uniforms[(((size_t)i * L + (g - 1)) * L + s) * B + b];
                const int64_t tok = canonical_draw(wts, V, u);
                out[s] = tok;
                /* replay mode (parity tests): feed back the given token instead */
                const int64_t fed = forced ? forced[(((size_t)i * L + (g - 1)) * B + b) * L + s] : tok;
                if (s + 1 < L) lstm_cell(w, &t, fed, h, c, gates);
            }
        }
        free(h); free(c); free(gates); free(logits); free(wts);
    }
    free(states);
    gen_t_free(&t);
}

/* LSTMGenerator.sample step loop (models/generator.py:71-81) for N rows:
 * zero state, first input start_letter.  uniforms [L,N]; out [N,L]. */
void orc_gen_sample(const OrcGenWeights* w, int N, int L, int64_t start_letter,
                    const float* uniforms, const int64_t* forced, int64_t* out) {
    GenT t; gen_t_init(&t, w);
    const int V = w->V, H = w->H;
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * H);
        float* c = (float*)malloc(sizeof(float) * H);
        float* gates = (float*)malloc(sizeof(float) * 4 * H);
        float* logits = (float*)malloc(sizeof(float) * V);
        float* wts = (float*)malloc(sizeof(float) * V);
#pragma omp for schedule(dynamic, 4)
        for (int n = 0; n < N; ++n) {
            memset(h, 0, sizeof(float) * H);
            memset(c, 0, sizeof(float) * H);
            int64_t tok = start_letter;
            for (int s = 0; s < L; ++s) {
                lstm_cell(w, &t, tok, h, c, gates);
                project(w, &t, h, logits);
                exp_weights(logits, V, wts);
                tok = canonical_draw(wts, V, uniforms[(size_t)s * N + n]);
                out[(size_t)n * L + s] = tok;
                if (forced) tok = forced[(size_t)n * L + s];
            }
        }
        free(h); free(c); free(gates); free(logits); free(wts);
    }
    gen_t_free(&t);
}

/* ------------------------------------------------------------------------ */
/* Discriminator                                                             */

/* CNNDiscriminator.forward / get_feature (models/discriminator.py:40-64):
 * embedding gather -> per width k: Conv2d(1,n_k,(f_k,Ed)) + ReLU + max over
 * time -> concat -> highway -> dropout -> Linear(F,n_out).
 * dropout: keep = u < 1-p, scaled by 1/(1-p) (nn.Dropout; the mask comes from
 * dropout_uniforms [N,F]; NULL or p == 0 = identity / eval mode).
 * Outputs (any may be NULL): logits_out [N,n_out], feature_out [N,F] (the
 * get_feature value, before dropout), reward_out [N] = softmax(logits)[col]
 * (utils/rollout.py:147-148).  Returns -1 if L < max filter width (the
 * reference raises RuntimeError from conv2d). */

/* Transposed scorer weights (conv [f*Ed][n_k], highway [F_in][F_out]) so the inner loops run over the output
 * index and vectorise without reassociating the k-sums. */
typedef struct DisT {
    float* cw_t[ORC_MAX_FILTERS];
    float* hw_t;
} DisT;

static void dis_t_init(DisT* t, const OrcDisWeights* w) {
    const int Ed = w->Ed, F = w->F, nf = w->n_filters;
    for (int k = 0; k < nf; ++k) {
        const int K = w->f[k] * Ed, n = w->n[k];
        t->cw_t[k] = (float*)malloc(sizeof(float) * (size_t)K * n);
        for (int m = 0; m < n; ++m)
            for (int kk = 0; kk < K; ++kk) t->cw_t[k][(size_t)kk * n + m] = w->conv_w[k][(size_t)m * K + kk];
    }
    t->hw_t = (float*)malloc(sizeof(float) * (size_t)F * F);
    for (int o = 0; o < F; ++o)
        for (int i = 0; i < F; ++i) t->hw_t[(size_t)i * F + o] = w->hw_w[(size_t)o * F + i];
}
static void dis_t_free(DisT* t, const OrcDisWeights* w) {
    for (int k = 0; k < w->n_filters; ++k) free(t->cw_t[k]);
    free(t->hw_t);
}

/* CNNDiscriminator.get_feature for one sequence (models/discriminator.py:51-64): embedding gather, per width
 * Conv2d + ReLU + max over time, concat, highway.  e [L*Ed], pooled [F], acc [1024] are scratch; hwy [F] is the
 * feature (before dropout). */
static void dis_feature_row(const OrcDisWeights* w, const DisT* t, const int64_t* seq, int L, float* e, float* pooled,
                            float* acc, float* hwy) {
    const int Ed = w->Ed, F = w->F, nf = w->n_filters;
    for (int s = 0; s < L; ++s) memcpy(e + (size_t)s * Ed, w->emb + (size_t)seq[s] * Ed, sizeof(float) * Ed);
    int off = 0;
    for (int k = 0; k < nf; ++k) {
        const int f = w->f[k], nk = w->n[k], K = f * Ed, T = L - f + 1;
        for (int m = 0; m < nk; ++m) pooled[off + m] = 0.f; /* relu floor */
        for (int tt = 0; tt < T; ++tt) {
            const float* x = e + (size_t)tt * Ed; /* window = K contiguous floats */
            for (int m = 0; m < nk; ++m) acc[m] = w->conv_b[k][m];
            for (int kk = 0; kk < K; ++kk) {
                const float xv = x[kk];
                const float* wr = t->cw_t[k] + (size_t)kk * nk;
                for (int m = 0; m < nk; ++m) acc[m] += wr[m] * xv;
            }
            for (int m = 0; m < nk; ++m) pooled[off + m] = acc[m] > pooled[off + m] ? acc[m] : pooled[off + m];
        }
        off += nk;
    }
    /* highway: sigmoid(h)*relu(h) + (1-sigmoid(h))*pred  (discriminator.py:61-62) */
    for (int o = 0; o < F; ++o) hwy[o] = w->hw_b[o];
    for (int i = 0; i < F; ++i) {
        const float pv = pooled[i];
        const float* wr = t->hw_t + (size_t)i * F;
        for (int o = 0; o < F; ++o) hwy[o] += wr[o] * pv;
    }
    for (int o = 0; o < F; ++o) {
        const float sg = sigmoidf_(hwy[o]);
        const float rl = hwy[o] > 0.f ? hwy[o] : 0.f;
        hwy[o] = sg * rl + (1.f - sg) * pooled[o];
    }
}

int orc_dis_forward(const OrcDisWeights* w, const int64_t* seqs, int N, int L, int col,
                    float dropout_p, const float* dropout_uniforms,
                    float* reward_out, float* logits_out, float* feature_out) {
    const int Ed = w->Ed, F = w->F, nf = w->n_filters, n_out = w->n_out;
    for (int k = 0; k < nf; ++k) if (w->f[k] > L) return -1;
    DisT t; dis_t_init(&t, w);
    const float keep_scale = (dropout_p > 0.f) ? 1.f / (1.f - dropout_p) : 1.f;
#pragma omp parallel
    {
        float* e = (float*)malloc(sizeof(float) * (size_t)L * Ed);
        float* pooled = (float*)malloc(sizeof(float) * F);
        float* hwy = (float*)malloc(sizeof(float) * F);
        float* acc = (float*)malloc(sizeof(float) * 1024);
#pragma omp for schedule(dynamic, 8)
        for (int n = 0; n < N; ++n) {
            dis_feature_row(w, &t, seqs + (size_t)n * L, L, e, pooled, acc, hwy);
            if (feature_out) memcpy(feature_out + (size_t)n * F, hwy, sizeof(float) * F);
            if (dropout_p > 0.f && dropout_uniforms) {
                const float* du = dropout_uniforms + (size_t)n * F;
                for (int o = 0; o < F; ++o) hwy[o] = (du[o] < 1.f - dropout_p) ? hwy[o] * keep_scale : 0.f;
            }
            float lg[64];
            for (int j = 0; j < n_out; ++j) {
                float a = w->out_b[j];
                const float* wr = w->out_w + (size_t)j * F;
                for (int o = 0; o < F; ++o) a += wr[o] * hwy[o];
                lg[j] = a;
                if (logits_out) logits_out[(size_t)n * n_out + j] = a;
            }
            if (reward_out) {
                float M = lg[0];
                for (int j = 1; j < n_out; ++j) M = lg[j] > M ? lg[j] : M;
                float S = 0.f;
                for (int j = 0; j < n_out; ++j) S += expf(lg[j] - M);
                reward_out[n] = expf(lg[col] - M) / S;
            }
        }
        free(e); free(pooled); free(hwy); free(acc);
    }
    dis_t_free(&t, w);
    return 0;
}

/* torch.mean(rewards.view(B, L, R), dim=-1) on the flat [R*L, B] buffer
 * (utils/rollout.py:153): out[b,t] = mean_r raw_flat[(b*L+t)*R + r]. */
void orc_reward_reduce(const float* raw, int B, int L, int R, float* out) {
    for (int bt = 0; bt < B * L; ++bt) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += raw[(size_t)bt * R + r];
        out[bt] = s / (float)R;
    }
}

/* ROLLOUT.get_reward (utils/rollout.py:128-154) end to end, rollouts
 * [i_lo,i_hi) only (rows of raw outside that range are left untouched).
 * raw [R*L,B]; completions scratch [R,L,B,L]; dropout_uniforms [R*L*B, F] or
 * NULL.  If out != NULL the reduce is applied (requires the full range). */
int orc_get_reward(const OrcGenWeights* g, const OrcDisWeights* d, const int64_t* sentences,
                   int B, int L, int R, int i_lo, int i_hi, int col, const float* uniforms,
                   float dropout_p, const float* dropout_uniforms,
                   int64_t* completions, float* raw, float* out) {
    orc_gen_rollout(g, sentences, B, L, R, i_lo, i_hi, uniforms, NULL, completions);
    const size_t row0 = (size_t)i_lo * L * B;
    const int n = (i_hi - i_lo) * L * B;
    int rc = orc_dis_forward(d, completions + row0 * L, n, L, col, dropout_p,
                             dropout_uniforms ? dropout_uniforms + row0 * d->F : NULL,
                             raw + row0, NULL, NULL);
    if (rc) return rc;
    if (out) orc_reward_reduce(raw, B, L, R, out);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* LeakGAN rollout                                         */

/* LeakGAN_G parameters (models/LeakGAN_G.py:40-48; state_dict keys embeddings.weight, worker.*_l0, manager.*_l0,
 * work2goal.*, mana2goal.*, goal2goal.weight, goal_init).  G = goal_out_size = the scorer's feature width F,
 * K = goal_size. */
typedef struct OrcLeakWeights {
    const float* emb;      /* [V,E]                                   */
    const float* w_wih;    /* worker.weight_ih_l0   [4H,E]            */
    const float* w_whh;    /* worker.weight_hh_l0   [4H,H]            */
    const float* w_bih;    /* [4H] */
    const float* w_bhh;    /* [4H] */
    const float* m_wih;    /* manager.weight_ih_l0  [4H,G]            */
    const float* m_whh;    /* manager.weight_hh_l0  [4H,H]            */
    const float* m_bih;    /* [4H] */
    const float* m_bhh;    /* [4H] */
    const float* w2g_w;    /* work2goal.weight      [V*K,H]           */
    const float* w2g_b;    /* work2goal.bias        [V*K]             */
    const float* m2g_w;    /* mana2goal.weight      [G,H]             */
    const float* m2g_b;    /* mana2goal.bias        [G]               */
    const float* g2g_w;    /* goal2goal.weight      [K,G] (no bias)   */
    const float* goal_init;/* goal_init             [B_init,G]        */
    int32_t V, E, H, G, K, step_size, B_init, _pad;
} OrcLeakWeights;

/* nn.LSTM cell on an input vector x [nin] (gates i,f,g,o; b_ih + b_hh), transposed weights wih_t [nin][4H],
 * whh_t [H][4H], bsum [4H].  Same arithmetic order as lstm_cell above. */
static void lstm_cell_vec(const float* wih_t, const float* whh_t, const float* bsum, int nin, int H, const float* x,
                          float* h, float* c, float* gates) {
    const int G4 = 4 * H;
    for (int r = 0; r < G4; ++r) gates[r] = bsum[r];
    for (int k = 0; k < nin; ++k) {
        const float xv = x[k];
        const float* wr = wih_t + (size_t)k * G4;
        for (int r = 0; r < G4; ++r) gates[r] += wr[r] * xv;
    }
    for (int k = 0; k < H; ++k) {
        const float hv = h[k];
        const float* wr = whh_t + (size_t)k * G4;
        for (int r = 0; r < G4; ++r) gates[r] += wr[r] * hv;
    }
    for (int j = 0; j < H; ++j) {
        const float ig = sigmoidf_(gates[j]);
        const float fg = sigmoidf_(gates[H + j]);
        const float gg = tanhf(gates[2 * H + j]);
        const float og = sigmoidf_(gates[3 * H + j]);
        const float cn = fg * c[j] + ig * gg;
        c[j] = cn;
        h[j] = og * tanhf(cn);
    }
}

static float* transpose_new(const float* a, int rows, int cols) { /* [rows][cols] -> [cols][rows] */
    float* t = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = a[(size_t)r * cols + c];
    return t;
}

/* F.normalize(x, p=2, dim=-1): x / max(||x||_2, 1e-12) (LeakGAN_G.py:74,76). */
static void l2_normalize(float* x, int n) {
    float ss = 0.f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    float nrm = sqrtf(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (int i = 0; i < n; ++i) x[i] = x[i] / nrm;
}

typedef struct LeakCtx {
    const OrcLeakWeights* w;
    const OrcDisWeights* d;
    DisT dt;
    float *w_wih_t, *w_whh_t, *m_wih_t, *m_whh_t, *w2g_t, *m2g_t, *w_bsum, *m_bsum;
} LeakCtx;

static void leak_ctx_init(LeakCtx* c, const OrcLeakWeights* w, const OrcDisWeights* d) {
    const int V = w->V, E = w->E, H = w->H, G = w->G, K = w->K;
    c->w = w; c->d = d;
    dis_t_init(&c->dt, d);
    c->w_wih_t = transpose_new(w->w_wih, 4 * H, E);
    c->w_whh_t = transpose_new(w->w_whh, 4 * H, H);
    c->m_wih_t = transpose_new(w->m_wih, 4 * H, G);
    c->m_whh_t = transpose_new(w->m_whh, 4 * H, H);
    c->w2g_t = transpose_new(w->w2g_w, V * K, H);   /* [H][V*K] */
    c->m2g_t = transpose_new(w->m2g_w, G, H);       /* [H][G]   */
    c->w_bsum = (float*)malloc(sizeof(float) * 4 * H);
    c->m_bsum = (float*)malloc(sizeof(float) * 4 * H);
    for (int r = 0; r < 4 * H; ++r) { c->w_bsum[r] = w->w_bih[r] + w->w_bhh[r]; c->m_bsum[r] = w->m_bih[r] + w->m_bhh[r]; }
}
static void leak_ctx_free(LeakCtx* c) {
    free(c->w_wih_t); free(c->w_whh_t); free(c->m_wih_t); free(c->m_whh_t); free(c->w2g_t); free(c->m2g_t);
    free(c->w_bsum); free(c->m_bsum);
    dis_t_free(&c->dt, c->d);
}

typedef struct LeakScratch {
    float *e, *pooled, *acc, *feat, *hw, *cw, *hm, *cm, *gates, *goals, *real_goal, *rg, *wo, *logits, *wts;
    int64_t* fed;
} LeakScratch;

static void leak_scratch_init(LeakScratch* s, const LeakCtx* c, int L) {
    const OrcLeakWeights* w = c->w;
    const int V = w->V, H = w->H, G = w->G, K = w->K;
    s->e = (float*)malloc(sizeof(float) * (size_t)L * c->d->Ed);
    s->pooled = (float*)malloc(sizeof(float) * G);
    s->acc = (float*)malloc(sizeof(float) * 1024);
    s->feat = (float*)malloc(sizeof(float) * G);
    s->hw = (float*)malloc(sizeof(float) * H); s->cw = (float*)malloc(sizeof(float) * H);
    s->hm = (float*)malloc(sizeof(float) * H); s->cm = (float*)malloc(sizeof(float) * H);
    s->gates = (float*)malloc(sizeof(float) * 4 * H);
    s->goals = (float*)malloc(sizeof(float) * (size_t)(L + 1) * G);  /* goal_array of one row */
    s->real_goal = (float*)malloc(sizeof(float) * G);
    s->rg = (float*)malloc(sizeof(float) * K);
    s->wo = (float*)malloc(sizeof(float) * (size_t)V * K);
    s->logits = (float*)malloc(sizeof(float) * V);
    s->wts = (float*)malloc(sizeof(float) * V);
    s->fed = (int64_t*)malloc(sizeof(int64_t) * L);
}
static void leak_scratch_free(LeakScratch* s) {
    free(s->e); free(s->pooled); free(s->acc); free(s->feat); free(s->hw); free(s->cw); free(s->hm); free(s->cm);
    free(s->gates); free(s->goals); free(s->real_goal); free(s->rg); free(s->wo); free(s->logits); free(s->wts); free(s->fed);
}

/* One row of rollout_mc_search_leakgan: sentence `sent` [L] (tokens < given are copied), goal_init row b.
 * u: the row's uniforms, u[pos * u_stride]; forced [L] or NULL; out [L].  If probe_logits != NULL the (temperature-
 * scaled, pre-log_softmax) outputs of position probe_pos are copied there and the row stops. */
static void leak_row(const LeakCtx* c, LeakScratch* s, const int64_t* sent, int b, int L, int given, const float* u,
                     size_t u_stride, const int64_t* forced, int64_t* out, int probe_pos, float* probe_logits) {
    const OrcLeakWeights* w = c->w;
    const int V = w->V, E = w->E, H = w->H, G = w->G, K = w->K, step = w->step_size;
    memset(s->hw, 0, sizeof(float) * H); memset(s->cw, 0, sizeof(float) * H);
    memset(s->hm, 0, sizeof(float) * H); memset(s->cm, 0, sizeof(float) * H);
    memset(s->goals, 0, sizeof(float) * (size_t)(L + 1) * G);
    memcpy(s->real_goal, w->goal_init + (size_t)b * G, sizeof(float) * G);
    memset(s->fed, 0, sizeof(int64_t) * L);   /* zero padding of the feature extractor's input */
    for (int pos = 0; pos < L; ++pos) {
        int64_t tok;
        if (pos < given) {
            tok = sent[pos];
            if (out) out[pos] = tok;
        } else {
            exp_weights(s->logits, V, s->wts);  /* exp(log_softmax) up to the normaliser (rollout.py:101) */
            tok = canonical_draw(s->wts, V, u[(size_t)pos * u_stride]);
            if (out) out[pos] = tok;
            if (forced) tok = forced[pos];
        }
        s->fed[pos] = tok;
        if (pos == L - 1) break;            /* the reference's last forward is never read */
        dis_feature_row(c->d, &c->dt, s->fed, L, s->e, s->pooled, s->acc, s->feat);
        /* manager (LeakGAN_G.py:72-74) */
        lstm_cell_vec(c->m_wih_t, c->m_whh_t, c->m_bsum, G, H, s->feat, s->hm, s->cm, s->gates);
        float* cur = s->goals + (size_t)pos * G;
        for (int o = 0; o < G; ++o) cur[o] = w->m2g_b[o];
        for (int k = 0; k < H; ++k) {
            const float hv = s->hm[k];
            const float* wr = c->m2g_t + (size_t)k * G;
            for (int o = 0; o < G; ++o) cur[o] += wr[o] * hv;
        }
        l2_normalize(cur, G);
        /* goal2goal on the real goal in force BEFORE this step's update (LeakGAN_G.py:75-76) */
        for (int k = 0; k < K; ++k) {
            float a = 0.f;
            const float* wr = w->g2g_w + (size_t)k * G;
            for (int o = 0; o < G; ++o) a += wr[o] * s->real_goal[o];
            s->rg[k] = a;
        }
        l2_normalize(s->rg, K);
        /* worker (LeakGAN_G.py:79-84) */
        lstm_cell_vec(c->w_wih_t, c->w_whh_t, c->w_bsum, E, H, w->emb + (size_t)tok * E, s->hw, s->cw, s->gates);
        const size_t VK = (size_t)V * K;
        for (size_t q = 0; q < VK; ++q) s->wo[q] = w->w2g_b[q];
        for (int k = 0; k < H; ++k) {
            const float hv = s->hw[k];
            const float* wr = c->w2g_t + (size_t)k * VK;
            for (size_t q = 0; q < VK; ++q) s->wo[q] += wr[q] * hv;
        }
        const float temperature = (pos > 1) ? 1.0f : 1.5f;   /* train=True (LeakGAN_G.py:87-95) */
        for (int v = 0; v < V; ++v) {
            float a = 0.f;
            const float* wr = s->wo + (size_t)v * K;
            for (int k = 0; k < K; ++k) a += wr[k] * s->rg[k];
            s->logits[v] = temperature * a;
        }
        if (probe_logits && pos == probe_pos) {
            memcpy(probe_logits, s->logits, sizeof(float) * V);
            return;
        }
        /* goal bookkeeping (rollout.py:89-93 / 117-121): the hard-coded window of 4 goals is the reference's */
        if (pos > 0 && pos % step == 0) {
            /* goal_array[:, pos-3:pos+1]: for pos < 3 (step_size < 3) the negative start makes the Python slice
             * EMPTY (it counts from the end of the L+1 axis), so the sum is zero */
            for (int o = 0; o < G; ++o) {
                float a = 0.f;
                if (pos >= 3)
                    for (int j = pos - 3; j <= pos; ++j) a += s->goals[(size_t)j * G + o];
                s->real_goal[o] = a;
            }
            if (pos == step)
                for (int o = 0; o < G; ++o) s->real_goal[o] += w->goal_init[(size_t)b * G + o];
        }
    }
}

static int leak_check(const OrcLeakWeights* w, const OrcDisWeights* d, int B, int L) {
    if (w->G != d->F || B > w->B_init || w->step_size < 1 || L < 2) return -2;
    for (int k = 0; k < d->n_filters; ++k) if (d->f[k] > L) return -1;
    return 0;
}

/* ROLLOUT.rollout_mc_search_leakgan for every (rollout i, t) pair of get_reward_leakgan's loops
 * (utils/rollout.py:57-126, 172-175; given_num = t*step_size + 1), one row per (i, t, b):
 *   for pos in 0..L-1:
 *     token  = sentences[b,pos] if pos < given_num, else the canonical draw from exp(out) of the step before
 *              (rollout.py:76-77, 101-102);
 *     feature = dis.get_feature(tokens 0..pos, zero-padded to L)                  (rollout.py:75-81, 105-108)
 *     LeakGAN_G.forward(pos, token, ..., feature, real_goal, train=True)          (LeakGAN_G.py:52-102):
 *        manager LSTM on the feature, cur_goal = normalize(mana2goal(h_m)), rg = normalize(goal2goal(real_goal)),
 *        worker LSTM on the token's embedding, out = temperature * (work2goal(h_w).view(V,K) @ rg),
 *        temperature 1.5 for pos <= 1 else 1.0 (train=True), log_softmax;
 *     goal_array[pos] = cur_goal; if pos > 0 and pos % step_size == 0: real_goal = sum(goal_array[pos-3..pos])
 *        (+ goal_init[b] when pos == step_size)                                   (rollout.py:89-93, 117-121)
 * The forward after the last token is skipped (its outputs are never read).
 * uniforms [R,T,L,B] (draw for position pos at [i,t,pos,b]); completions [R,T,B,L];
 * forced [R,T,B,L] or NULL = replay mode: the drawn token is written, the forced one is fed on (to the worker and to
 * the feature extractor). */
int orc_leak_rollout(const OrcLeakWeights* w, const OrcDisWeights* d, const int64_t* sentences, int B, int L, int R,
                     int i_lo, int i_hi, const float* uniforms, const int64_t* forced, int64_t* completions) {
    const int T = L / w->step_size;
    int rc = leak_check(w, d, B, L);
    if (rc) return rc;
    LeakCtx c; leak_ctx_init(&c, w, d);
    const long n_items = (long)(i_hi - i_lo) * T * B;
#pragma omp parallel
    {
        LeakScratch s; leak_scratch_init(&s, &c, L);
#pragma omp for schedule(dynamic, 1)
        for (long it = 0; it < n_items; ++it) {
            const int b = (int)(it % B);
            const int t = (int)((it / B) % T);
            const int i = (int)(it / ((long)B * T)) + i_lo;
            const size_t row = ((size_t)i * T + t) * B + b;
            leak_row(&c, &s, sentences + (size_t)b * L, b, L, t * w->step_size + 1,
                     uniforms + ((size_t)i * T + t) * L * B + b, (size_t)B, forced ? forced + row * L : NULL,
                     completions + row * L, -1, NULL);
        }
        leak_scratch_free(&s);
    }
    leak_ctx_free(&c);
    return 0;
}

/* The generator's outputs (temperature * work2goal(h_w).view(V,K) @ rg, before log_softmax) after feeding
 * tokens[0..pos] of one row whose goal_init row is b: what the draw of position pos+1 is taken from.  For the
 * parity tests' CDF-tie check. */
int orc_leak_logits_at(const OrcLeakWeights* w, const OrcDisWeights* d, const int64_t* tokens, int b, int L, int pos,
                       float* logits_out) {
    int rc = leak_check(w, d, b + 1, L);
    if (rc || pos < 0 || pos >= L - 1) return rc ? rc : -2;
    LeakCtx c; leak_ctx_init(&c, w, d);
    LeakScratch s; leak_scratch_init(&s, &c, L);
    leak_row(&c, &s, tokens, b, L, L, NULL, 0, NULL, NULL, pos, logits_out);
    leak_scratch_free(&s);
    leak_ctx_free(&c);
    return 0;
}

/* ROLLOUT.get_reward_leakgan (utils/rollout.py:156-184) end to end: completions [R,T,B,L] (T = L // step_size),
 * raw [R*T, B] (row i*T + t = softmax(dis(samples))[:, col], D's dropout from dropout_uniforms [R*T*B, F] or NULL),
 * out [B,T] = mean over the last axis of raw viewed as [B,T,R] (rollout.py:182-183: a reinterpretation of the flat
 * buffer, like get_reward's) or NULL. */
int orc_get_reward_leakgan(const OrcLeakWeights* g, const OrcDisWeights* d, const int64_t* sentences, int B, int L,
                           int R, int i_lo, int i_hi, int col, const float* uniforms, float dropout_p,
                           const float* dropout_uniforms, int64_t* completions, float* raw, float* out) {
    const int T = L / g->step_size;
    int rc = orc_leak_rollout(g, d, sentences, B, L, R, i_lo, i_hi, uniforms, NULL, completions);
    if (rc) return rc;
    const size_t row0 = (size_t)i_lo * T * B;
    const int n = (i_hi - i_lo) * T * B;
    rc = orc_dis_forward(d, completions + row0 * L, n, L, col, dropout_p,
                         dropout_uniforms ? dropout_uniforms + row0 * d->F : NULL, raw + row0, NULL, NULL);
    if (rc) return rc;
    if (out) orc_reward_reduce(raw, B, T, R, out);
    return 0;
}
