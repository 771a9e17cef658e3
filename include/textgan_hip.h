/*
 * textgan_hip.h -- C ABI of libtextgan_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the Monte-Carlo rollout reward path of
 * williamSYSU/TextGAN-PyTorch.  The reference has no FFI of its own (it is
 * pure Python on ATen), so every entry point below names the reference Python
 * symbol (file:line under the reference tree) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - all floating-point data is fp32, all token ids are int64 (the
 *     reference's dtypes: utils/rollout.py:41, models/generator.py:68);
 *   - row-major, innermost index last;
 *   - every call enqueues work on `stream` and returns without synchronising;
 *     inputs are borrowed, outputs are caller-allocated;
 *   - return value: 0 = ok, <0 = error (message via tg_last_error()).
 *   - no entry point allocates device memory: scratch comes in through the
 *     `workspace` arguments, sized by the *_workspace_bytes() helpers.
 */
#ifndef TEXTGAN_HIP_H
#define TEXTGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_MAX_FILTERS 16

#define TG_OK 0
#define TG_ERR_ARG (-1)      /* bad argument / unsupported shape            */
#define TG_ERR_LAUNCH (-2)   /* HIP launch or runtime error                 */
#define TG_ERR_WORKSPACE (-3)/* workspace too small                         */

/* Alignment.  Weight tensors (every pointer in TgGenWeights / TgLeakWeights; the embedding, conv and highway
 * matrices of TgDisWeights) must be 16-byte aligned -- torch parameters are (own allocations, 256-byte aligned);
 * a misaligned one is rejected with TG_ERR_ARG.  Workspaces must be 256-byte aligned.  Token, uniform and output
 * arrays need only their natural alignment (the scorer takes a slower exact-fp32 highway when its bias, output layer,
 * dropout uniforms or feature output are not 16-byte aligned).
 * Empty batches (N = 0, B = 0, or an empty rollout-index range) are no-ops that return TG_OK whatever the data
 * pointers hold, as ATen handles a zero-row batch in the reference; negative sizes and rollout_num 0 are errors. */

/* LSTMGenerator parameters: models/generator.py:33-35 (state_dict keys
 * embeddings.weight, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0,
 * lstm.bias_hh_l0, lstm2out.weight, lstm2out.bias).  Gate order i,f,g,o. */
typedef struct TgGenWeights {
    const float* emb;    /* [V, E]   */
    const float* w_ih;   /* [4H, E]  */
    const float* w_hh;   /* [4H, H]  */
    const float* b_ih;   /* [4H]     */
    const float* b_hh;   /* [4H]     */
    const float* w_out;  /* [V, H]   */
    const float* b_out;  /* [V]      */
    int32_t V, E, H;
    int32_t precision;   /* TG_GEN_FP32 or TG_GEN_SPLIT_F16 (vocabulary projection of the sampling kernels) */
} TgGenWeights;

/* Arithmetic of the vocabulary projection h.W_out^T inside tg_gen_rollout / tg_gen_sample (the LSTM cell, the
 * softmax, the sampler's prefix sums and everything in tg_gen_teacher_forced are fp32 / fp64 regardless):
 *   TG_GEN_FP32       exact fp32 (v_mfma_f32_32x32x2_f32)
 *   TG_GEN_SPLIT_F16  operands scaled by a power of two and split into fp16 hi + lo (22 mantissa bits), three
 *                     fp16 MFMAs per product with fp32 accumulation, bias added exactly; logits move by ~1e-6.
 *                     E = H = 32 (every SeqGAN / SentiGAN config of the reference): inside the persistent kernel.
 *                     Other shapes (E, H multiples of 32, V <= 65536 -- the scaled config): tg_gen_rollout and
 *                     tg_gen_sample / tg_gen_nll run step-synchronous split-fp16 GEMMs for the gates and the
 *                     projection (tg_genw.hip); tg_gen_teacher_forced stays fp32 there. */
#define TG_GEN_FP32 0
#define TG_GEN_SPLIT_F16 1

/* CNNDiscriminator parameters: models/discriminator.py:30-36 (state_dict keys
 * embeddings.weight, convs.k.weight [n_k,1,f_k,Ed], convs.k.bias,
 * highway.weight/bias [F,F], feature2out.weight/bias [n_out,F]). */
typedef struct TgDisWeights {
    const float* emb;                     /* [V, Ed]                 */
    const float* conv_w[TG_MAX_FILTERS];  /* [n_k, f_k*Ed]           */
    const float* conv_b[TG_MAX_FILTERS];  /* [n_k]                   */
    const float* hw_w;                    /* [F, F]                  */
    const float* hw_b;                    /* [F]                     */
    const float* out_w;                   /* [n_out, F]              */
    const float* out_b;                   /* [n_out]                 */
    int32_t f[TG_MAX_FILTERS];            /* filter widths f_k       */
    int32_t n[TG_MAX_FILTERS];            /* filters per width n_k   */
    int32_t n_filters;                    /* number of widths        */
    int32_t V, Ed, F, n_out;
    int32_t precision;                    /* TG_DIS_FP32 or TG_DIS_SPLIT_F16 (conv + highway)   */
    uint32_t dropout_seed;                /* used when dropout_p > 0 and dropout_uniforms == NULL:
                                           * element (n, f) of the call is kept iff
                                           * hash32(n*F + f + seed*0x9E3779B9) < (1-p)*2^32 (lowbias32
                                           * finaliser, tg_common.h) -- no [N,F] tensor of uniforms
                                           * is materialised.  Vary the seed between calls.      */
} TgDisWeights;

/* Arithmetic of the scorer's convolutions and highway GEMM (gate, dropout, output layer, softmax: always fp32):
 *   TG_DIS_FP32       exact fp32 (v_mfma_f32_32x32x2_f32)
 *   TG_DIS_SPLIT_F16  each operand x*2^k = hi + lo in fp16 (22 mantissa bits), three fp16 MFMAs with fp32
 *                     accumulation per product (hi.hi + hi.lo + lo.hi): ~5e-7 relative per product,
 *                     about 4x the fp32 throughput. */
#define TG_DIS_FP32 0
#define TG_DIS_SPLIT_F16 1

/* Library / build identification ("textgan_hip <ver> gfx950"). host string. */
const char* tg_version(void);

/* Last error message of the calling thread. host string. */
const char* tg_last_error(void);

/* 0: the (V,E,H) generator shape is supported by every entry point and precision (H in {32,64,128,256,512}, E a
 * multiple of 32 up to 512); 1: E, H multiples of 32 up to 1024 outside that set -- tg_gen_rollout, tg_gen_sample and
 * tg_gen_nll only, with TG_GEN_SPLIT_F16 and V <= 65536 (the GEMM path); -1: not supported. */
int tg_gen_supported(int V, int E, int H);

/* ---------------------------------------------------------------------------
 * Teacher-forced generator pass.
 * Replaces LSTMGenerator.forward (models/generator.py:40-60) over a whole
 * [N,L] input with zero initial state (init_hidden, :102-109), as used by
 * NLL.cal_nll (metrics/nll.py:59-61) and by the rollout prefix
 * (utils/rollout.py:35-38).
 *   inp        [N,L]      int64, token fed at each step
 *   target     [N,L]      int64 or NULL; if given, logp_out[n,t] =
 *                         log_softmax(logits[n,t,:])[target[n,t]]
 *   logp_out   [N,L]      or NULL
 *   lse_out    [N,L]      or NULL: log-sum-exp of logits[n,t,:]
 *   logits_out [N,L,V]    or NULL: raw lstm2out outputs (parity tests only)
 *   states_out [L+1,N,2,H] or NULL: (h,c) after 0..L consumed tokens
 *                         (slot 0 = zeros); the rollout prefix states.
 * If target, lse_out and logits_out are all NULL the vocab projection is
 * skipped entirely (prefix-state mode). */
int tg_gen_teacher_forced(const TgGenWeights* w, const int64_t* inp,
                          const int64_t* target, int N, int L,
                          float* logp_out, float* lse_out, float* logits_out,
                          float* states_out, void* stream);

/* tg_gen_teacher_forced for EVERY generator shape tg_gen_supported accepts (0 or 1): shapes of class 0 run the
 * same persistent kernel (the workspace is not touched and may be NULL); class 1 -- E, H multiples of 32 up to 1024
 * outside the persistent kernel's set: H = 96, 160, 768, ... -- runs the pass as exact-fp32 GEMMs on
 * v_mfma_f32_32x32x2_f32 (tg_pgw.hip: per position the gates GEMM + cell, then the projection over chunks of (t, n)
 * rows with log-sum-exp and the target's logit per row).  Same outputs, same fp32 arithmetic contract.
 *   states_in  [N,2,H] fp32 or NULL: the (h, c) the pass starts from -- LSTMGenerator.forward's `hidden` argument
 *              (models/generator.py:40-51 accepts any state, not only init_hidden's zeros); NULL = zeros.  With
 *              states_out, slot 0 then holds this state.  16-byte aligned.  (ABI 0.3: inserted after L.)
 *   workspace  tg_gen_teacher_forced_workspace_bytes(w, N, L, logits_out != NULL, states_out != NULL) bytes */
size_t tg_gen_teacher_forced_workspace_bytes(const TgGenWeights* w, int N, int L, int have_logits_out,
                                             int have_states_out);
int tg_gen_teacher_forced_ws(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                             const float* states_in, float* logp_out, float* lse_out, float* logits_out,
                             float* states_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * NLL pass: tg_gen_teacher_forced restricted to what NLL.cal_nll needs (metrics/nll.py:49-63: the log-probability
 * of each target token), honouring w->precision.  With TG_GEN_SPLIT_F16 (E = H = 32) the log-sum-exp comes from
 * the split-fp16 projection (probabilities within ~1e-6 relative of fp32) while the target's own logit is an exact
 * fp32 dot product; with TG_GEN_FP32, or any other shape, the result is tg_gen_teacher_forced's, bit for bit.
 * Shapes beyond E = H = 32 with TG_GEN_SPLIT_F16: the step-synchronous GEMM path of tg_genw.hip (cell and projection
 * on split-fp16 MFMAs, log-sum-exp from the per-tile records, the target's logit an fp32 dot product).
 *   logp_out [N,L] required; lse_out [N,L] or NULL; workspace of tg_gen_nll_workspace_bytes(w, N) (0 for fp32). */
size_t tg_gen_nll_workspace_bytes(const TgGenWeights* w, int N);
int tg_gen_nll(const TgGenWeights* w, const int64_t* inp, const int64_t* target,
               int N, int L, float* logp_out, float* lse_out, void* workspace,
               size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Monte-Carlo completions.
 * Replaces ROLLOUT.rollout_mc_search (utils/rollout.py:25-55) for every
 * (i, given_num) pair of ROLLOUT.get_reward's double loop (:143-145) in one
 * launch, sharing the prefix LSTM state between all completions of a sentence.
 *   sentences   [B,L]        int64
 *   uniforms    [R,L,L,B]    fp32 in [0,1): the draw for rollout i,
 *                            given_num g (1-based), position t, sentence b is
 *                            uniforms[((i*L + g-1)*L + t)*B + b] (t >= g used)
 *   completions [R,L,B,L]    int64: completions[i, g-1] is the [B,L] tensor
 *                            rollout_mc_search(sentences, g) returns; g = L is
 *                            the sentence itself.  Only i in [i_lo,i_hi) and
 *                            g in [g_lo,g_hi] are written.
 *   forced      [R,L,B,L]    int64 or NULL.  Replay mode for parity tests: the
 *                            drawn token is still what is written, but the
 *                            token fed to the next step is forced[...], so one
 *                            fp32 CDF tie does not cascade down the row.
 *   workspace   prefix states (+ the split projection's W_out fragments; shapes beyond E = H = 32 in
 *               TG_GEN_SPLIT_F16: operand fragments, per-row state and sampler records of one chunk of rows),
 *               tg_gen_rollout_workspace_bytes(w,B,L,i_hi-i_lo) bytes.
 * Sampling is inverse-CDF in the canonical order documented in DESIGN.md
 * ("canonical sampler") -- it replaces torch.multinomial(exp(out),1)
 * (utils/rollout.py:49). */
size_t tg_gen_rollout_workspace_bytes(const TgGenWeights* w, int B, int L, int R /* rollout indices of the call: i_hi - i_lo */);
int tg_gen_rollout(const TgGenWeights* w, const int64_t* sentences, int B,
                   int L, int R, int g_lo, int g_hi, int i_lo, int i_hi,
                   const float* uniforms, const int64_t* forced,
                   int64_t* completions, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Free-running sampling.
 * Replaces the step loop of LSTMGenerator.sample (models/generator.py:71-81)
 * for N rows at once: zero state, first input `start_letter`.
 *   uniforms [L,N] fp32: draw for row n at position t is uniforms[t*N + n]
 *   forced   [N,L] int64 or NULL (replay mode, as in tg_gen_rollout)
 *   out      [N,L] int64
 *   workspace  tg_gen_sample_workspace_bytes(w, N) bytes (0 for TG_GEN_FP32: may be NULL) */
size_t tg_gen_sample_workspace_bytes(const TgGenWeights* w, int N);
int tg_gen_sample(const TgGenWeights* w, int N, int L, int64_t start_letter,
                  const float* uniforms, const int64_t* forced, int64_t* out,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * tg_gen_sample without any allocation, by the library or by the caller.
 * No counterpart in the reference: this entry point exists for callers that cannot allocate device memory -- code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc -- and so cannot bring the
 * `workspace` tg_gen_sample asks for.  Same arguments (minus the workspace), same kernels, same tokens as
 * tg_gen_sample with TG_GEN_SPLIT_F16 at E = H = 32; the scratch (the split projection's |max| header and W_out
 * fragments) is a static device array of tg_gen_sample_static_bytes() = 2 621 696 bytes that every GPU the library
 * runs on holds until the process ends.  Limits: E = H = 32, TG_GEN_SPLIT_F16, V <= TG_STATIC_MAX_V (TG_ERR_ARG with
 * a message otherwise).  Calls no hipMalloc / hipFree.  All calls on one GPU share that GPU's array: two calls that
 * overlap on one GPU (different streams, different threads) race -- serialise them.
 * Under HIP graph capture: make ONE ordinary call per device outside capture first.  The first call on a device
 * resolves the array's address (hipGetSymbolAddress under a host mutex), the first launch of each kernel
 * instantiation sets its dynamic-LDS attribute (hipFuncSetAttribute) and loads the code object -- none of which
 * belongs inside a capturing stream; later calls only enqueue kernels and one hipMemsetAsync on `stream`. */
#define TG_STATIC_MAX_V 16384
size_t tg_gen_sample_static_bytes(void);
int tg_gen_sample_static(const TgGenWeights* w, int N, int L, int64_t start_letter,
                         const float* uniforms, const int64_t* forced, int64_t* out,
                         void* stream);

/* ---------------------------------------------------------------------------
 * CNN discriminator scorer.
 * Replaces CNNDiscriminator.forward / get_feature
 * (models/discriminator.py:40-64) followed by F.softmax(...)[:, col]
 * (utils/rollout.py:146-148).
 *   seqs             [N,L]      int64
 *   col                         softmax column (current_k + 1)
 *   dropout_p                   0 = eval mode (identity)
 *   dropout_uniforms [N,F] fp32 or NULL: explicit U[0,1) draws (keep iff u < 1-p);
 *                               NULL with dropout_p > 0 = counter-based masks from
 *                               TgDisWeights.dropout_seed
 *                               kept values are scaled 1/(1-p) (nn.Dropout, discriminator.py:36,47)
 *   reward_out       [N]        or NULL
 *   logits_out       [N,n_out]  or NULL (feature2out outputs, parity tests)
 *   feature_out      [N,F]      or NULL (get_feature output, parity tests)
 *   workspace        tg_dis_workspace_bytes(w,N,L) bytes. */
size_t tg_dis_workspace_bytes(const TgDisWeights* w, int N, int L);
int tg_dis_score(const TgDisWeights* w, const int64_t* seqs, int N, int L,
                 int col, float dropout_p, const float* dropout_uniforms,
                 float* reward_out, float* logits_out, float* feature_out,
                 void* workspace, size_t workspace_bytes, void* stream);
/* tg_dis_score in two calls on the same workspace, same results to the bit (see tg_dis_score_rollout_prepare below):
 * ..._prepare does what needs only D's parameters, ..._prepared the rest.  get_reward_leakgan (utils/rollout.py:156-184)
 * scores its completions after the rollouts; the shim enqueues the first part on a second stream beside them.
 * dropout_uniforms / feature_out: the pointers the second call will pass (their alignment decides which highway
 * kernel the weights are prepared for), or NULL; nothing is read or written through them here. */
int tg_dis_score_prepare(const TgDisWeights* w, int N, int L, const float* dropout_uniforms, const float* feature_out,
                         void* workspace, size_t workspace_bytes, void* stream);
int tg_dis_score_prepared(const TgDisWeights* w, const int64_t* seqs, int N, int L,
                          int col, float dropout_p, const float* dropout_uniforms,
                          float* reward_out, float* logits_out, float* feature_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * CNN discriminator scorer for rollout completions.
 * Same result as tg_dis_score(completions viewed as [R*L*B, L]) -- the
 * dis.forward(samples) calls inside ROLLOUT.get_reward's loop
 * (utils/rollout.py:146-148) -- but exploits what rollout_mc_search guarantees
 * (utils/rollout.py:42): completions[i, g-1, b, :g] == sentences[b, :g].  Conv
 * windows that lie inside that shared prefix are evaluated once per sentence
 * (a running max over time per filter) instead of once per completion; max
 * pooling is order-independent, so the result is bit-identical.
 *   sentences        [B,L]        int64
 *   completions      [R,L,B,L]    int64 (tg_gen_rollout's output layout)
 *   dropout_uniforms [R*L*B, F]   or NULL
 *   reward_out       [R*L, B]     raw reward rows (row i*L + g-1)
 *   workspace        tg_dis_score_rollout_workspace_bytes(w,B,L,R) bytes. */
size_t tg_dis_score_rollout_workspace_bytes(const TgDisWeights* w, int B, int L, int R);
int tg_dis_score_rollout(const TgDisWeights* w, const int64_t* sentences,
                         const int64_t* completions, int B, int L, int R,
                         int col, float dropout_p,
                         const float* dropout_uniforms, float* reward_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* tg_dis_score_rollout in two calls on the same workspace, same results to the bit.  ..._prepare does what needs only
 * D's parameters and the sentences (models/discriminator.py:40-64 reads neither anything else nor the completions for
 * it: the weights' preparation, the sentences' window values and their running maxima); ..._prepared does the rest.
 * ROLLOUT.get_reward (utils/rollout.py:128-154) calls gen and dis one after the other; with the split the first part
 * can be enqueued on a second stream beside the generator's kernels (rollout.py of the shim does).  The caller orders
 * the two calls (and any earlier use of the workspace memory) itself.  dropout_uniforms: the pointer the second call
 * will pass (its alignment decides which highway kernel the weights are prepared for), or NULL. */
int tg_dis_score_rollout_prepare(const TgDisWeights* w, const int64_t* sentences, int B, int L, int R,
                                 const float* dropout_uniforms, void* workspace, size_t workspace_bytes,
                                 void* stream);
int tg_dis_score_rollout_prepared(const TgDisWeights* w, const int64_t* sentences,
                                  const int64_t* completions, int B, int L, int R,
                                  int col, float dropout_p,
                                  const float* dropout_uniforms, float* reward_out,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Dropout-only re-scoring (MaliGAN reward).
 * Replaces the loop of MaliGANInstructor.get_mali_reward
 * (instructor/real_data/maligan_instructor.py:142-147 and the oracle_data
 * twin): rollout_num forward passes of D over the SAME samples differ only in
 * their dropout masks, so the features (tg_dis_score's feature_out, the highway
 * output before dropout) are computed once and this entry point applies R masks
 * and the output layer:
 *   reward_out[r, n] = softmax(dropout_r(feature[n, :]) . out_w^T + out_b)[col]
 *   feature           [N,F]    fp32
 *   dropout_uniforms  [R,N,F]  fp32: keep iff u < 1-p, kept values scaled 1/(1-p)
 *   reward_out        [R,N]    fp32
 * dropout_p = 0 gives R identical rows (eval mode). */
int tg_dis_rescore_dropout(const TgDisWeights* w, const float* feature, int N,
                           int R, int col, float dropout_p,
                           const float* dropout_uniforms, float* reward_out,
                           void* stream);

/* ---------------------------------------------------------------------------
 * Backward of the policy-gradient loss.
 * Replaces what autograd does behind SeqGAN_G.batchPGLoss (models/SeqGAN_G.py:21-39), MaliGAN_G.adv_loss
 * (models/MaliGAN_G.py:21-39) and SentiGAN_G.batchPGLoss (models/SentiGAN_G.py:47-65): those build the
 * [B*L, V] log-softmax matrix, multiply it with a [B, L, V] one-hot of the target and back-propagate through both.
 * The forward half is tg_gen_teacher_forced (logp_out, lse_out, states_out); this call takes
 *   coeff  [N,L]  dLoss/dlogp[n,t]   (SeqGAN / MaliGAN: -reward[n,t] * grad_out;
 *                                     SentiGAN, which sums probabilities: -(1-reward[n,t]) * exp(logp[n,t]) * grad_out)
 *   lse    [N,L], states [L+1,N,2,H]  as written by tg_gen_teacher_forced for the same (w, inp)
 * and writes the gradients of all seven parameters (each buffer fully overwritten; shapes as in TgGenWeights;
 * b_ih and b_hh receive the same values).  Row padding_idx of the embedding gradient stays zero, as
 * nn.Embedding(padding_idx=...) has it (models/generator.py:33); pass -1 for none.
 * The logits are recomputed tile by tile on the matrix cores; no [N*L, V] buffer exists.  fp32 throughout.
 * The embedding gradient is accumulated with float atomics (the order of additions, and so the last bit, can
 * vary between calls -- as torch's own embedding backward on a GPU); everything else is summed in fixed order.
 * E = H = 32 (every SeqGAN / MaliGAN / SentiGAN configuration of the reference): the fused kernels of tg_pg.hip.
 * Any other E, H (multiples of 32 up to 1024, e.g. the scaled config's 512): the same gradients as exact-fp32 GEMMs
 * (tg_pgw.hip): the logits of <= 1 GiB worth of (t, n) rows at a time are recomputed, turned into dlogits in place
 * and contracted three ways (dh_proj, dW_out, db_out), then L steps back through time, then the weight gradients of
 * all steps as three large GEMMs. */
typedef struct TgGenGrads {
    float* emb;    /* [V, E]   */
    float* w_ih;   /* [4H, E]  */
    float* w_hh;   /* [4H, H]  */
    float* b_ih;   /* [4H]     */
    float* b_hh;   /* [4H]     */
    float* w_out;  /* [V, H]   */
    float* b_out;  /* [V]      */
} TgGenGrads;
size_t tg_gen_pg_backward_workspace_bytes(const TgGenWeights* w, int N, int L);
/* Forward half for the same loss, shaped for an adversarial batch (few sequences): the cell-only pass of
 * tg_gen_teacher_forced writes states_out [L+1,N,2,H] (required), then the projection, lse_out [N,L] and
 * logp_out[n,t] = log_softmax(logits[n,t,:])[target[n,t]] are computed for all N*L rows side by side instead of
 * step after step (LSTMGenerator.forward models/generator.py:40-60 + the one-hot pick of models/SeqGAN_G.py:35-36).
 * Same results as tg_gen_teacher_forced to fp32 rounding.  E = H = 32: no workspace (may be NULL); other shapes run
 * tg_gen_teacher_forced_ws's GEMM formulation in tg_gen_pg_forward_workspace_bytes(w, N, L) bytes. */
size_t tg_gen_pg_forward_workspace_bytes(const TgGenWeights* w, int N, int L);  /* 0 for E = H = 32 */
int tg_gen_pg_forward(const TgGenWeights* w, const int64_t* inp,
                      const int64_t* target, int N, int L, float* logp_out,
                      float* lse_out, float* states_out, void* workspace,
                      size_t workspace_bytes, void* stream);
int tg_gen_pg_backward(const TgGenWeights* w, const int64_t* inp,
                       const int64_t* target, const float* coeff,
                       const float* lse, const float* states, int N, int L,
                       int64_t padding_idx, const TgGenGrads* grads,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * LeakGAN Monte-Carlo completions.
 * Replaces ROLLOUT.rollout_mc_search_leakgan (utils/rollout.py:57-126) for every (rollout index i, given_num) pair
 * of ROLLOUT.get_reward_leakgan's loops (utils/rollout.py:172-175) in one call: per position the discriminator's
 * get_feature over the tokens so far (models/discriminator.py:51-64, zero padded as rollout.py:75-76 / 105 have it),
 * LeakGAN_G.forward(idx, token, ..., feature, real_goal, train=True) (models/LeakGAN_G.py:52-102: manager LSTM,
 * worker LSTM, work2goal(h_w).view(V,K) @ normalize(goal2goal(real_goal)), temperature, log_softmax), the goal
 * bookkeeping (rollout.py:89-93, 117-121) and torch.multinomial(exp(out), 1) as the canonical inverse-CDF draw.
 * LeakGAN_G parameters: models/LeakGAN_G.py:40-48 (state_dict keys embeddings.weight, worker.*_l0, manager.*_l0,
 * work2goal.*, mana2goal.*, goal2goal.weight, goal_init); G = goal_out_size must equal the scorer's F.
 *   sentences   [B,L]       int64
 *   given_nums  [T] (HOST)  given_num of column t (get_reward_leakgan: t*step_size + 1; one value for a single
 *                           rollout_mc_search_leakgan call), each in [1, L]; T <= 64
 *   uniforms    [R,T,L,B]   fp32 in [0,1): the draw for rollout i, column t, position pos, sentence b is
 *                           uniforms[((i*T + t)*L + pos)*B + b] (pos >= given_nums[t] used)
 *   forced      [R,T,B,L]   int64 or NULL: replay mode (the drawn token is written, the forced one is fed on, to the
 *                           worker and to the feature extractor)
 *   completions [R,T,B,L]   int64; only i in [i_lo, i_hi) is written
 *   workspace   tg_leak_rollout_workspace_bytes(g, d, B, L, T, i_hi - i_lo) bytes
 * Supported: hidden size a multiple of 32 up to 128, goal_size <= 64, V <= 65536, B <= goal_init's rows; the scorer
 * as in tg_dis_score (d->precision selects its arithmetic).  The generator side is fp32 MFMA / fp64 except the
 * work2goal projection -- outputs[n, v] = z[n, :] . [work2goal.weight rows of v | work2goal.bias of v], 2 (K H + K) V
 * flops per row and position, nearly all of the generator's -- which g->precision selects: TG_GEN_FP32 exact, or
 * TG_GEN_SPLIT_F16 as three fp16 products of (hi, lo) halves with fp32 accumulation (22 mantissa bits per operand,
 * as in tg_gen_rollout; tokens then agree with the exact path up to ties of the inverse-CDF draw).
 * Streams: the call is ordered on `stream` like every other entry point (inputs must be ready on it, the completions
 * are complete on it), but part of its work runs on three streams the library creates per device on first use and
 * keeps (the feature passes and the manager's cells run beside the tokens' chain; events tie them to `stream`).
 * Calls on one device take turns while they enqueue (a mutex; the GPU work of two calls may overlap if their streams
 * allow it).  The environment variable TG_LEAK_ONE_STREAM=1 keeps everything on `stream` (same results, slower).
 * Not exercised under stream capture. */
typedef struct TgLeakWeights {
    const float* emb;       /* embeddings.weight     [V,E]    */
    const float* w_wih;     /* worker.weight_ih_l0   [4H,E]   */
    const float* w_whh;     /* worker.weight_hh_l0   [4H,H]   */
    const float* w_bih;     /* worker.bias_ih_l0     [4H]     */
    const float* w_bhh;     /* worker.bias_hh_l0     [4H]     */
    const float* m_wih;     /* manager.weight_ih_l0  [4H,G]   */
    const float* m_whh;     /* manager.weight_hh_l0  [4H,H]   */
    const float* m_bih;     /* manager.bias_ih_l0    [4H]     */
    const float* m_bhh;     /* manager.bias_hh_l0    [4H]     */
    const float* w2g_w;     /* work2goal.weight      [V*K,H]  */
    const float* w2g_b;     /* work2goal.bias        [V*K]    */
    const float* m2g_w;     /* mana2goal.weight      [G,H]    */
    const float* m2g_b;     /* mana2goal.bias        [G]      */
    const float* g2g_w;     /* goal2goal.weight      [K,G]    */
    const float* goal_init; /* goal_init             [B_init,G] */
    int32_t V, E, H, G, K;  /* K = goal_size, G = goal_out_size */
    int32_t step_size;      /* LeakGAN_G.step_size (models/LeakGAN_G.py:36) */
    int32_t B_init;         /* rows of goal_init (cfg.batch_size at construction, models/LeakGAN_G.py:48) */
    float temperature;      /* LeakGAN_G.temperature (1.5, :38): applied at positions 0 and 1 */
    int32_t precision;      /* TG_GEN_FP32 or TG_GEN_SPLIT_F16: arithmetic of the work2goal projection (ABI 0.4) */
} TgLeakWeights;
size_t tg_leak_rollout_workspace_bytes(const TgLeakWeights* g, const TgDisWeights* d, int B, int L, int T,
                                       int n_i /* rollout indices of the call: i_hi - i_lo */);
int tg_leak_rollout(const TgLeakWeights* g, const TgDisWeights* d, const int64_t* sentences,
                    int B, int L, int R, const int32_t* given_nums, int T, int i_lo, int i_hi,
                    const float* uniforms, const int64_t* forced, int64_t* completions,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Reward reshape + mean.
 * Replaces torch.mean(rewards.view(B, L, R), dim=-1) (utils/rollout.py:153):
 * a reinterpretation of the flat [R*L, B] buffer, not a permute:
 *   out[b,t] = mean_{r<R} raw_flat[(b*L + t)*R + r]. */
int tg_reward_reduce(const float* raw, int B, int L, int R, float* out,
                     void* stream);

/* ---------------------------------------------------------------------------
 * Exchange of the raw reward rows between ranks (the reference is single-process and has no
 * counterpart: its rewards tensor, utils/rollout.py:138-149, is filled by one process).  Rank r has written rows
 * [r*rows_per_rank, (r+1)*rows_per_rank) of raw [world*rows_per_rank, B]; after the call (in stream order) every
 * rank holds all rows: one in-place ncclAllGather over xGMI on `rccl_comm` (an ncclComm_t owned by the caller).
 * RCCL is not linked into this library: the symbols are resolved from the RCCL copy the calling process has loaded.
 * The Python shim does the same exchange with torch.distributed.all_gather_into_tensor (dist.py). */
int tg_allgather_rows(float* raw, size_t rows_per_rank, size_t B, void* rccl_comm, void* stream);

/* ---------------------------------------------------------------------------
 * Measurement hook (no counterpart in the reference): when enabled, every
 * kernel launch is bracketed by hipEvents on its launch stream.
 * tg_profile_read() synchronises on the recorded events of one kernel id,
 * returns their summed duration and count, and forgets them. */
#define TG_KERNEL_GEN_TEACHER 0   /* k_gen<false>: teacher-forced / prefix states */
#define TG_KERNEL_GEN_SAMPLE 1    /* k_gen<true>:  rollout / sample               */
#define TG_KERNEL_DIS_CONV 2      /* k_dis_conv                                   */
#define TG_KERNEL_DIS_HIGHWAY 3   /* k_dis_highway                                */
#define TG_KERNEL_DIS_FINAL 4     /* k_dis_final                                  */
#define TG_KERNEL_REDUCE 5        /* k_reward_reduce                              */
#define TG_KERNEL_DIS_PREFIX 6    /* k_dis_conv<store-all> + k_dis_prefix_scan    */
#define TG_KERNEL_PG_BACKWARD 7   /* k_pg_prep .. k_pg_reduce (one bracket)       */
#define TG_KERNEL_PG_FORWARD 8    /* k_pg_fwd_rows                                */
#define TG_KERNEL_GENW_CELL 9     /* k_genw_gemm<cell>: gates GEMM + state update  */
#define TG_KERNEL_GENW_PROJ 10    /* k_genw_gemm<proj>: vocabulary GEMM + records  */
#define TG_KERNEL_GENW_DRAW 11    /* k_genw_draw                                   */
#define TG_KERNEL_LEAK_DRAW 12    /* k_leak_draw                                   */
#define TG_KERNEL_LEAK_CELLS 13   /* k_leak_cells                                  */
#define TG_KERNEL_LEAK_PROJ 14    /* k_leak_proj                                   */
#define TG_KERNEL_PGW_FORWARD 15  /* GEMM-path teacher-forced forward (tg_pgw.hip, one bracket)  */
#define TG_KERNEL_PGW_BACKWARD 16 /* GEMM-path policy-gradient backward (tg_pgw.hip) without the vocabulary products */
#define TG_KERNEL_PGW_PROJ 17     /* the GEMM path's logits = h W_out^T + b (forward, and again in the backward): k_fgemm / k_gemm32 */
#define TG_KERNEL_PGW_PROJ_DH 18  /* dh_proj = dlogits . W_out   */
#define TG_KERNEL_PGW_PROJ_DW 19  /* dW_out = dlogits^T . h      */
void tg_profile_enable(int on);
int tg_profile_read(int kernel_id, double* total_ms, int* count);

#ifdef __cplusplus
}
#endif
#endif /* TEXTGAN_HIP_H */
