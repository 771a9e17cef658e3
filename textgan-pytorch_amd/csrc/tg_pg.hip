// Backward of the policy-gradient loss (SeqGAN_G.batchPGLoss models/SeqGAN_G.py:21-39, MaliGAN_G.adv_loss
// models/MaliGAN_G.py:21-39, SentiGAN_G.batchPGLoss models/SentiGAN_G.py:47-65) for the E = H = 32 generator.
//
// The reference builds out = log_softmax(lstm2out(lstm(emb(inp)))) as a [B*L, V] matrix, multiplies it with a
// [B, L, V] one-hot of the target and lets autograd walk back through both.  Here the forward is the teacher-forced
// generator kernel (tg_gen.hip: logp, lse and the (h, c) states, no [B*L, V] matrix), and the backward recomputes
// the logits tile by tile on the matrix cores:
//     dlogits[n,t,v] = coeff[n,t] * (1[v == target[n,t]] - exp(logits[n,t,v] - lse[n,t])),   coeff = dLoss/dlogp
//   k_pg_rows   one workgroup per 32 (n,t) rows, sweeps the vocabulary:   dh_proj = dlogits . W_out
//   k_pg_bptt_vocab, one launch, two kinds of workgroup:
//     vocab     one workgroup per 32 vocabulary entries, sweeps the rows:  dW_out = dlogits^T . h,  db_out
//     bptt      one workgroup per 32 sequences, t = L-1 .. 0: gates recomputed from the saved states, the
//               recurrent and input gradients, and the weight gradients accumulated in MFMA registers
//   k_pg_reduce fixed-order sum of the per-workgroup weight-gradient partials
// Everything is fp32 on v_mfma_f32_32x32x2_f32 (exact products, fp32 accumulation).  Every MFMA operand comes
// straight from registers or from a contiguous load: the k index of a product is a free permutation as long as A
// and B agree, so "the 16 values this lane already holds" is made the k order wherever possible.
#include "tg_common.h"
#include "tg_pgw.h"

namespace {

constexpr int kHid = 32;     // E = H = 32
constexpr int kLds = 36;     // LDS row stride (floats): 16-B aligned rows, conflict-free column reads
constexpr int kPart = 2 * 128 * 32 + 128;  // per-workgroup partial: dW_ih, dW_hh, db

struct PgArgs {
    TgGenWeights w;
    TgGenGrads g;
    const int64_t* inp;
    const int64_t* target;
    const float* coeff;
    const float* lse;
    const float* states;  // [L+1, N, 2, 32]
    int N, L, NL, NLp, Vp, NTv;
    long long padding_idx;
    // workspace
    float* woutT;     // [32][Vp]   W_out transposed, zero columns past V
    float* boutp;     // [Vp]
    float* lse_p;     // [NLp]      per (n,t) row, padded with coeff = 0 rows
    float* coeff_p;   // [NLp]
    int* tgt_p;       // [NLp]
    unsigned* hoff_p; // [NLp]      offset of h(n,t) in states
    float* dhproj;    // [NLp][32]
    float* part;      // [ceil(N/32)][kPart]
};

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

__global__ __launch_bounds__(256) void k_pg_prep(const PgArgs a) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int V = a.w.V;
    for (size_t i = i0; i < (size_t)kHid * a.Vp; i += stride) {
        const int k = (int)(i / a.Vp), v = (int)(i - (size_t)k * a.Vp);
        a.woutT[i] = v < V ? a.w.w_out[(size_t)v * kHid + k] : 0.f;
    }
    for (size_t i = i0; i < (size_t)a.Vp; i += stride) a.boutp[i] = (int)i < V ? a.w.b_out[i] : 0.f;
    for (size_t i = i0; i < (size_t)a.NLp; i += stride) {
        const bool ok = (int)i < a.NL;
        const int r = ok ? (int)i : a.NL - 1;
        const int n = r / a.L, t = r - n * a.L;
        a.lse_p[i] = ok ? a.lse[r] : 0.f;
        a.coeff_p[i] = ok ? a.coeff[r] : 0.f;
        a.tgt_p[i] = ok ? clamp_tok(a.target[r], V) : -1;
        a.hoff_p[i] = (unsigned)(((size_t)(t + 1) * a.N + n) * 2 * kHid);
    }
    for (size_t i = i0; i < (size_t)V * kHid; i += stride) a.g.emb[i] = 0.f;
}

// dh_proj[row][k] = sum_v dlogits[row][v] * W_out[v][k].  Workgroup = 32 rows, wave w takes the 32-entry vocabulary
// tiles j = w, w+4, ...  Logits tile: A = W_out rows (k = 16hh + s), B = h of the rows (registers).  The tile lands
// as lane (row, hh) reg r <-> v = 32j + (r&3) + 8(r>>2) + 4hh; that order is the k order of the second product, whose
// A operand is then four 16-B loads from the transposed copy of W_out.
__global__ __launch_bounds__(256) void k_pg_rows(const PgArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][32][kLds];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, nl = lane & 31, hh = lane >> 5;
    const int V = a.w.V;
    const int row = blockIdx.x * 32 + nl;  // < NLp
    float hb[16];
    load16(a.states + a.hoff_p[row] + 16 * hh, hb);
    const float my_lse = a.lse_p[row], my_c = a.coeff_p[row];
    const int my_t = a.tgt_p[row];
    struct Frag { float a1[16]; f32x4 a2[4]; f32x4 b[4]; };
    auto load_frag = [&](int j, Frag& f) {
        load16(a.w.w_out + (size_t)tg_min(32 * j + nl, V - 1) * kHid + 16 * hh, f.a1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f.a2[q] = *reinterpret_cast<const f32x4*>(a.woutT + (size_t)nl * a.Vp + 32 * j + 8 * q + 4 * hh);
            f.b[q] = *reinterpret_cast<const f32x4*>(a.boutp + 32 * j + 8 * q + 4 * hh);
        }
    };
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    if (w < a.NTv) {
        Frag cur;
        load_frag(w, cur);
        for (int j = w; j < a.NTv; j += 4) {
            Frag nxt;
            load_frag(tg_min(j + 4, a.NTv - 1), nxt);  // clamped, not skipped: keeps the body one block
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = TG_MFMA32(cur.a1[s], hb[s], acc);
            float dl[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float p = tg_expf(acc[r] + cur.b[r >> 2][r & 3] - my_lse);
                const float d = my_c * ((v == my_t ? 1.f : 0.f) - p);
                dl[r] = v < V ? d : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2 = TG_MFMA32(cur.a2[r >> 2][r & 3], dl[r], acc2);
            cur = nxt;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]};
        *reinterpret_cast<f32x4*>(&red[w][nl][8 * q + 4 * hh]) = v;
    }
    __syncthreads();
    {
        const int n2 = tid >> 3, kc = tid & 7;
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][n2][4 * kc]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += *reinterpret_cast<const f32x4*>(&red[ww][n2][4 * kc]);
        *reinterpret_cast<f32x4*>(a.dhproj + ((size_t)blockIdx.x * 32 + n2) * kHid + 4 * kc) = s;
    }
}

// dW_out[v][k] = sum_rows dlogits[row][v] * h[row][k], db_out[v] = sum_rows dlogits[row][v].  One workgroup per 32
// vocabulary entries (its W_out rows stay in registers as the B operand); its four waves share the sweep over the
// rows, 32 at a time, and meet in LDS at the end.  The
// logits tile lands as lane (v, hh) reg r <-> row 32rb + (r&3) + 8(r>>2) + 4hh, again the k order of the second
// product; its A operand h[row][k] is read column-wise (16 dword loads, each 128 B across the lanes).
__device__ __forceinline__ void pg_vocab_body(const PgArgs& a, const int j, float (*red)[32][kLds], float (*redb)[kLds]) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, vl = lane & 31, hh = lane >> 5;
    const int V = a.w.V;
    const int v = 32 * j + vl;
    const bool vok = v < V;
    const int vc = tg_min(v, V - 1);
    float wb[16];
    load16(a.w.w_out + (size_t)vc * kHid + 16 * hh, wb);
    const float bias = a.w.b_out[vc];
    struct RB { float a1[16]; float a2[16]; f32x4 lse[4], c[4]; int t[16]; };
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_rb = [&](int rb, RB& f) {
        load16(a.states + a.hoff_p[32 * rb + vl] + 16 * hh, f.a1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r0 = 32 * rb + 8 * q + 4 * hh;
            const u32x4 off = *reinterpret_cast<const u32x4*>(a.hoff_p + r0);
            const i32x4 tt = *reinterpret_cast<const i32x4*>(a.tgt_p + r0);
            f.lse[q] = *reinterpret_cast<const f32x4*>(a.lse_p + r0);
            f.c[q] = *reinterpret_cast<const f32x4*>(a.coeff_p + r0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f.a2[4 * q + e] = a.states[(size_t)off[e] + vl];
                f.t[4 * q + e] = tt[e];
            }
        }
    };
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    float dbacc = 0.f;
    const int NRB = a.NLp >> 5;
    RB cur;
    load_rb(tg_min(w, NRB - 1), cur);
    for (int rb = w; rb < NRB; rb += 4) {  // the four waves share the row sweep
        RB nxt;
        load_rb(tg_min(rb + 4, NRB - 1), nxt);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = TG_MFMA32(cur.a1[s], wb[s], acc);
        float dl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = tg_expf(acc[r] + bias - cur.lse[r >> 2][r & 3]);
            const float d = cur.c[r >> 2][r & 3] * ((v == cur.t[r] ? 1.f : 0.f) - p);
            dl[r] = vok ? d : 0.f;
            dbacc += dl[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2 = TG_MFMA32(cur.a2[r], dl[r], acc2);
        cur = nxt;
    }
    const float db = dbacc + __shfl_xor(dbacc, 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 o = {acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]};
        *reinterpret_cast<f32x4*>(&red[w][vl][8 * q + 4 * hh]) = o;
    }
    if (hh == 0) redb[w][vl] = db;
    __syncthreads();
    {   // fixed-order sum over the waves; thread -> (entry tid>>3, four columns)
        const int v2 = tid >> 3, kc = tid & 7, vv = 32 * j + v2;
        f32x4 o = *reinterpret_cast<const f32x4*>(&red[0][v2][4 * kc]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) o += *reinterpret_cast<const f32x4*>(&red[ww][v2][4 * kc]);
        if (vv < V) {
            *reinterpret_cast<f32x4*>(a.g.w_out + (size_t)vv * kHid + 4 * kc) = o;
            if (kc == 0) a.g.b_out[vv] = redb[0][v2] + redb[1][v2] + redb[2][v2] + redb[3][v2];
        }
    }
}

// Back-propagation through time.  Workgroup = 32 sequences; wave w owns hidden units 8w..8w+7 (its 32 gate rows),
// lane (n, hh) the gates i,f,g,o of units 8w + 4hh + 0..3 of sequence n, exactly as the forward cell does.
//   gate pre-activations: recomputed from x_t and the saved h_{t-1} (32 MFMAs), c_t from the saved c_{t-1}
//   da (the 16 pre-activation gradients of the lane) is used three ways:
//     - as the B operand, straight from registers, of  W_hh^T . da  and  W_ih^T . da  (k = the lane's own 16 gate
//       rows; A = the matching columns of W_hh / W_ih, loaded once); the four waves' partial sums meet in LDS;
//     - transposed through LDS as the A operand of  da^T . x_t  and  da^T . h_{t-1}  (k = the 32 sequences),
//       accumulated over all steps in two MFMA register tiles per wave;
//     - summed over the sequences for the bias gradient.
__device__ __forceinline__ void pg_bptt_body(const PgArgs& a, const int blk, float (*part_h)[32][kLds],
                                             float (*part_x)[32][kLds], float (*daT)[32][kLds], float (*xs)[kLds],
                                             float (*hs)[kLds]) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, nl = lane & 31, hh = lane >> 5;
    const int N = a.N, L = a.L, V = a.w.V;
    const int n = blk * 32 + nl;
    const bool valid = n < N;
    const int nc = tg_min(n, N - 1);

    const int i = nl;  // as an MFMA row / column index of this lane
    const int grow_f = (i >> 3) * kHid + 8 * w + (i & 7);
    float wih[16], whh[16], atih[16], athh[16];
    load16(a.w.w_ih + (size_t)grow_f * kHid + 16 * hh, wih);
    load16(a.w.w_hh + (size_t)grow_f * kHid + 16 * hh, whh);
    f32x16 gbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = (r >> 2) * kHid + 8 * w + 4 * hh + (r & 3);
        gbias[r] = a.w.b_ih[gr] + a.w.b_hh[gr];
        atih[r] = a.w.w_ih[(size_t)gr * kHid + i];
        athh[r] = a.w.w_hh[(size_t)gr * kHid + i];
    }
    f32x16 dWih, dWhh;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dWih[r] = 0.f; dWhh[r] = 0.f; }
    float dbacc = 0.f;
    f32x4 dc = {0.f, 0.f, 0.f, 0.f}, dhrec = {0.f, 0.f, 0.f, 0.f};
    // the embedding-gradient scatter: thread -> (sequence tid>>3, four columns)
    const int n2 = tid >> 3, kc = tid & 7;
    const int row2 = blk * 32 + n2;

    // operands of a step, fetched one step ahead (the token two steps ahead): the step itself is a chain of
    // dependent MFMAs and two barriers, there is nothing else to hide a gather behind
    struct StepIn { float xb[16], hb[16]; f32x4 cprev, dh; };
    auto load_step = [&](int t, int tok, StepIn& o) {
        load16(a.w.emb + (size_t)tok * kHid + 16 * hh, o.xb);
        const float* sp = a.states + ((size_t)t * N + nc) * 2 * kHid;
        load16(sp + 16 * hh, o.hb);
        o.cprev = *reinterpret_cast<const f32x4*>(sp + kHid + 8 * w + 4 * hh);
        o.dh = *reinterpret_cast<const f32x4*>(a.dhproj + ((size_t)nc * L + t) * kHid + 8 * w + 4 * hh);
    };
    StepIn cur;
    load_step(L - 1, clamp_tok(a.inp[(size_t)nc * L + L - 1], V), cur);
    int tok_nxt = clamp_tok(a.inp[(size_t)nc * L + tg_max(L - 2, 0)], V);
    for (int t = L - 1; t >= 0; --t) {
        StepIn nxt;
        load_step(tg_max(t - 1, 0), tok_nxt, nxt);
        tok_nxt = clamp_tok(a.inp[(size_t)nc * L + tg_max(t - 2, 0)], V);
        const float (&xb)[16] = cur.xb;
        const float (&hb)[16] = cur.hb;
        const f32x4 cprev = cur.cprev;
        f32x4 dh = cur.dh;
        dh += dhrec;
        f32x16 acc = gbias;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = TG_MFMA32(wih[s], xb[s], acc);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = TG_MFMA32(whh[s], hb[s], acc);
        float da[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = tg_sigmoid(acc[e]);
            const float fg = tg_sigmoid(acc[4 + e]);
            const float gg = tg_tanh(acc[8 + e]);
            const float og = tg_sigmoid(acc[12 + e]);
            const float cn = fg * cprev[e] + ig * gg;
            const float tc = tg_tanh(cn);
            const float dct = dc[e] + dh[e] * og * (1.f - tc * tc);
            dc[e] = dct * fg;
            da[e] = valid ? dct * gg * ig * (1.f - ig) : 0.f;
            da[4 + e] = valid ? dct * cprev[e] * fg * (1.f - fg) : 0.f;
            da[8 + e] = valid ? dct * ig * (1.f - gg * gg) : 0.f;
            da[12 + e] = valid ? dh[e] * tc * og * (1.f - og) : 0.f;
        }
        f32x16 ph, px;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ph[r] = 0.f; px[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ph = TG_MFMA32(athh[r], da[r], ph);
            px = TG_MFMA32(atih[r], da[r], px);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 vh = {ph[4 * q], ph[4 * q + 1], ph[4 * q + 2], ph[4 * q + 3]};
            const f32x4 vx = {px[4 * q], px[4 * q + 1], px[4 * q + 2], px[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&part_h[w][nl][8 * q + 4 * hh]) = vh;
            *reinterpret_cast<f32x4*>(&part_x[w][nl][8 * q + 4 * hh]) = vx;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) daT[w][16 * hh + r][nl] = da[r];
        if (w < 2) {
            float(*dst)[kLds] = w == 0 ? xs : hs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = w == 0 ? xb[4 * q + e] : hb[4 * q + e];
                *reinterpret_cast<f32x4*>(&dst[nl][16 * hh + 4 * q]) = v;
            }
        }
        __syncthreads();
        // recurrent gradient for step t-1: this lane's four units, summed over the waves' partials
        dhrec = *reinterpret_cast<const f32x4*>(&part_h[0][nl][8 * w + 4 * hh]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) dhrec += *reinterpret_cast<const f32x4*>(&part_h[ww][nl][8 * w + 4 * hh]);
        // embedding gradient (nn.Embedding leaves the padding_idx row at zero)
        if (row2 < N) {
            const int64_t raw = a.inp[(size_t)row2 * L + t];
            if (raw != a.padding_idx) {
                f32x4 s = *reinterpret_cast<const f32x4*>(&part_x[0][n2][4 * kc]);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) s += *reinterpret_cast<const f32x4*>(&part_x[ww][n2][4 * kc]);
                float* d = a.g.emb + (size_t)clamp_tok(raw, V) * kHid + 4 * kc;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(d + e, s[e]);
            }
        }
        // weight gradients: A = da^T (gate row i of this wave, k = sequence 16hh + s), B = x_t / h_{t-1}
        float aT[16], bx[16], bh[16];
        load16(&daT[w][i][16 * hh], aT);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dbacc += aT[s];
            bx[s] = xs[16 * hh + s][i];
            bh[s] = hs[16 * hh + s][i];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dWih = TG_MFMA32(aT[s], bx[s], dWih);
            dWhh = TG_MFMA32(aT[s], bh[s], dWhh);
        }
        cur = nxt;
        __syncthreads();
    }
    // partials: D lane (column c = i, hh) reg r <-> local gate row jl = (r&3) + 8(r>>2) + 4hh of this wave, which is
    // (producer half jl>>4, producer reg jl&15) of the da layout
    float* part = a.part + (size_t)blk * kPart;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jl = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int rj = jl & 15, hj = jl >> 4;
        const int gr = (rj >> 2) * kHid + 8 * w + 4 * hj + (rj & 3);
        part[gr * kHid + i] = dWih[r];
        part[128 * kHid + gr * kHid + i] = dWhh[r];
    }
    const float db = dbacc + __shfl_xor(dbacc, 32);
    if (hh == 0) {
        const int rj = i & 15, hj = i >> 4;
        const int gr = (rj >> 2) * kHid + 8 * w + 4 * hj + (rj & 3);
        part[2 * 128 * kHid + gr] = db;
    }
}

// One launch for the two independent tails of the backward: the first nwg workgroups walk back through time
// (two workgroups at batch 64, the long pole), the others sweep the rows for dW_out / db_out beside them.
__global__ __launch_bounds__(256) void k_pg_bptt_vocab(const PgArgs a, const int nwg) {
    __shared__ __attribute__((aligned(16))) float part_h[4][32][kLds];
    __shared__ __attribute__((aligned(16))) float part_x[4][32][kLds];
    __shared__ __attribute__((aligned(16))) float daT[4][32][kLds];
    __shared__ __attribute__((aligned(16))) float xs[32][kLds];
    __shared__ __attribute__((aligned(16))) float hs[32][kLds];
    if ((int)blockIdx.x < nwg) pg_bptt_body(a, blockIdx.x, part_h, part_x, daT, xs, hs);
    else pg_vocab_body(a, blockIdx.x - nwg, part_h, xs);
}

// Forward for few sequences (an adversarial batch is 64 x L): the persistent generator kernel walks the vocabulary
// once per step and sequence tile, L steps in a row on two workgroups.  Teacher forcing does not need that order:
// the states come from the cell-only pass of tg_gen_teacher_forced, and the projection, its log-sum-exp and the
// target's logit are computed here for all N*L rows side by side -- 32 rows per workgroup, the four waves sharing
// the vocabulary sweep, (max, sum) kept per lane and merged in fp64 at the end.
__global__ __launch_bounds__(256) void k_pg_fwd_rows(const TgGenWeights wt, const int64_t* __restrict__ target,
                                                     const float* __restrict__ states, int N, int L,
                                                     float* __restrict__ logp_out, float* __restrict__ lse_out) {
    __shared__ float sm_m[8][32], sm_s[8][32], sm_t[8][32];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, nl = lane & 31, hh = lane >> 5;
    const int V = wt.V, NL = N * L, NTv = (V + 31) >> 5;
    const int row = blockIdx.x * 32 + nl;
    const int rc = tg_min(row, NL - 1);
    const int n = rc / L, t = rc - n * L;
    float hb[16];
    load16(states + ((size_t)(t + 1) * N + n) * 2 * kHid + 16 * hh, hb);
    const int my_t = clamp_tok(target[rc], V);
    struct Frag { float a1[16]; f32x4 b[4]; };
    auto load_frag = [&](int j, Frag& f) {
        load16(wt.w_out + (size_t)tg_min(32 * j + nl, V - 1) * kHid + 16 * hh, f.a1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v0 = 32 * j + 8 * q + 4 * hh;
#pragma unroll
            for (int e = 0; e < 4; ++e) f.b[q][e] = wt.b_out[tg_min(v0 + e, V - 1)];
        }
    };
    float m_run = TG_NEG_BIG, s_run = 0.f, lt = TG_NEG_BIG;
    if (w < NTv) {
        Frag cur;
        load_frag(w, cur);
        for (int j = w; j < NTv; j += 4) {
            Frag nxt;
            load_frag(tg_min(j + 4, NTv - 1), nxt);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = TG_MFMA32(cur.a1[s], hb[s], acc);
            float tmax = TG_NEG_BIG;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float l = acc[r] + cur.b[r >> 2][r & 3];
                acc[r] = v < V ? l : TG_NEG_BIG;
                lt = v == my_t ? l : lt;
                tmax = __builtin_fmaxf(tmax, acc[r]);
            }
            const float m_new = __builtin_fmaxf(m_run, tmax);
            float ts = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ts += tg_expf_neg(acc[r] - m_new);
            s_run = s_run * tg_expf_neg(m_run - m_new) + ts;
            m_run = m_new;
            cur = nxt;
        }
    }
    sm_m[2 * w + hh][nl] = m_run;
    sm_s[2 * w + hh][nl] = s_run;
    sm_t[2 * w + hh][nl] = lt;
    __syncthreads();
    if (tid < 32 && row < NL) {
        float M = sm_m[0][tid], T = sm_t[0][tid];
#pragma unroll
        for (int p = 1; p < 8; ++p) { M = __builtin_fmaxf(M, sm_m[p][tid]); T = __builtin_fmaxf(T, sm_t[p][tid]); }
        double S = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) S += (double)sm_s[p][tid] * exp((double)(sm_m[p][tid] - M));  // empty part: s = 0
        const float lse = (float)((double)M + log(S));
        if (lse_out) lse_out[row] = lse;
        if (logp_out) logp_out[row] = T - lse;
    }
}

__global__ __launch_bounds__(256) void k_pg_reduce(const PgArgs a, int nwg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kPart) return;
    float s = 0.f;
    for (int b = 0; b < nwg; ++b) s += a.part[(size_t)b * kPart + idx];
    if (idx < 128 * kHid) a.g.w_ih[idx] = s;
    else if (idx < 2 * 128 * kHid) a.g.w_hh[idx - 128 * kHid] = s;
    else {
        a.g.b_ih[idx - 2 * 128 * kHid] = s;
        a.g.b_hh[idx - 2 * 128 * kHid] = s;
    }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct PgLayout {
    size_t woutT, boutp, lse_p, coeff_p, tgt_p, hoff_p, dhproj, part, total;
    int NL, NLp, Vp, NTv, nwg;
};

PgLayout make_layout(int V, int N, int L) {
    PgLayout p;
    p.NL = N * L;
    p.NLp = (p.NL + 31) / 32 * 32;
    p.Vp = (V + 31) / 32 * 32;
    p.NTv = p.Vp / 32;
    p.nwg = (N + 31) / 32;
    size_t o = 0;
    p.woutT = o; o += align256(sizeof(float) * kHid * p.Vp);
    p.boutp = o; o += align256(sizeof(float) * p.Vp);
    p.lse_p = o; o += align256(sizeof(float) * p.NLp);
    p.coeff_p = o; o += align256(sizeof(float) * p.NLp);
    p.tgt_p = o; o += align256(sizeof(int) * p.NLp);
    p.hoff_p = o; o += align256(sizeof(unsigned) * p.NLp);
    p.dhproj = o; o += align256(sizeof(float) * kHid * p.NLp);
    p.part = o; o += align256(sizeof(float) * kPart * p.nwg);
    p.total = o;
    return p;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("%s launch: %s", what, hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

bool shape_ok(const TgGenWeights* w, int N, int L) {
    // hoff_p holds float offsets into states as 32-bit values
    return w && w->E == kHid && w->H == kHid && w->V >= 1 && N >= 1 && L >= 1 &&
           (size_t)(L + 1) * N * 2 * kHid < ((size_t)1 << 32) && (size_t)N * L < ((size_t)1 << 31) - 64;
}

}  // namespace

static bool pg_wide(const TgGenWeights* w) { return w && !(w->E == kHid && w->H == kHid) && tg_pgw_shape_ok(w); }
// beyond E = H = 32, w->precision selects the arithmetic of the three vocabulary products (TG_GEN_SPLIT_F16: split-fp16
// GEMMs, TG_GEN_FP32: exact fp32 MFMA); everything else is fp32 in both
static bool pg_split(const TgGenWeights* w) { return w->precision == TG_GEN_SPLIT_F16; }

extern "C" size_t tg_gen_pg_forward_workspace_bytes(const TgGenWeights* w, int N, int L) {
    if (!pg_wide(w) || N < 1 || L < 1) return 0;
    return tg_pgw_forward_workspace_bytes(w, N, L, false, true, pg_split(w));
}

extern "C" int tg_gen_pg_forward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, int N, int L,
                                 float* logp_out, float* lse_out, float* states_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (pg_wide(w)) {  // any other E, H: exact-fp32 GEMMs (tg_pgw.hip)
        if (!w->emb || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out || !inp || !target ||
            !states_out || N < 0 || L < 1 || (size_t)N * L >= ((size_t)1 << 31) - 256) {
            tg_set_error("tg_gen_pg_forward: bad arguments (N=%d L=%d; weights, inp, target and states_out are required)", N, L);
            return TG_ERR_ARG;
        }
        if (N == 0) return TG_OK;
        return tg_pgw_forward(w, inp, target, N, L, logp_out, lse_out, nullptr, states_out, workspace, workspace_bytes,
                              (hipStream_t)stream, nullptr, pg_split(w));
    }
    if (!w || w->E != kHid || w->H != kHid) {
        tg_set_error("tg_gen_pg_forward: generator shape not supported (E, H multiples of 32 up to 1024)");
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    if (!target || !states_out || (size_t)N * L >= ((size_t)1 << 31) - 64) {
        tg_set_error("tg_gen_pg_forward: bad arguments (N=%d L=%d, target and states_out are required)", N, L);
        return TG_ERR_ARG;
    }
    // states only: the cell pass of the generator kernel, no vocabulary projection (checks w, inp, N, L)
    int rc = tg_gen_teacher_forced(w, inp, nullptr, N, L, nullptr, nullptr, nullptr, states_out, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    tg_prof_begin(TG_KERNEL_PG_FORWARD, s);
    hipLaunchKernelGGL(k_pg_fwd_rows, dim3((N * L + 31) / 32), dim3(256), 0, s, *w, target, states_out, N, L, logp_out,
                       lse_out);
    tg_prof_end(s);
    return check_launch("k_pg_fwd_rows");
}

extern "C" size_t tg_gen_pg_backward_workspace_bytes(const TgGenWeights* w, int N, int L) {
    if (pg_wide(w) && N >= 1 && L >= 1) return tg_pgw_backward_workspace_bytes(w, N, L, pg_split(w));
    if (!shape_ok(w, N, L)) return 0;
    return make_layout(w->V, N, L).total;
}

extern "C" int tg_gen_pg_backward(const TgGenWeights* w, const int64_t* inp, const int64_t* target, const float* coeff,
                                  const float* lse, const float* states, int N, int L, int64_t padding_idx,
                                  const TgGenGrads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !w->emb || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out) {
        tg_set_error("tg_gen_pg_backward: null generator weights");
        return TG_ERR_ARG;
    }
    if (pg_wide(w)) {  // any other E, H: exact-fp32 GEMMs (tg_pgw.hip)
        if (N == 0) return TG_OK;
        if (N < 0 || L < 1 || (size_t)N * L >= ((size_t)1 << 31) - 256 || !inp || !target || !coeff || !lse || !states ||
            !grads || !grads->emb || !grads->w_ih || !grads->w_hh || !grads->b_ih || !grads->b_hh || !grads->w_out ||
            !grads->b_out) {
            tg_set_error("tg_gen_pg_backward: bad arguments (N=%d L=%d)", N, L);
            return TG_ERR_ARG;
        }
        return tg_pgw_backward(w, inp, target, coeff, lse, states, N, L, padding_idx, grads, workspace, workspace_bytes,
                               (hipStream_t)stream, pg_split(w));
    }
    if (w->E != kHid || w->H != kHid) {
        tg_set_error("tg_gen_pg_backward: generator shape E=%d H=%d not supported (multiples of 32 up to 1024)", w->E, w->H);
        return TG_ERR_ARG;
    }
    if (N == 0) return TG_OK;
    if (!shape_ok(w, N, L) || !inp || !target || !coeff || !lse || !states || !grads || !grads->emb || !grads->w_ih ||
        !grads->w_hh || !grads->b_ih || !grads->b_hh || !grads->w_out || !grads->b_out) {
        tg_set_error("tg_gen_pg_backward: bad arguments (N=%d L=%d)", N, L);
        return TG_ERR_ARG;
    }
    const PgLayout p = make_layout(w->V, N, L);
    if (!workspace || workspace_bytes < p.total) {
        tg_set_error("tg_gen_pg_backward: workspace of %zu bytes needed, %zu given", p.total, workspace_bytes);
        return TG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PgArgs a = {};
    a.w = *w;
    a.g = *grads;
    a.inp = inp;
    a.target = target;
    a.coeff = coeff;
    a.lse = lse;
    a.states = states;
    a.N = N;
    a.L = L;
    a.NL = p.NL;
    a.NLp = p.NLp;
    a.Vp = p.Vp;
    a.NTv = p.NTv;
    a.padding_idx = padding_idx;
    a.woutT = (float*)(ws + p.woutT);
    a.boutp = (float*)(ws + p.boutp);
    a.lse_p = (float*)(ws + p.lse_p);
    a.coeff_p = (float*)(ws + p.coeff_p);
    a.tgt_p = (int*)(ws + p.tgt_p);
    a.hoff_p = (unsigned*)(ws + p.hoff_p);
    a.dhproj = (float*)(ws + p.dhproj);
    a.part = (float*)(ws + p.part);
    int rc;
    tg_prof_begin(TG_KERNEL_PG_BACKWARD, s);
    hipLaunchKernelGGL(k_pg_prep, dim3(512), dim3(256), 0, s, a);
    if ((rc = check_launch("k_pg_prep"))) return rc;
    hipLaunchKernelGGL(k_pg_rows, dim3(p.NLp / 32), dim3(256), 0, s, a);
    if ((rc = check_launch("k_pg_rows"))) return rc;
    hipLaunchKernelGGL(k_pg_bptt_vocab, dim3(p.nwg + p.NTv), dim3(256), 0, s, a, p.nwg);
    if ((rc = check_launch("k_pg_bptt_vocab"))) return rc;
    hipLaunchKernelGGL(k_pg_reduce, dim3((kPart + 255) / 256), dim3(256), 0, s, a, p.nwg);
    if ((rc = check_launch("k_pg_reduce"))) return rc;
    tg_prof_end(s);
    return TG_OK;
}
