// Microbenchmark: the projection loop of k_gen in isolation (32 rows x V vocab, H = 32), per-feature ablation.
//   bit0 softmax VALU work   bit1 A fragments loaded from an L2-resident W_out (3 tiles ahead)
//   bit2 per-tile record store to LDS   bit3 split-fp16 x3 MFMAs (7 x 32x32x16 f16) instead of 17 x 32x32x2 f32
// hipcc --offload-arch=gfx950 -O3 proj_loop.hip -o proj_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct AFrag { f32x4 w[4]; float bias; };

template <int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k(const float* __restrict__ W, const float* __restrict__ bvec, float* out, int NT, int steps, int V) {
    constexpr bool SM = F & 1, LD = F & 2, REC = F & 4, H16 = F & 8, NOEXP = F & 16, NOSCHED = F & 32, COAL = F & 64;
    __shared__ float2 rec[16 * 8 * 32];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int sp = 2 * w + hh, Ls = 16 * NT;
    const int am = lane & 31, a_sp = 2 * w + ((am >> 2) & 1), a_r = (am & 3) + 4 * (am >> 3);
    float hb[16];
    for (int j = 0; j < 16; ++j) hb[j] = 0.01f * (j + 1) + 1e-4f * tid;
    f16x8 hbh[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) hbh[i][e] = (_Float16)(0.01f * (e + 1 + i));
    f16x8 bone;
    for (int e = 0; e < 8; ++e) bone[e] = (_Float16)((hh == 0 && e < 2) ? 1.f : 0.f);
    const float bonef = hh ? 0.f : 1.f;
    float total = 0.f;
    for (int s = 0; s < steps; ++s) {
        float m_run = -1e30f, c_grp = 0.f;
        auto load_a = [&](int q, AFrag& d) {
            if (LD && COAL) {  // fragment-major copy of W_out: every load instruction of a wave reads 1 KiB contiguous
                const f32x4* p = reinterpret_cast<const f32x4*>(W) + ((size_t)(w * NT + q) * 4) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) d.w[i] = p[i * 64];
                d.bias = bvec[(w * NT + q) * 64 + lane];
            } else if (LD) {
                const int v = a_sp * Ls + 16 * q + a_r;
                const f32x4* p = reinterpret_cast<const f32x4*>(W + (size_t)min(v, V - 1) * 32 + 16 * hh);
#pragma unroll
                for (int i = 0; i < 4; ++i) d.w[i] = p[i];
                d.bias = hh ? 0.f : (v < V ? bvec[v] : -1e30f);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) d.w[i] = f32x4{0.1f * q, 0.2f, 0.3f, 0.4f + i};
                d.bias = 0.5f;
            }
        };
        auto mfma_tile = [&](const AFrag& af) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (H16) {
                f16x8 ah[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ah[i] = __builtin_bit_cast(f16x8, af.w[i]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], hbh[0], acc, 0, 0, 0);  // hi.hi
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], hbh[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], hbh[2], acc, 0, 0, 0);  // hi.lo
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], hbh[3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2], hbh[0], acc, 0, 0, 0);  // lo.hi
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[3], hbh[1], acc, 0, 0, 0);
                f16x8 bf;
#pragma unroll
                for (int e = 0; e < 8; ++e) bf[e] = (_Float16)0.f;
                const auto h2 = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, af.bias);
                bf[0] = h2[0]; bf[1] = h2[1];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, bone, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w[j >> 2][j & 3], hb[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.bias, bonef, acc, 0, 0, 0);
            }
            return acc;
        };
        auto softmax_tile = [&](const f32x16& acc, int q) {
            if (SM) {
                float tmax = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = __builtin_fmaxf(tmax, acc[r]);
                const float m_new = __builtin_fmaxf(m_run, tmax);
                const float keep = ((q & 3) == 0) ? 0.f : 1.f;
                c_grp = c_grp * __builtin_amdgcn_exp2f(m_run - m_new) * keep;
                float ev[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = NOEXP ? (acc[r] - m_new) * 0.5f : __builtin_amdgcn_exp2f(acc[r] - m_new);
                float ts = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ts += ev[r];
                c_grp += ts;
                m_run = m_new;
            } else {
                c_grp += acc[0] + acc[15];
            }
            if (REC) rec[((q >> 2) * 8 + sp) * 32 + n] = make_float2(m_run, c_grp);
        };
        auto interleave = [&]() {
            if (NOSCHED) return;
            if (H16) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 13, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
        };
        AFrag a0, a1, a2;
        f32x16 acc0, acc1;
        load_a(0, a0); load_a(1, a1); load_a(2, a2);
        acc0 = mfma_tile(a0);
        int q = 0;
        for (; q + 6 < NT; q += 6) {
            load_a(q + 3, a0); acc1 = mfma_tile(a1); softmax_tile(acc0, q);     interleave();
            load_a(q + 4, a1); acc0 = mfma_tile(a2); softmax_tile(acc1, q + 1); interleave();
            load_a(q + 5, a2); acc1 = mfma_tile(a0); softmax_tile(acc0, q + 2); interleave();
            load_a(q + 6, a0); acc0 = mfma_tile(a1); softmax_tile(acc1, q + 3); interleave();
            if (q + 7 < NT) load_a(q + 7, a1);
            acc1 = mfma_tile(a2); softmax_tile(acc0, q + 4); interleave();
            if (q + 8 < NT) load_a(q + 8, a2);
            acc0 = mfma_tile(a0); softmax_tile(acc1, q + 5); interleave();
        }
        softmax_tile(acc0, q);
        total += c_grp + m_run;
        if (REC) { __syncthreads(); total += rec[tid].y; __syncthreads(); }
        // feed the result back so steps stay dependent (as h does in the real kernel)
        hb[0] += 1e-9f * total;
        hbh[0][0] += (_Float16)(1e-9f * total);
    }
    out[blockIdx.x * 256 + tid] = total;
}

template <int F>
void run(const float* W, const float* b, float* out, int wgs_per_cu, const char* name) {
    const int V = 5255, NT = 42 + 1, steps = 200;  // 43 tiles: 7 x 6 pipelined + the drained one
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int grid = 256 * wgs_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<F>, dim3(grid), dim3(256), 0, 0, W, b, out, NT, steps, V);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double tiles_per_simd = (double)wgs_per_cu * steps * NT;
    const double ns_tile = ms * 1e6 / tiles_per_simd;
    const double flops = (double)grid * 4 * steps * NT * 2.0 * 32 * 32 * 32;  // useful: 32x32 outputs x K=32
    printf("%-44s waves/SIMD %d: %7.3f ms  %6.0f ns/tile/SIMD (%5.0f cyc@2.4)  %6.1f useful TFLOP/s\n", name,
           wgs_per_cu, ms, ns_tile, ns_tile * 2.4, flops / ms / 1e9);
}

int main() {
    float *W, *b, *out;
    hipMalloc(&W, 1 << 20); hipMalloc(&b, 1 << 16); hipMalloc(&out, 512 * 256 * 4);
    hipMemset(W, 0, 1 << 20); hipMemset(b, 0, 1 << 16);
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0>(W, b, out, wpc, "f32 mfma only");
        run<1>(W, b, out, wpc, "f32 mfma + softmax");
        run<3>(W, b, out, wpc, "f32 mfma + softmax + loads");
        run<7>(W, b, out, wpc, "f32 mfma + softmax + loads + rec");
        run<2>(W, b, out, wpc, "f32 mfma + loads");
        run<66>(W, b, out, wpc, "f32 mfma + coalesced loads");
        run<71>(W, b, out, wpc, "f32 mfma + softmax + coalesced loads + rec");
        run<79>(W, b, out, wpc, "f16x3 mfma + softmax + coalesced loads + rec");
        run<17>(W, b, out, wpc, "f32 mfma + softmax(no exp)");
        run<33>(W, b, out, wpc, "f32 mfma + softmax, compiler schedule");
        run<25>(W, b, out, wpc, "f16x3 mfma + softmax(no exp)");
        run<41>(W, b, out, wpc, "f16x3 mfma + softmax, compiler schedule");
        run<8>(W, b, out, wpc, "f16x3 mfma only");
        run<9>(W, b, out, wpc, "f16x3 mfma + softmax");
        run<11>(W, b, out, wpc, "f16x3 mfma + softmax + loads");
        run<15>(W, b, out, wpc, "f16x3 mfma + softmax + loads + rec");
    }
    return 0;
}
