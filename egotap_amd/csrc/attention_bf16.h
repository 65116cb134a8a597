// Fused softmax attention on the bf16 matrix cores with fp32 tensors in HBM (fast mode of egotap_set_precision):
//   ctx[b, n, h*128:(h+1)*128] = softmax(Q_h K_h^T / sqrt(128)) V_h          (modeling_vit.py:226-252)
// Same dataflow as attention_f32.h -- q|k|v read in place from the fused QKV buffer, S^T = K Q^T with the key on the
// accumulator row and the query on the lane, online softmax as per-lane scalars, O^T += V^T P^T with the probability
// accumulators used as the B operand where they stand -- but every product runs on v_mfma_f32_32x32x16_bf16:
//   NP = 3: each fp32 operand (Q, K, V and the probabilities) is split into hi + lo bf16 in registers and a product is
//           hi*hi + hi*lo + lo*hi (16 significant bits per operand, fp32 accumulate);  NP = 1: plain bf16 rounding.
// Operand maps (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"):
//   S^T:  A = K tile [key][d] from LDS (one ds_read_b128 per 16-d step and image), B = Q^T held in registers for all
//         128 d (hi + lo = 64 VGPRs, what the fp32 kernel spends on Q alone).
//   P^T:  accumulator registers 8s..8s+7 -> bf16 are the B fragment of k-step s; element j of lane half h is key
//         16s + 8(j>>2) + 4h + (j&3).
//   V^T:  A operand [d][key] in exactly that key order, read from the row-major [key][d] LDS image with the 4 x 16
//         transposing read ds_read_b64_tr_b16: block rows 16s + 4h (+8 for elements 4..7), columns 32*dt + 16g.
// LDS images are bf16: K rows of 272 bytes (17 x 16: conflict-free b128 reads), V rows of 320 bytes (the four rows of a
// transposed block sit 64 bytes apart modulo 256).  The next key tile's global loads are in flight during the current
// tile's MFMAs; conversion + LDS write sit between the two barriers, where the CU's second block keeps the pipe busy.
#pragma once
#include "gemm_tn_bf16.h"
#include <math.h>

template <int NW, int NP_>
struct AttnBfCfg {
    static constexpr int NP = NP_, NIMG = NP_ == 3 ? 2 : 1;
    static constexpr int DH = 128, KT = 32, THREADS = 64 * NW;
    static constexpr int KSTR = 136, VSTR = 160;                       // image row strides in bf16
    static constexpr int KIMG = KT * KSTR, VIMG = KT * VSTR;
    static constexpr int TILE_BYTES = NIMG * (KIMG + VIMG) * 2;
    static constexpr int OLD = DH + 4;                                 // output transpose row (floats)
    static constexpr int OUT_BYTES = NW * 32 * OLD * 4;
    static constexpr int LDS_BYTES = TILE_BYTES > OUT_BYTES ? TILE_BYTES : OUT_BYTES;
    static constexpr int V4 = 2 * KT * (DH / 4) / THREADS;             // float4 per thread per tile (K then V)
    static_assert((KT * (DH / 4)) % THREADS == 0, "a K (or V) tile must divide evenly over the block");
};

template <int NW, int NP>
__global__ __launch_bounds__(64 * NW, 2) void attention_bf16_kernel(const float* __restrict__ QKV, float* __restrict__ CTX, int N,
                                                                  int heads, int qgroups, float scale_log2e,
                                                                  float* __restrict__ LSE) {
    using Cfg = AttnBfCfg<NW, NP>;
    constexpr int DH = Cfg::DH, KT = Cfg::KT, THREADS = Cfg::THREADS, KSTR = Cfg::KSTR, VSTR = Cfg::VSTR, NIMG = Cfg::NIMG;
    constexpr int V4 = Cfg::V4, HALF = V4 / 2, OLD = Cfg::OLD;
    extern __shared__ __attribute__((aligned(16))) __bf16 simg[];
    __bf16* Kimg = simg;                               // [NIMG][32][KSTR]
    __bf16* Vimg = simg + NIMG * Cfg::KIMG;            // [NIMG][32][VSTR]

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = bid & 7;
    const int lin = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (bid >> 3);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld = 3L * D;
    const float* base = QKV + (long)b * N * ld + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;                    // wave-uniform; invalid waves still stage and run the (discarded) math,
                                                       // so EXEC stays all ones around the transposing reads

    // Q^T fragments for all 128 d: lane (q = l31, half lh) holds Q[q][16 step + 8 lh + j]
    bf16x8 qf[NIMG][8];
    {
        const float* qp = base + (long)(min(qb * 32, N - 32) + l31) * ld + 8 * lh;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 v0 = *(const f32x4*)(qp + 16 * s), v1 = *(const f32x4*)(qp + 16 * s + 4);
            if (NP == 3) bf16_split8(v0, v1, qf[0][s], qf[NIMG - 1][s]);
            else qf[0][s] = bf16_round8(v0, v1);
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    f32x4 st[V4];
    auto gload = [&](int kt) __attribute__((always_inline)) {
        const float* kp = base + (long)(kt * KT) * ld + D;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int idx = tid + (i % HALF) * THREADS, row = idx >> 5, c4 = idx & 31;
            st[i] = *(const f32x4*)(kp + (long)row * ld + (i >= HALF ? D : 0) + c4 * 4);
        }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int idx = tid + (i % HALF) * THREADS, row = idx >> 5, c4 = idx & 31;
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (__bf16)st[i][e];
                lo[e] = (__bf16)(st[i][e] - (float)hi[e]);
            }
            __bf16* img0 = i >= HALF ? Vimg + row * VSTR + c4 * 4 : Kimg + row * KSTR + c4 * 4;
            *(bf16x4*)img0 = hi;
            if (NP == 3) *(bf16x4*)(img0 + (i >= HALF ? Cfg::VIMG : Cfg::KIMG)) = lo;
        }
    };

    const int k_off = l31 * KSTR + 8 * lh;                                        // + 16 * step
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int v_off = (4 * lh + tq) * VSTR + 16 * tg + 4 * tp;                     // + (16 s + 8 half) * VSTR + 32 dt
    auto vfrag = [&](const __bf16* image, int s, int dt) __attribute__((always_inline)) {
        const __bf16* p = image + v_off + 16 * s * VSTR + 32 * dt;
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
        const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 8 * VSTR));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = c[e];
        }
        return f;
    };

    const int ntiles = N / KT;
    gload(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                   // every wave is done with the previous tile's images
        lstore();
        __syncthreads();
        if (kt + 1 < ntiles) gload(kt + 1);
        // S^T = K Q^T over 8 steps of 16 d
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bf16x8 kh = *(const bf16x8*)(Kimg + k_off + 16 * t);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][t], s, 0, 0, 0);
            if (NP == 3) {
                const bf16x8 kl = *(const bf16x8*)(Kimg + Cfg::KIMG + k_off + 16 * t);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[NIMG - 1][t], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][t], s, 0, 0, 0);
            }
        }
        // online softmax; the 32 keys of this tile sit in 16 registers x 2 lane halves
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * scale_log2e);
        const float mneg = -m_new * scale_log2e;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = exp2f(fmaf(s[r], scale_log2e, mneg));
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // P^T fragments: registers 8s..8s+7 of the accumulator, as they stand
        bf16x8 pf[NIMG][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = s[8 * ss + e];
                const __bf16 ph = (__bf16)pv;
                pf[0][ss][e] = ph;
                if (NP == 3) pf[NIMG - 1][ss][e] = (__bf16)(pv - (float)ph);
            }
        // O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 vh = vfrag(Vimg, ss, dt);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pf[0][ss], o[dt], 0, 0, 0);
                if (NP == 3) {
                    const bf16x8 vl = vfrag(Vimg + Cfg::VIMG, ss, dt);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pf[NIMG - 1][ss], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, pf[0][ss], o[dt], 0, 0, 0);
                }
            }
    }
    __syncthreads();   // K/V images are dead: reuse the LDS to turn O^T into row-major rows
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (LSE != nullptr && lh == 0) LSE[((long)b * heads + h) * N + qb * 32 + l31] = m_run * (scale_log2e * 0.6931471805599453f) + logf(l_tot);
        float* Os = (float*)simg + wid * 32 * OLD;     // [32 q][132]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c] * inv;
                *(f32x4*)(Os + l31 * OLD + dt * 32 + 8 * g + 4 * lh) = v;
            }
        float* out = CTX + ((long)b * N + qb * 32) * D + h * DH;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 2 + lh;
            const f32x4 v = *(const f32x4*)(Os + row * OLD + l31 * 4);
            *(f32x4*)(out + (long)row * D + l31 * 4) = v;
        }
    }
}

template <int NP>
static hipError_t attention_bf16_launch(const float* QKV, float* CTX, int B, int N, int heads, hipStream_t stream,
                                        float* LSE = nullptr) {
    constexpr int NW = 4;
    using Cfg = AttnBfCfg<NW, NP>;
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    auto kern = attention_bf16_kernel<NW, NP>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int qgroups = (N / 32 + NW - 1) / NW;
    const float scale_log2e = 1.4426950408889634f / sqrtf(128.0f);
    hipLaunchKernelGGL(kern, dim3(B * heads * qgroups), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, QKV, CTX, N, heads,
                       qgroups, scale_log2e, LSE);
    return hipGetLastError();
}
