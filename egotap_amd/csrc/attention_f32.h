// Fused softmax attention, fp32 on the matrix cores (modeling_vit.py:226-252):
//   ctx[b, n, h*128:(h+1)*128] = softmax(Q_h K_h^T / sqrt(128)) V_h,   no mask, no dropout.
// Q, K, V are read in place from the fused QKV GEMM output [B*N, 3*D] (no permute copies) and the
// [B, heads, N, N] score tensor of the reference (2.7 GB per layer at B = 256) is never materialised.
//
// One wave = 32 query rows; a workgroup of NW waves shares 32-key K/V tiles through LDS.
//   S^T = K_tile Q^T : 32x32 MFMA tile with the KEY on the accumulator row and the QUERY on the lane
//         column, so the online-softmax state (running max / sum) is one scalar per lane and the
//         accumulator registers are, as they stand, the B operand of the next product:
//   O^T += V_tile^T P^T : four 32(d) x 32(q) MFMA tiles, A operand = V read row-wise from LDS.
// The MFMA k index is permuted like in gemm_f32.h so Q lives in 64 registers loaded as float4 and
// each K fragment is one ds_read_b128 (K rows padded to 132 floats: conflict free).
#pragma once
#include "common.h"
#include <math.h>
#ifdef EGOTAP_ATTN_F32_OLD      // A/B builds (EGOTAP_CXXFLAGS=-DEGOTAP_ATTN_F32_OLD): round 2's kernel, both operands staged in 64 registers
#include "../../tools/experiments/attention_f32_r2.h"
#else

template <int NW>
struct AttnCfg {
    static constexpr int DH = 128, KT = 32, KLD = DH + 4, THREADS = 64 * NW;
    static constexpr int TILE_FLOATS = KT * KLD + KT * DH;
    static constexpr int OUT_FLOATS = NW * 32 * KLD;
    static constexpr int LDS_BYTES = 4 * (TILE_FLOATS > OUT_FLOATS ? TILE_FLOATS : OUT_FLOATS);
};

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attention_f32_kernel(const float* __restrict__ QKV,
                                                                 float* __restrict__ CTX, int N, int heads,
                                                                 int qgroups, float scale_log2e, float* __restrict__ LSE, int ksplit) {
    // [r4] ksplit > 1 (serving batches: B x heads x query groups is a fraction of the chip): workgroup (pair, query group, split) attends to
    // key tiles [split * ntiles / ksplit, ...) only and writes its own normalised output and log-sum-exp -- CTX / LSE are then the PARTIAL
    // buffers [ksplit][B * N][D] / [ksplit][B * heads * N], merged by attention_f32_merge_kernel.
    using Cfg = AttnCfg<NW>;
    constexpr int DH = Cfg::DH, KT = Cfg::KT, KLD = Cfg::KLD, THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                 // [32][132]
    float* Vs = smem + KT * KLD;      // [32][128]

    // blocks that share an L2 (same blockIdx % 8) get a contiguous run of (batch, head) pairs
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = bid & 7;
    const int lin = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (bid >> 3);
    const int split = ksplit > 1 ? lin % ksplit : 0;
    const int lq = ksplit > 1 ? lin / ksplit : lin;
    const int bh = lq / qgroups, qg = lq - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld = 3L * D;
    const float* base = QKV + (long)b * N * ld + h * DH;     // q of token 0 of this (b, h)
    const int qb = qg * NW + wid;                              // 32-row query block of this wave
    const bool valid = qb * 32 < N;                            // wave-uniform
    // [r5] N need not be a multiple of 32 (the reference allows any heatmap side that is a multiple of 16, net_architecture.py:327: N = 144 at
    // 32 x 32 heatmaps, 1296 at 96 x 96): the LAST query block and the LAST key tile start at N - 32 and overlap their predecessors; the
    // overlapping keys -- already counted by the tile before -- are masked to -inf, the overlapping queries are simply computed twice (same bits)
    const int q0 = min(qb * 32, N - 32);

    float qreg[64];
    {
        const float* qp = base + (long)(q0 + l31) * ld + 4 * lh;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 v = *(const f32x4*)(qp + 8 * t);
            qreg[4 * t + 0] = v[0]; qreg[4 * t + 1] = v[1]; qreg[4 * t + 2] = v[2]; qreg[4 * t + 3] = v[3];
        }
    }
    float m_run = -INFINITY, l_run = 0.f;

    // K / V tiles go global -> registers -> LDS in two steps (cdna_hip_programming.md T14), and [r3] the two operands take turns in
    // ONE set of staging registers: K(t + 1) is requested under tile t's P V MFMAs of the PREVIOUS iteration and written right behind the
    // barrier that ends tile t's score MFMAs (the last readers of K(t)); V(t + 1) is then requested into the same registers, flies under
    // the softmax and the P V MFMAs, and is written behind the barrier that ends them.  Still two barriers per tile, half the staging
    // registers (32 instead of 64): the kernel no longer spills (7 VGPRs before, whose scratch reloads waited -- vmcnt is one in-order
    // counter -- for the K / V loads in flight).
    constexpr int PER = KT * (DH / 4) / THREADS;
    f32x4 stg[PER];
    auto tile_req = [&](int kt, int vpart) __attribute__((always_inline)) {      // vpart: 0 = K rows, 1 = V rows of key tile kt
        const float* kp = base + (long)min(kt * KT, N - KT) * ld + D + vpart * D;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            stg[i] = *(const f32x4*)(kp + (size_t)(unsigned)(row * (int)ld + c4 * 4));
        }
    };
    auto tile_put = [&](int vpart) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            if (vpart) *(f32x4*)(Vs + row * DH + c4 * 4) = stg[i];
            else *(f32x4*)(Ks + row * KLD + c4 * 4) = stg[i];
        }
    };
    const int tiles_all = (N + KT - 1) / KT;
    const int overlap = tiles_all * KT - N;                     // keys of the last tile that the tile before it already covered (0: N is a multiple of 32)
    const int kt0 = ksplit > 1 ? (int)((long)split * tiles_all / ksplit) : 0;
    const int ntiles = ksplit > 1 ? (int)((long)(split + 1) * tiles_all / ksplit) : tiles_all;      // one past this workgroup's last key tile
    // prologue: K(kt0) and V(kt0) into LDS, K(kt0 + 1) into the staging registers.  [r4] K(kt0) and V(kt0) are requested TOGETHER (the output
    // accumulators are not live yet: a second staging set costs nothing) -- they were two dependent round trips in front of every block's first MFMA
    {
        f32x4 stg2[PER];
        const float* vp = base + (long)min(kt0 * KT, N - KT) * ld + 2 * D;
        tile_req(kt0, 0);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            stg2[i] = *(const f32x4*)(vp + (size_t)(unsigned)(row * (int)ld + c4 * 4));
        }
        tile_put(0);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            *(f32x4*)(Vs + row * DH + c4 * 4) = stg2[i];
        }
    }
    if (kt0 + 1 < ntiles) tile_req(kt0 + 1, 0);
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    __syncthreads();
    // The running maximum is raised (and the output rescaled) only when a query's scores exceed it by more than 2^RESC in the
    // softmax's base-2 units (T13): probabilities stay below 2^RESC -- harmless in fp32 -- and the 64 multiplies per tile vanish
    // from almost every tile.
    constexpr float RESC = 8.0f;
    for (int kt = kt0; kt < ntiles; ++kt) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if (valid) {
            const float* kf = Ks + l31 * KLD + 4 * lh;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 a = *(const f32x4*)(kf + 8 * t);
#pragma unroll
                for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], qreg[4 * t + u], s, 0, 0, 0);
            }
        }
        __syncthreads();   // A: every wave has read K(kt)
        if (kt + 1 < ntiles) {
            tile_put(0);                      // K(kt + 1)
            tile_req(kt + 1, 1);              // V(kt + 1) flies under the softmax and the P V MFMAs
        }
        if (valid) {
            if (overlap > 0 && kt == tiles_all - 1) {          // (wave-uniform) register r of lane half lh holds key (r & 3) + 8 (r >> 2) + 4 lh of the tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) + 4 * lh < overlap) s[r] = -INFINITY;
            }
            // online softmax; the 32 keys of this tile sit in 16 registers x 2 lane halves
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
            const bool raise = mx > m_run + RESC;
            if (__builtin_amdgcn_ballot_w64(raise) != 0) {
                const float m_new = raise ? mx : m_run;
                const float alpha = exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));      // (v_exp_f32: arguments <= 2^RESC; what exp2f adds is the denormal range, 1e-38 of a probability)
                psum += s[r];
            }
            l_run += psum;                   // per lane-half partial sum; halves are added at the end
            // O^T += V^T P^T : step r contracts keys key(r,0), key(r,1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vf = Vs + ((r & 3) + 8 * (r >> 2) + 4 * lh) * DH + l31;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dt * 32], s[r], o[dt], 0, 0, 0);
            }
        }
        __syncthreads();   // B: every wave has read V(kt)
        if (kt + 1 < ntiles) {
            tile_put(1);                      // V(kt + 1)
            if (kt + 2 < ntiles) tile_req(kt + 2, 0);      // K(kt + 2) flies under the next tile's score MFMAs
        }
    }
    __syncthreads();   // K/V tiles are dead: reuse the LDS to turn O^T into row-major rows
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        // training: log-sum-exp of the scaled scores (natural log) per query row, for the flash-style backward
        const long part = ksplit > 1 ? (long)split * (gridDim.x / (ksplit * qgroups)) : 0;      // split * (B * heads): offset of this split's partial buffers, in pairs
        if (LSE != nullptr && lh == 0) LSE[(part + (long)b * heads + h) * N + q0 + l31] = m_run * 0.6931471805599453f + logf(l_tot);      // m_run is in base-2 units
        float* Os = smem + wid * 32 * KLD;     // [32 q][132]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c] * inv;
                *(f32x4*)(Os + l31 * KLD + dt * 32 + 8 * g + 4 * lh) = v;
            }
        // same wave reads back what it wrote: no barrier needed, only LDS completion (compiler waits)
        float* out = CTX + (part / heads * N + (long)b * N + q0) * D + h * DH;       // (part / heads = split * B images)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 2 + lh;
            const f32x4 v = *(const f32x4*)(Os + row * KLD + l31 * 4);
            *(f32x4*)(out + (long)row * D + l31 * 4) = v;
        }
    }
}

// ctx[b, n, h, :] = sum_s w_s part_s[b, n, h, :],  w_s = exp(lse_s - lse) / sum_s' exp(lse_s' - lse): the exact merge of softmax attention over
// disjoint key ranges (each part is normalised by its own sum).  One thread = 4 channels of one (token, head).
static __global__ __launch_bounds__(256) void attention_f32_merge_kernel(const float* __restrict__ part, const float* __restrict__ lse_part, float* __restrict__ ctx,
                                                                         int B, int N, int heads, int ksplit) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;         // over B * N * heads * 32 float4
    const long total = (long)B * N * heads * 32;
    if (i >= total) return;
    const int c4 = (int)(i & 31);
    const long t = i >> 5;                                              // (b * N + n) * heads + h
    const int h = (int)(t % heads);
    const long bn = t / heads;
    const int b = (int)(bn / N), n = (int)(bn - (long)b * N);
    float l[8], m = -INFINITY;
    for (int sp = 0; sp < ksplit; ++sp) {
        l[sp] = lse_part[(((long)sp * B + b) * heads + h) * N + n];
        m = fmaxf(m, l[sp]);
    }
    float wsum = 0.f;
    for (int sp = 0; sp < ksplit; ++sp) {
        l[sp] = expf(l[sp] - m);
        wsum += l[sp];
    }
    const float inv = 1.0f / wsum;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    const long D = (long)heads * 128, off = bn * D + h * 128 + c4 * 4;
    for (int sp = 0; sp < ksplit; ++sp) {
        const f32x4 v = *(const f32x4*)(part + (long)sp * B * N * D + off);
        const float w = l[sp] * inv;
        acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2]; acc[3] += w * v[3];
    }
    *(f32x4*)(ctx + off) = acc;
}

// scratch / scratch_floats (inference at serving batches only; LSE must be null): room for the key-split partials.  The split count
// fills ~4.5 two-wave workgroups per CU: B = 1 / 2 (72 / 144 workgroups) -> 6 ranges of 3 key tiles, B = 4 -> 3, B = 8 -> 2, B >= 16 -> none.
static hipError_t attention_f32_launch(const float* QKV, float* CTX, int B, int N, int heads, hipStream_t stream,
                                       float* LSE = nullptr, float* scratch = nullptr, size_t scratch_floats = 0, int num_cu = 256) {
    constexpr int NW = 2;
    using Cfg = AttnCfg<NW>;
    if (B <= 0) return hipSuccess;
    if (N < 32 || N % 4 != 0) return hipErrorInvalidValue;
    auto kern = attention_f32_kernel<NW>;
    const int qgroups = ((N + 31) / 32 + NW - 1) / NW;
    const float scale_log2e = 1.4426950408889634f / sqrtf(128.0f);
    const long wgs = (long)B * heads * qgroups;
    int ksplit = 1;
    if (scratch != nullptr && LSE == nullptr) {
        const int ntiles = (N + 31) / 32;
        // as many ranges as keep the chip at ~2 waves per SIMD (a workgroup is two waves: 4.5 workgroups per CU), each a whole number of key tiles
        for (int k = 8; k >= 2; --k)
            if (ntiles % k == 0 && wgs * k <= 9L * num_cu / 2 && (size_t)k * ((size_t)B * N * heads * 128 + (size_t)B * heads * N) <= scratch_floats) { ksplit = k; break; }
    }
    if (ksplit > 1) {
        float* part = scratch;
        float* lse_part = scratch + (size_t)ksplit * B * N * heads * 128;
        hipLaunchKernelGGL(kern, dim3((unsigned)(wgs * ksplit)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, QKV, part, N, heads, qgroups, scale_log2e, lse_part, ksplit);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const long total = (long)B * N * heads * 32;
        hipLaunchKernelGGL(attention_f32_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float*)part, (const float*)lse_part, CTX, B, N, heads, ksplit);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(B * heads * qgroups), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, QKV, CTX, N, heads,
                       qgroups, scale_log2e, LSE, 1);
    return hipGetLastError();
}
#endif
