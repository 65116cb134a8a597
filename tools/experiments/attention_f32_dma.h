// EXPERIMENT (not part of the build): fused softmax attention, exact fp32, K/V tiles staged global -> LDS by DMA, double buffered.
// Result on MI355X (B = 256, N = 576, 8 heads x 128): correct (all attention / lifting parity tests pass with it), but 3.4 ms per
// launch against 2.88 ms for attention_f32_kernel: 64 KB of stages per workgroup leave one wave per SIMD, and with a single wave
// the softmax VALU (~900 clk per tile), the 16 DMA issues per tile (~56 matrix-pipe cycles each) and the barrier are all exposed,
// which costs more than the asynchronous staging saves.  Kept for the record; a version worth building needs 2 waves per SIMD
// (K and V tiles shared by four query blocks, which the 18 query blocks of N = 576 do not divide into) or the softmax of tile
// kt interleaved with the score MFMAs of tile kt+1.
//
//
// Same arithmetic layout as attention_f32.h (S^T = K_tile Q^T with the key on the accumulator row and the query on the lane
// column, online softmax with one scalar state per lane, O^T += V_tile^T P^T with the S accumulators as the B operand).  What
// changes is the staging: attention_f32_kernel loads a K/V tile through registers between two barriers, so within a workgroup
// nothing overlaps the load and the matrix pipe is kept busy only by the other three workgroups of the CU (measured MFMA busy
// 0.77).  Here tile kt+1 is in flight (global_load_lds_dwordx4, issued right after the barrier that publishes tile kt) while tile
// kt multiplies: two stages of (K 16 KB + V 16 KB) per workgroup = 64 KB -> two workgroups of two waves per CU, ONE wave per SIMD.
// With no second wave to hide behind, the 64 score MFMAs of a tile go to four accumulators (t mod 4) instead of one dependent
// chain, and every LDS fragment is read one MFMA group ahead.
// LDS image: a K row is 128 floats = 32 chunks of 16 bytes, unpadded (the DMA writes lane i at base + 16 i: two whole rows per
// wave instruction); chunk c of key row r sits at chunk position c ^ (r & 15), applied on the global side, so the ds_read_b128 of
// 16 consecutive keys at one logical chunk covers 16 distinct bank groups.  V rows are read by column (4-byte reads of
// consecutive lanes) and are stored as they come.
#pragma once
#include "common.h"
#include <math.h>

struct AttnDmaCfg {
    static constexpr int NW = 2, DH = 128, KT = 32, THREADS = 64 * NW;
    static constexpr int TILE_BYTES = KT * DH * 4;                 // 16 KiB per operand
    static constexpr int STAGE = 2 * TILE_BYTES;                   // K | V
    static constexpr int OLD = DH + 4;                             // output transpose rows (floats)
    static constexpr int LDS_BYTES = 2 * STAGE;                    // 64 KiB (>= NW * 32 * OLD * 4 for the output transpose)
    static_assert(NW * 32 * OLD * 4 <= LDS_BYTES, "output transpose fits in the tile stages");
};

static __global__ __launch_bounds__(AttnDmaCfg::THREADS, 1) void attention_f32_dma_kernel(const float* __restrict__ QKV, float* __restrict__ CTX, int N,
                                                                                  int heads, int qgroups, float scale_log2e,
                                                                                  float* __restrict__ LSE) {
    using Cfg = AttnDmaCfg;
    constexpr int NW = Cfg::NW, DH = Cfg::DH, KT = Cfg::KT, OLD = Cfg::OLD;
    extern __shared__ __attribute__((aligned(16))) char smem_att[];

    // blocks that share an L2 (same blockIdx % 8) get a contiguous run of (batch, head) pairs
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = bid & 7;
    const int lin = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (bid >> 3);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld = 3L * D;
    const float* base = QKV + (long)b * N * ld + h * DH;     // q of token 0 of this (b, h)
    const int qb = qg * NW + wid;                              // 32-row query block of this wave
    const bool valid = qb * 32 < N;                            // wave-uniform

    // DMA duty of a wave per tile: 8 of the 16 two-row groups of K and of V.  Lane -> (row lane >> 5 of the pair, chunk position
    // lane & 31); K fetches logical chunk (lane & 31) ^ (key & 15), V fetches chunk lane & 31.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_att;
    const int ntiles = N / KT;
    auto dma1 = [&](const float* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    auto dma_tile = [&](int kt, int st) __attribute__((always_inline)) {
        const int ktc = min(kt, ntiles - 1);                       // past the end: re-read the last tile into a dead stage (constant vmcnt)
        const float* kp = base + (long)(ktc * KT) * ld + D;
        const unsigned sa = lds0 + st * Cfg::STAGE;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int key = 2 * (wid * 8 + g) + (lane >> 5);
            dma1(kp + (long)key * ld + (((lane & 31) ^ (key & 15)) << 2), sa + (wid * 8 + g) * 1024);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int key = 2 * (wid * 8 + g) + (lane >> 5);
            dma1(kp + (long)key * ld + D + ((lane & 31) << 2), sa + Cfg::TILE_BYTES + (wid * 8 + g) * 1024);
        }
    };

    float qreg[64];
    {
        const float* qp = base + (long)(min(qb * 32, N - 32) + l31) * ld + 4 * lh;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 v = *(const f32x4*)(qp + 8 * t);
            qreg[4 * t + 0] = v[0]; qreg[4 * t + 1] = v[1]; qreg[4 * t + 2] = v[2]; qreg[4 * t + 3] = v[3];
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // K fragment of k-step t: key row l31, logical chunk 2 t + lh -> position (2 t + lh) ^ (l31 & 15)
    const int krow = l31 * (DH * 4), ksw = l31 & 15;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the q loads: from here on only DMAs are counted
    dma_tile(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int st = kt & 1;
        // tile kt has landed when nothing older than ... is outstanding: the only DMAs in flight are tile kt's 16; the barrier
        // publishes it and retires every read of the other stage (tile kt-1), which the next DMA overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma_tile(kt + 1, st ^ 1);
        if (valid) {
            const char* Ks = smem_att + st * Cfg::STAGE;
            const float* Vs = (const float*)(Ks + Cfg::TILE_BYTES);
            f32x16 s4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) s4[c][r] = 0.f;
            // one wave per SIMD: nothing else hides the LDS latency, so fragment t+2 is requested before the MFMAs of fragment t
            // (the order is pinned; left alone the scheduler put every read right in front of its use)
            f32x4 ka[3];
            ka[0] = *(const f32x4*)(Ks + krow + (((0 + lh) ^ ksw) << 4));
            ka[1] = *(const f32x4*)(Ks + krow + (((2 + lh) ^ ksw) << 4));
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t + 2 < 16) ka[(t + 2) % 3] = *(const f32x4*)(Ks + krow + (((2 * (t + 2) + lh) ^ ksw) << 4));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t % 3][u], qreg[4 * t + u], s4[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (s4[0][r] + s4[1][r]) + (s4[2][r] + s4[3][r]);
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
            l_run = l_run * alpha + psum;     // per lane-half partial sum; halves are added at the end
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            // O^T += V^T P^T : step r contracts keys key(r,0), key(r,1)
            float vv[3][4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float* vf = Vs + ((p & 3) + 8 * (p >> 2) + 4 * lh) * DH + l31;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vv[p][dt] = vf[dt * 32];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 2 < 16) {
                    const float* vf = Vs + (((r + 2) & 3) + 8 * ((r + 2) >> 2) + 4 * lh) * DH + l31;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) vv[(r + 2) % 3][dt] = vf[dt * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r % 3][dt], s[r], o[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the trailing dummy DMA must not outlive the LDS it writes
    __syncthreads();   // K/V tiles are dead: reuse the LDS to turn O^T into row-major rows
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        // training: log-sum-exp of the scaled scores (natural log) per query row, for the flash-style backward
        if (LSE != nullptr && lh == 0) LSE[((long)b * heads + h) * N + qb * 32 + l31] = m_run * (scale_log2e * 0.6931471805599453f) + logf(l_tot);
        float* Os = (float*)smem_att + wid * 32 * OLD;     // [32 q][132]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c] * inv;
                *(f32x4*)(Os + l31 * OLD + dt * 32 + 8 * g + 4 * lh) = v;
            }
        // same wave reads back what it wrote: no barrier needed, only LDS completion (compiler waits)
        float* out = CTX + ((long)b * N + qb * 32) * D + h * DH;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 2 + lh;
            const f32x4 v = *(const f32x4*)(Os + row * OLD + l31 * 4);
            *(f32x4*)(out + (long)row * D + l31 * 4) = v;
        }
    }
}

static hipError_t attention_f32_dma_launch(const float* QKV, float* CTX, int B, int N, int heads, hipStream_t stream, float* LSE = nullptr) {
    using Cfg = AttnDmaCfg;
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0 || ((uintptr_t)QKV & 15) != 0) return hipErrorInvalidValue;
    auto kern = attention_f32_dma_kernel;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int qgroups = (N / 32 + Cfg::NW - 1) / Cfg::NW;
    const float scale_log2e = 1.4426950408889634f / sqrtf(128.0f);
    hipLaunchKernelGGL(kern, dim3(B * heads * qgroups), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, QKV, CTX, N, heads, qgroups, scale_log2e, LSE);
    return hipGetLastError();
}
