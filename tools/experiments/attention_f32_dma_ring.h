// EXPERIMENT (not part of the build): fused softmax attention, exact fp32, K/V tiles staged global -> LDS by DMA in a four-slot ring.
// Result on MI355X (B = 256, N = 576, 8 heads x 128): correct (all attention / lifting / training parity tests pass with it), two
// waves per SIMD as in attention_f32_kernel, but 3.03 ms per launch against 2.88: with four workgroups per CU the register-staged
// kernel already overlaps one workgroup's loads with the others' MFMAs, and the ring pays four barriers per tile instead of two.
//
//
// Arithmetic and register layout are those of attention_f32.h (S^T = K_tile Q^T with the key on the accumulator row and the query
// on the lane column, online softmax with one scalar state per lane, O^T += V_tile^T P^T with the S accumulators as the B operand).
// attention_f32_kernel stages a K/V tile through registers between two barriers: inside a workgroup nothing overlaps the load, the
// matrix pipe is kept busy only by the other workgroups of the CU (MFMA busy 0.77).  Staging by DMA needs no registers, but a
// double-buffered 32 KB tile per workgroup would leave one wave per SIMD (tools/experiments/attention_f32_dma.h: slower).  So the
// tile is cut along the head dimension into FOUR 8 KB units -- K[:, 0:64], K[:, 64:128], V[:, 0:64], V[:, 64:128] -- which are
// consumed in that order anyway (score MFMAs over d = 0..63, then 64..127; output columns 0..63, then 64..127).  The ring holds
// four units (unit u lives in slot u & 3 = its phase, so every LDS address is static), three units are in flight while one
// multiplies, and a workgroup needs 33 KB: four workgroups of two waves per CU as before, i.e. two waves per SIMD.
// LDS image of a unit: 32 key rows of 64 floats (256 B = 16 chunks of 16 B), unpadded (a DMA instruction writes lane i at
// base + 16 i: four whole rows).  K: chunk c of key row r sits at chunk position c ^ (r & 15) (applied on the global side), so the
// ds_read_b128 of 16 consecutive keys at one logical chunk covers 16 distinct bank groups.  V is read by column (4-byte reads of
// consecutive lanes) and stored as it comes.
#pragma once
#include "common.h"
#include <math.h>

struct AttnDmaCfg {
    static constexpr int NW = 2, DH = 128, KT = 32, HALF = 64, THREADS = 64 * NW;
    static constexpr int UNIT = KT * HALF * 4;                      // 8 KiB
    static constexpr int OLD = DH + 4;                              // output transpose rows (floats)
    static constexpr int RING = 4 * UNIT, OUT = NW * 32 * OLD * 4;
    static constexpr int LDS_BYTES = RING > OUT ? RING : OUT;       // 33 792 B: four workgroups per CU
};

static __global__ __launch_bounds__(AttnDmaCfg::THREADS, 2) void attention_f32_dma_kernel(const float* __restrict__ QKV, float* __restrict__ CTX,
                                                                                         int N, int heads, int qgroups, float scale_log2e,
                                                                                         float* __restrict__ LSE) {
    using Cfg = AttnDmaCfg;
    constexpr int NW = Cfg::NW, DH = Cfg::DH, KT = Cfg::KT, HALF = Cfg::HALF, OLD = Cfg::OLD, UNIT = Cfg::UNIT;
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
    const int ntiles = N / KT, nunits = 4 * ntiles;

    float qreg[64];
    {
        const float* qp = base + (long)(min(qb * 32, N - 32) + l31) * ld + 4 * lh;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 v = *(const f32x4*)(qp + 8 * t);
            qreg[4 * t + 0] = v[0]; qreg[4 * t + 1] = v[1]; qreg[4 * t + 2] = v[2]; qreg[4 * t + 3] = v[3];
        }
    }

    // DMA duty of a wave per unit: 4 of the 8 four-row groups.  Lane -> (row lane >> 4 of the group, chunk position lane & 15).
    // The DMA is inline asm with hand-counted vmcnt (see gemm_f32_dma.h); units past the end re-read the last unit into a slot
    // nobody reads, so that 4 instructions per unit and wave are in flight at all times.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_att;
    const int drow = lane >> 4, dpos = lane & 15;
    auto dma_unit = [&](int u) __attribute__((always_inline)) {
        const int uc = min(u, nunits - 1);
        const int kt = uc >> 2, ph = uc & 3;
        const float* src = base + (long)(kt * KT) * ld + (ph < 2 ? D : 2 * D) + (ph & 1) * HALF;
        const unsigned dst = lds0 + (u & 3) * UNIT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = wid * 4 + j, key = 4 * g + drow;
            const int chunk = ph < 2 ? (dpos ^ (key & 15)) : dpos;
            const float* gp = src + (long)key * ld + (chunk << 2);
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(__builtin_amdgcn_readfirstlane(dst + g * 1024)) : "memory");
        }
    };
    // unit u has landed once at most the 8 DMA instructions of units u+1, u+2 are outstanding; the barrier publishes it and retires
    // every read of unit u-1, whose slot unit u+3 then overwrites
    auto next_unit = [&](int u) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        dma_unit(u + 3);
    };

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const char* K0 = smem_att;                      // slots 0, 1: K halves; 2, 3: V halves
    const float* V0 = (const float*)(smem_att + 2 * UNIT);
    const int krow = l31 * (HALF * 4), ksw = l31 & 15;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the q loads: from here on only DMAs are counted
    dma_unit(0);
    dma_unit(1);
    dma_unit(2);
    for (int kt = 0; kt < ntiles; ++kt) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            next_unit(4 * kt + hf);
            if (valid) {
                const char* Ks = K0 + hf * UNIT;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const f32x4 a = *(const f32x4*)(Ks + krow + (((2 * t + lh) ^ ksw) << 4));
#pragma unroll
                    for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], qreg[4 * (8 * hf + t) + u], s, 0, 0, 0);
                }
            }
        }
        if (valid) {
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
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            next_unit(4 * kt + 2 + hf);
            if (valid) {
                const float* Vs = V0 + hf * (UNIT / 4);
                // O^T += V^T P^T : step r contracts keys key(r,0), key(r,1); this unit holds output columns 64 hf .. 64 hf + 63
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vf = Vs + ((r & 3) + 8 * (r >> 2) + 4 * lh) * HALF + l31;
#pragma unroll
                    for (int dl = 0; dl < 2; ++dl)
                        o[2 * hf + dl] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dl * 32], s[r], o[2 * hf + dl], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the trailing dummy DMAs must not outlive the LDS they write
    __syncthreads();   // the ring is dead: reuse the LDS to turn O^T into row-major rows
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
