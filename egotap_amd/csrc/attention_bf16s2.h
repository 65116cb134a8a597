// [r3] Attention backward on bf16 tensors, second generation (bf16-storage training mode; model/modeling_vit.py:226-252 backward).
//
// Round 2's dK + dV kernel (attention_bf16s.h) staged each 32-query tile of Q and dO through registers into FOUR padded LDS images
// (a row image and a transposed-read image each) with NO prefetch -- 160 accumulator / fragment registers left no room for staging
// registers -- so every tile paid a global-load round trip between two barriers: MFMA busy 0.295, the furthest kernel from its
// roofline in the whole step.  Here:
//   * ONE image per tile serves the row reads (ds_read_b128: S = Q K^T, dP = dO V^T) AND the transposed reads (ds_read_b64_tr_b16:
//     dV^T += dO^T P, dK^T += Q^T dS): cdna_hip_programming.md T10 image (a), 8-row x 32-column subtiles of 512 bytes,
//       off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)),
//     conflict-free for both kinds of read with only two base registers each (constant offsets fold into the DS immediates);
//   * the image is filled by the LDS DMA (global_load_lds_dwordx4: 16 wave-instructions per 32-query step and workgroup, no staging
//     registers, no ds_write); the swizzle is applied on the global side -- which 16 bytes a lane fetches -- the DMA writes lane i
//     at base + 16 i (one instruction = 8 rows x two 64-byte column blocks);
//   * double buffered: the DMA of tile t + 1 is issued right after the barrier that opens tile t and lands under tile t's 32 MFMAs;
//     ONE barrier per tile (own DMA retired by s_waitcnt vmcnt(0), then s_barrier: every wave's pieces have landed and every wave
//     is done reading the buffer about to be refilled);
//   * log-sum-exp and delta of the tile's 32 queries ride along as one global_load_lds_dword.
// Arithmetic per (32 queries x 32 keys) pair and wave is round 2's: S, dV, dP, dK = 4 products, P and dS rounded to bf16 where they
// become MFMA operands, fixed summation order, no atomics -> bitwise reproducible.
#pragma once
#include "attention_bf16s.h"
#include "lds_dma.h"

namespace att2 {
using namespace attnbf;
constexpr int TILEB = 32 * 256;                    // one 32 x 128 bf16 image
constexpr int LSB = 256;                           // 32 log-sum-exp + 32 delta values (fp32)
constexpr int BUF = 2 * TILEB + LSB;               // Q image | dO image | lse, delta

__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
// [r5] the same DMA with the address as a wave-uniform 64-bit base (scalar registers) + a 32-bit byte offset per lane: every tile of these kernels is
// "uniform tile origin + a lane pattern fixed for the whole kernel", so the per-lane 64-bit add (and the 64-bit address operand) of the flat form is dropped.
// -DEGOTAP_ATT_DMA_FLAT keeps the per-lane pointer form (A/B: profiles/r05_attention_ab.log).
__device__ __forceinline__ unsigned long long uniform64(const void* p) { return lds_dma_base(p); }      // lds_dma.h
__device__ __forceinline__ void dma16s(unsigned voff, unsigned long long sbase, unsigned lds_addr) {
#ifdef EGOTAP_ATT_DMA_FLAT
    dma16((const char*)(size_t)sbase + voff, lds_addr);
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
#endif
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

// element offset (row * ld + column) this lane fetches for DMA piece e (0..7) of a 32 x 128 tile whose rows are ld elements apart:
// piece e covers image bytes [1024 e, 1024 e + 1024) = rows 8 (e >> 1) .. + 7, column blocks 2 (e & 1) and 2 (e & 1) + 1
__device__ __forceinline__ int dma_src(int e, int lane, int ld) {
    const int row = 8 * (e >> 1) + ((lane >> 2) & 7);
    const int ch = 4 * (2 * (e & 1) + (lane >> 5)) + ((lane & 3) ^ ((row >> 2) & 3));
    return row * ld + 8 * ch;
}

struct LaneAddr {
    unsigned row0, row1;     // row read (32x32x16 A / B operand: row lane & 31, chunk 2 t + lane >> 5): t even / odd; + 512 (t >> 1)
    unsigned tr0, tr1;       // transposed read: rows 16 ss + 4 lh + q (tr0) and + 8 (tr1); + 4096 ss + 512 dt
};
__device__ __forceinline__ LaneAddr lane_addr(int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    const int s2 = (l31 >> 2) & 3;
    const unsigned rb = 2048 * (l31 >> 3) + 64 * (l31 & 7);
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const unsigned tb = 64 * (4 * lh + tq) + 8 * (tp & 1);
    LaneAddr a;
    a.row0 = rb + 16 * (lh ^ s2);
    a.row1 = rb + 16 * ((lh ^ s2) ^ 2);
    a.tr0 = tb + 16 * ((2 * tg + (tp >> 1)) ^ lh);
    a.tr1 = tb + 2048 + 16 * ((2 * tg + (tp >> 1)) ^ (lh + 2));
    return a;
}

// T[32][32] = A_img[32 rows][128] x B^T, B given as 8 k-step fragments in registers: acc col = B's row (lane), rows = A's rows
__device__ __forceinline__ f32x16 rows_x_frags(const char* img, const LaneAddr& la, const Frags<1>& fr) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bf16x8 a = *(const bf16x8*)(img + ((t & 1) ? la.row1 : la.row0) + 512 * (t >> 1));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fr.f[0][t], s, 0, 0, 0);
    }
    return s;
}
// the same with B read from a second image (same lane pattern): T = A_img x B_img^T
__device__ __forceinline__ f32x16 rows_x_rows(const char* aimg, const char* bimg, const LaneAddr& la) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const unsigned o = ((t & 1) ? la.row1 : la.row0) + 512 * (t >> 1);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(aimg + o), *(const bf16x8*)(bimg + o), s, 0, 0, 0);
    }
    return s;
}
// o^T[4 d-tiles][d][lane col] += sum_rows img[row][d] * p[row][lane col]: p's accumulator registers, rounded to bf16 where they
// stand, are the B operand (k order 16 ss + 8 (j >> 2) + 4 lh + (j & 3)); the image comes back transposed in exactly that order
__device__ __forceinline__ void imgT_x_p(f32x16 (&o)[4], const char* img, const LaneAddr& la, const f32x16& p) {
    bf16x8 ph[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int e = 0; e < 8; ++e) ph[ss][e] = (__bf16)p[8 * ss + e];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr0 + 4096 * ss + 512 * dt));
            const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr1 + 4096 * ss + 512 * dt));
            bf16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a[e];
                f[4 + e] = c[e];
            }
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, ph[ss], o[dt], 0, 0, 0);
        }
}
}  // namespace att2

// ------------------------------------------------------------------------------------------------- dK and dV
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_bf16s2_kernel(const __bf16* __restrict__ QKV, const __bf16* __restrict__ dO,
                                                                        const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                        __bf16* __restrict__ dQKV, int N, int heads, int kgroups, float scale,
                                                                        float* __restrict__ colpart) {
    // colpart (or null): fp32 [B * N / 32][3 * heads * 128]; row (b, key block) receives the column sums of the block's 32 dK rows at
    // columns D + h * 128 .. and of its dV rows at 2 D + h * 128 ..  (the dQ kernel fills columns h * 128 .. of row (b, query block))
    using namespace att2;
    static_assert(NW == 4, "a workgroup's four waves stage the four 8-row groups of a tile");
    extern __shared__ __attribute__((aligned(16))) char sm2[];
    const int lin = xcd_lin(blockIdx.x, gridDim.x);          // the key blocks of one (batch, head) share Q / dO tiles in one L2
    const int bh = lin / kgroups, kg = lin - bh * kgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const int ld3 = 3 * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const __bf16* dob = dO + (long)b * N * D + h * DH;
    const float* lsep = LSE + (long)bh * N;
    const float* delp = DELTA + (long)bh * N;
    const int kb = kg * NW + wid;
    const bool valid = kb * 32 < N;            // invalid waves (last group of a head) redo the last key block and skip the store
    const int k0 = min(kb * 32, N - 32);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm2;
    const unsigned vimg = 2 * BUF + wid * TILEB;               // this wave's V rows (dP = dO V^T reads them as the B operand)

    Frags<1> kf;                                               // K rows of this wave's 32 keys, all 128 d: B operand of S = Q K^T
    attns::load_row_frags(kf, qkv + (long)(k0 + l31) * ld3 + D, lh);
#pragma unroll
    for (int e = 0; e < 8; ++e) dma16(qkv + (long)k0 * ld3 + 2 * D + dma_src(e, lane, ld3), lds0 + vimg + 1024 * e);

    // DMA duty per tile: 8-row group `wid` of the Q tile and of the dO tile (two pieces each); wave 0 also fetches lse | delta
    const unsigned oq0 = 2u * dma_src(2 * wid, lane, ld3), oq1 = 2u * dma_src(2 * wid + 1, lane, ld3);      // byte offsets from the tile's first row
    const unsigned od0 = 2u * dma_src(2 * wid, lane, D), od1 = 2u * dma_src(2 * wid + 1, lane, D);
    auto issue = [&](int qt, unsigned boff) __attribute__((always_inline)) {
        const unsigned long long qs = uniform64(qkv + (long)(qt * 32) * ld3);
        const unsigned long long ds = uniform64(dob + (long)(qt * 32) * D);
        const unsigned a = lds0 + boff + 2048 * wid;
        dma16s(oq0, qs, a);
        dma16s(oq1, qs, a + 1024);
        dma16s(od0, ds, a + TILEB);
        dma16s(od1, ds, a + TILEB + 1024);
        if (wid == 0) dma4((lane < 32 ? lsep : delp - 32) + qt * 32 + lane, lds0 + boff + 2 * TILEB);
    };
    const LaneAddr la = lane_addr(lane);
    const float c2 = scale * 1.4426950408889634f;
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;

    const int ntiles = N / 32;
    issue(0, 0);
    attns::settle_frags(kf);                                    // (the waits for these loads belong here, not inside the tile loop: attention_bf16s.h)
    auto step = [&](int qt, unsigned boff) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of tile qt (issued a whole tile ago) have landed ...
        __builtin_amdgcn_s_barrier();                           // ... and so have everyone's; everyone is done with the other buffer
        __builtin_amdgcn_sched_barrier(0);
        if (qt + 1 < ntiles) issue(qt + 1, boff ^ BUF);         // lands under this tile's MFMAs
        if (!valid) return;                                     // (wave-uniform) a wave past the last key block only stages: its SIMD's matrix pipe goes to the CU's other workgroup
        const char* Qi = sm2 + boff;
        const char* Di = sm2 + boff + TILEB;
        const float* Ls = (const float*)(sm2 + boff + 2 * TILEB);
        f32x16 p = rows_x_frags(Qi, la, kf);                   // S[q][key]
        // accumulator register r holds query (r & 3) + 8 (r >> 2) + 4 lh: registers 4 g .. 4 g + 3 are four consecutive queries
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *(const f32x4*)(Ls + 8 * g + 4 * lh);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[4 * g + c] = __builtin_amdgcn_exp2f(fmaf(p[4 * g + c], c2, -1.4426950408889634f * l4[c]));
        }
        imgT_x_p(dv, Di, la, p);                                // dV^T[d][key] += dO^T P
        const f32x16 dp = rows_x_rows(Di, sm2 + vimg, la);      // dP[q][key] = dO V^T
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 d4 = *(const f32x4*)(Ls + 32 + 8 * g + 4 * lh);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[4 * g + c] = p[4 * g + c] * (dp[4 * g + c] - d4[c]) * scale;      // dS[q][key]
        }
        imgT_x_p(dk, Qi, la, p);                                // dK^T[d][key] += Q^T dS
    };
    for (int qt = 0; qt < ntiles; qt += 2) {
        step(qt, 0);
        if (qt + 1 < ntiles) step(qt + 1, BUF);
    }
    __syncthreads();                                           // the images are dead: their space becomes the output patches
    if (valid) {
        float* patch = (float*)sm2 + wid * 32 * OLD;
        __bf16* dst = dQKV + ((long)b * N + k0) * ld3 + h * DH;
        float* cp = colpart ? colpart + ((long)b * (N / 32) + kb) * ld3 + h * DH : nullptr;
        attns::store_rows_bf16(dk, 1.0f, patch, dst + D, ld3, lane, cp ? cp + D : nullptr);
        attns::store_rows_bf16(dv, 1.0f, patch, dst + 2 * D, ld3, lane, cp ? cp + 2 * D : nullptr);    // same wave, same patch: program order
    }
}

static hipError_t attention_bf16s2_dkv_launch(const __bf16* QKV, const __bf16* dO, const float* LSE, const float* DELTA, __bf16* dQKV, int B, int N,
                                              int heads, hipStream_t stream, float* colpart) {
    using namespace att2;
    constexpr int NW = 4;
    constexpr size_t img = 2 * BUF + (size_t)NW * TILEB, patch = (size_t)NW * 32 * OLD * 4;
    constexpr size_t lds = img > patch ? img : patch;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    auto kern = attn_bwd_dkv_bf16s2_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int groups = (N / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3(B * heads * groups), dim3(64 * NW), lds, stream, QKV, dO, LSE, DELTA, dQKV, N, heads, groups, 1.0f / sqrtf((float)DH),
                       colpart);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------- dQ (+ delta), generation 2
// Per 32-query block and wave: S^T = K Q^T, dP^T = V dO^T (A operand = K / V rows from the image, B = Q / dO fragments in
// registers), dQ^T += K^T dS^T (K transposed from the SAME image).  64 keys per step: K and V tiles as 2 + 2 images, DMA-staged and
// double buffered as above (32 pieces per step and workgroup, 8 per wave).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_bf16s2_kernel(const __bf16* __restrict__ QKV, const __bf16* __restrict__ O,
                                                                       const __bf16* __restrict__ dO, const float* __restrict__ LSE,
                                                                       __bf16* __restrict__ dQKV, float* __restrict__ DELTA, int N, int heads,
                                                                       int qgroups, float scale, float* __restrict__ colpart) {
    using namespace att2;
    static_assert(NW == 4, "four waves share the DMA duty of a 64-key step");
    constexpr unsigned KVBUF = 4 * TILEB;                      // K keys 0-31 | K keys 32-63 | V keys 0-31 | V keys 32-63
    extern __shared__ __attribute__((aligned(16))) char sm2[];
    const int lin = xcd_lin(blockIdx.x, gridDim.x);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const int ld3 = 3 * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const long orow = ((long)b * N + q0 + l31) * D + h * DH;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm2;
    unsigned ok[4];                                            // this wave's 4 K pieces of a step, in bytes from the step's first row (the V pieces sit D elements further)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = 4 * wid + i;
        ok[i] = 2u * (unsigned)(32 * (e >> 3) * ld3 + D + dma_src(e & 7, lane, ld3));
    }
    auto issue = [&](int kt, unsigned boff) __attribute__((always_inline)) {
        const unsigned long long src = uniform64(qkv + (long)(kt * 64) * ld3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a = lds0 + boff + 1024 * (4 * wid + i);         // piece e of the step: image e >> 3, piece e & 7 = byte 1024 e
            dma16s(ok[i], src, a);
            dma16s(ok[i], src + 2ull * D, a + 2 * TILEB);
        }
    };
    issue(0, 0);
    Frags<1> qf, dof;
    attns::load_row_frags(qf, qkv + (long)(q0 + l31) * ld3, lh);
    attns::load_row_frags(dof, dO + orow, lh);
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bf16x8 o8 = *(const bf16x8*)(O + orow + 16 * s + 8 * lh);
#pragma unroll
        for (int u = 0; u < 8; ++u) delta += (float)o8[u] * (float)dof.f[0][s][u];
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse = LSE[(long)bh * N + q0 + l31];
    if (valid && lh == 0) DELTA[(long)bh * N + q0 + l31] = delta;
    const float c2 = scale * 1.4426950408889634f, lse2 = lse * 1.4426950408889634f;
    attns::settle_frags(qf);
    attns::settle_frags(dof);
    asm volatile("" ::"v"(lse2), "v"(delta));
    const LaneAddr la = lane_addr(lane);
    f32x16 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const int ntiles = N / 64;
    auto step = [&](int kt, unsigned boff) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ntiles) issue(kt + 1, boff ^ KVBUF);
        if (!valid) return;                                     // (wave-uniform) stage only
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const char* Ki = sm2 + boff + sub * TILEB;
            const char* Vi = sm2 + boff + 2 * TILEB + sub * TILEB;
            f32x16 s = rows_x_frags(Ki, la, qf);                 // S^T[key][q]
            const f32x16 dp = rows_x_frags(Vi, la, dof);         // dP^T[key][q]
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse2)) * (dp[r] - delta) * scale;   // dS^T
            imgT_x_p(dq, Ki, la, s);                             // dQ^T[d][q] += K^T dS^T
        }
    };
    for (int kt = 0; kt < ntiles; kt += 2) {
        step(kt, 0);
        if (kt + 1 < ntiles) step(kt + 1, KVBUF);
    }
    __syncthreads();
    if (valid)
        attns::store_rows_bf16(dq, 1.0f, (float*)sm2 + wid * 32 * OLD, dQKV + ((long)b * N + q0) * ld3 + h * DH, ld3, lane,
                               colpart ? colpart + ((long)b * (N / 32) + qb) * ld3 + h * DH : nullptr);
}

// ------------------------------------------------------------------------------------------------- forward, 32-key steps, three workgroups per CU
// [r3] At N = 576 a (batch, head) pair is 4.5 workgroups of 9 steps each and the per-block cost (dispatch, Q load, first tile's round trip,
// output store) is worth 5.4 steps (measured: 0.67 PF at N = 576 against 0.93 PF at N = 2304 on the same kernel).  With two workgroups per
// CU there is one partner to run under it.  The forward needs 160 VGPRs, so three waves per SIMD fit; what held it at two was LDS
// (64 KB of images, a 68 KB output patch).  This variant steps 32 keys at a time (32 KB of images) and stores its output through a
// half-width patch in two passes (35 KB): three workgroups per CU.  Same arithmetic and summation order as the 64-key kernel.
namespace att2 {
__device__ __forceinline__ void store_rows_bf16_2pass(const f32x16 (&o)[4], float mul, float* patch, __bf16* out, long ld, int lane) {
    constexpr int PLD = 68;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = o[2 * half + dd][4 * g + c] * mul;
                *(f32x4*)(patch + l31 * PLD + dd * 32 + 8 * g + 4 * lh) = v;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), c8 = lane & 7;
            const f32x4 a = *(const f32x4*)(patch + row * PLD + c8 * 8), b = *(const f32x4*)(patch + row * PLD + c8 * 8 + 4);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            store_bf16x8(out + (long)row * ld + half * 64 + c8 * 8, v);
        }
    }
}
}  // namespace att2

namespace att2 {
// o^T += V^T P for bf16 P fragments that already exist (the deferred form keeps a tile's P across a step)
__device__ __forceinline__ void imgT_x_ph(f32x16 (&o)[4], const char* img, const LaneAddr& la, const bf16x8 (&ph)[2]) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr0 + 4096 * ss + 512 * dt));
            const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr1 + 4096 * ss + 512 * dt));
            bf16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a[e];
                f[4 + e] = c[e];
            }
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, ph[ss], o[dt], 0, 0, 0);
        }
}
}  // namespace att2

template <int NW>
__global__ __launch_bounds__(64 * NW, 3) void attention_bf16s3_kernel(const __bf16* __restrict__ QKV, __bf16* __restrict__ CTX, int N, int heads,
                                                                     int qgroups, float scale_log2e, float* __restrict__ LSE) {
    using namespace att2;
    static_assert(NW >= 2 && NW <= 4, "the workgroup's waves share the 8 + 8 DMA pieces of a 32-key step");
    constexpr unsigned KVBUF = 2 * TILEB;                      // K image | V image
    constexpr float RESC = 6.0f;
    extern __shared__ __attribute__((aligned(16))) char sm3[];
    const int lin = xcd_lin(blockIdx.x, gridDim.x);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const int ld3 = 3 * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm3;
    constexpr int PPW = (8 + NW - 1) / NW;                     // pieces per wave: wave w takes pieces w, w + NW, ... < 8 of the K and of the V image
    unsigned ok[PPW];                                          // bytes from the tile's first row
#pragma unroll
    for (int i = 0; i < PPW; ++i) ok[i] = 2u * (unsigned)(D + dma_src(min(wid + i * NW, 7), lane, ld3));
    auto issue = [&](int kt, unsigned boff) __attribute__((always_inline)) {
        const unsigned long long src = uniform64(qkv + (long)(kt * 32) * ld3);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (wid + i * NW < 8) {                             // wave-uniform
                const unsigned a = lds0 + boff + 1024 * (wid + i * NW);
                dma16s(ok[i], src, a);
                dma16s(ok[i], src + 2ull * D, a + TILEB);
            }
        }
    };
    const int ntiles = N / 32;
    issue(0, 0);
    Frags<1> qf;
    attns::load_row_frags(qf, qkv + (long)(q0 + l31) * ld3, lh);
    attns::settle_frags(qf);
    const LaneAddr la = lane_addr(lane);
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // [r5] P V of tile t - 1 runs DURING the softmax of tile t.  Through round 4 a wave's step was the dependent chain  S (8 MFMAs) -> row maximum ->
    // 16 exp2 -> bf16 P -> transposed V reads -> P V (8 MFMAs): its VALU part had no matrix work of its own beside it (round-4 verdict: "the waves
    // run the same chain in phase").  Here the 8 MFMAs and 16 transposed reads of the PREVIOUS tile's P V are interleaved with the exponentials of this
    // tile, group by group (sched_group_barrier): chain per step  S -> {softmax(t) || P V(t - 1)}.  The V image of tile t - 1 lives one step longer:
    // the three K | V buffers hold tiles t - 1, t and (in flight) t + 1 -- ONE tile ahead (round 4 ran two ahead and measured that the DMA was never
    // late), requested right after the barrier that ends every wave's use of tile t - 2.
    // Measured (same-call A/Bs against the round-4 kernel, profiles/r05_attention_ab.log and DESIGN 3.13): 1.704 -> 1.685 ms per layer at B = 1024,
    // N = 576 and 5.36 -> 5.30 ms at N = 2304 in one call, 1.710 / 1.714 and 5.382 / 5.381 in another: at most one per cent, inside the run-to-run
    // spread; bit-identical context and log-sum-exp.  So the chain was NOT what holds the kernel at half the matrix pipe -- with three waves per SIMD
    // the partners already ran under each other's softmax.  What the SIMD runs out of is issue time: per 32 x 32 wave-step 16 MFMAs x 8 cycles of
    // issue, ~60 VALU, 16 transcendentals, 24 LDS reads and 4 LDS-DMA pieces at ~60 cycles each, times three waves, against 1536 matrix cycles.
    // (Also measured: the same deferral left to hipcc's scheduler at two waves per SIMD: 236 VGPRs, 1.98 ms; the next tile's DMA issued from
    // inside the softmax instead of behind the barrier: -0.8 % / -1.9 %.)
    // ONE code path writes O (two paths -- a plain P V before a rescale, an interleaved one otherwise -- made hipcc copy the accumulator tuples
    // between them: 210 spilled registers): the exponentials of tile t are taken against the NEW maximum while P V(t - 1) accumulates at the old
    // scale; O and l are rescaled after it, before P V(t) in the next step.
    bf16x8 ph[2];
    auto softmax_head = [&](const f32x16& s, float& alpha, bool& any) __attribute__((always_inline)) {
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
        const bool raise = mx > m_run + RESC;
        any = __builtin_amdgcn_ballot_w64(raise) != 0;
        const float m_new = raise ? mx : m_run;
        alpha = any ? exp2f(m_run - m_new) : 1.f;
        m_run = m_new;
    };
    auto open = [&](int kt, unsigned bnext) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ntiles) issue(kt + 1, bnext);
    };
    auto to_ph = [&](const f32x16& s) __attribute__((always_inline)) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int e = 0; e < 8; ++e) ph[ss][e] = (__bf16)s[8 * ss + e];
    };
    // tile 0: nothing pending (m_run = -inf: alpha = 0 on an O and l of zeros)
    open(0, KVBUF);
    if (valid) {
        f32x16 s = rows_x_frags(sm3, la, qf);
        float alpha;
        bool any;
        softmax_head(s, alpha, any);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
            psum += s[r];
        }
        l_run = psum;
        to_ph(s);
    }
    auto step = [&](int kt, unsigned boff, unsigned bnext, unsigned bprev) __attribute__((always_inline)) {
        open(kt, bnext);
        if (!valid) return;
        f32x16 s = rows_x_frags(sm3 + boff, la, qf);
        float alpha;
        bool any;
        softmax_head(s, alpha, any);
        float psum = 0.f;
        // 8 groups of { 1 MFMA of P V(t - 1) on fragments read one group ahead, the next group's 2 transposed reads, 2 exponentials of tile t }
        // (left to itself hipcc hoists all 16 reads and all the exponentials: 236 VGPRs, two waves per SIMD, 1.98 ms)
        const char* img = sm3 + bprev + TILEB;
        auto frag = [&](int g) __attribute__((always_inline)) {
            const int dt = g >> 1, ss = g & 1;
            const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr0 + 4096 * ss + 512 * dt));
            const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + la.tr1 + 4096 * ss + 512 * dt));
            bf16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a[e];
                f[4 + e] = c[e];
            }
            return f;
        };
        bf16x8 fc = frag(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            bf16x8 fn = fc;
            if (g < 7) fn = frag(g + 1);
            o[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc, ph[g & 1], o[g >> 1], 0, 0, 0);
#pragma unroll
            for (int r = 2 * g; r < 2 * g + 2; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
                psum += s[r];
            }
            fc = fn;
        }
        // the order above, spelled out for the machine scheduler (the IR optimiser sinks the exponentials below the last MFMA otherwise, and
        // scheduling fences between the groups cannot bring them back): per group 1 MFMA, the next group's 2 reads, 2 fma + 2 exp + 2 add
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        l_run = l_run * alpha + psum;
        to_ph(s);                                                // (before the branch below: hipcc's IR passes sink the exponentials to their first use otherwise, out of the scheduled region)
        asm volatile("" : "+v"(ph[0]), "+v"(ph[1]), "+v"(l_run));
        __builtin_amdgcn_sched_barrier(0);
        if (any) {                                               // (wave-uniform, rare: the maximum grew by more than 2^RESC)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
    };
    for (int kt = 1; kt < ntiles; kt += 3) {                 // buffer of tile t: t mod 3; tile t + 1 goes where tile t - 2 was
        step(kt, KVBUF, 2 * KVBUF, 0);
        if (kt + 1 < ntiles) step(kt + 1, 2 * KVBUF, 0, KVBUF);
        if (kt + 2 < ntiles) step(kt + 2, 0, KVBUF, 2 * KVBUF);
    }
    if (valid) imgT_x_ph(o, sm3 + ((ntiles - 1) % 3) * KVBUF + TILEB, la, ph);
    __syncthreads();
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (LSE != nullptr && lh == 0) LSE[(long)bh * N + q0 + l31] = m_run * 0.6931471805599453f + logf(l_tot);
        store_rows_bf16_2pass(o, 1.0f / l_tot, (float*)sm3 + wid * 32 * 68, CTX + ((long)b * N + q0) * D + h * DH, D, lane);
    }
}

template <int NW>
static hipError_t attention_bf16s3_fwd_launch_t(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream) {
    using namespace att2;
    constexpr size_t img = 3 * 2 * (size_t)TILEB, patch = (size_t)NW * 32 * 68 * 4;           // three K | V buffers
    constexpr size_t lds = img > patch ? img : patch;
    static_assert(NW == 2 || (12 / NW) * lds <= 160 * 1024, "twelve waves (three per SIMD) per CU (two-wave workgroups: five per CU, ten waves)");
    auto kern = attention_bf16s3_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int groups = (N / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3(B * heads * groups), dim3(64 * NW), lds, stream, QKV, CTX, N, heads, groups, 1.4426950408889634f / sqrtf(128.0f), LSE);
    return hipGetLastError();
}
static hipError_t attention_bf16s3_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream) {
    // four waves per workgroup: three-wave workgroups (no idle wave at N = 576: 18 query blocks = 6 x 3) measured equal at N = 576 and
    // 7 % slower at N = 2304, two-wave ones 20 % slower (each workgroup streams the pair's whole K and V)
    return attention_bf16s3_fwd_launch_t<4>(QKV, CTX, LSE, B, N, heads, stream);
}

static hipError_t attention_bf16s2_dq_launch(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                             int heads, hipStream_t stream, float* colpart) {
    using namespace att2;
    constexpr int NW = 4;
    constexpr size_t img = 2 * 4 * (size_t)TILEB, patch = (size_t)NW * 32 * OLD * 4;
    constexpr size_t lds = img > patch ? img : patch;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    auto kern = attn_bwd_dq_bf16s2_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int groups = (N / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3(B * heads * groups), dim3(64 * NW), lds, stream, QKV, O, dO, LSE, dQKV, DELTA, N, heads, groups, 1.0f / sqrtf((float)DH),
                       colpart);
    return hipGetLastError();
}
