// Backward of the fused softmax attention (autograd of modeling_vit.py:233-252), fp32 on the matrix cores,
// flash style: P is recomputed from Q, K and the forward's per-row log-sum-exp; the N x N matrices never exist.
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta) / sqrt(dh),  delta_q = sum_d dO[q,d] O[q,d],
//   dQ = dS K,    dK = dS^T Q.
// dQ needs a sum over keys per query and dK / dV sums over queries per key; instead of float atomics (slow, and not
// reproducible) the query side and the key side each get a kernel that owns its accumulators: 7 MFMA products instead of 5,
// every sum in a fixed order.
//   attn_bwd_dq_kernel  : wave = 32 queries on the lanes, streams key tiles   (S^T, dP^T, dQ^T; also writes delta)
//   attn_bwd_dkv_kernel : wave = 32 keys on the lanes,    streams query tiles (S, dP, dV^T, dK^T); V of the wave's keys in LDS
// Both use the forward kernel's tricks: the score tile's accumulator registers are, as they stand, the B operand
// of the next product; the fixed operand lives in 64 registers (k-permuted float4 loads); tiles are padded to 132
// floats for conflict-free ds_read_b128 fragments; results leave through an LDS transpose as whole 512-byte rows.
// Gradients are written in the fused [B*N, 3*D] q|k|v layout the QKV input-gradient GEMM reads directly.
#pragma once
#include "common.h"
#include <math.h>

namespace attnbwd {
constexpr int DH = 128, KT = 32, LD = DH + 4;

// stage a 32-row x 128 tile (row stride ld floats in memory) into LDS rows of LD floats
template <int THREADS>
__device__ __forceinline__ void stage_tile(float* dst, const float* src, long ld, int tid) {
    constexpr int PER = KT * (DH / 4) / THREADS;
    f32x4 st[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
        st[i] = *(const f32x4*)(src + (long)row * ld + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
        *(f32x4*)(dst + row * LD + c4 * 4) = st[i];
    }
}

// 64 registers of one 128-float row in the k-permuted order (lane half h: d = 8t + 4h + u)
__device__ __forceinline__ void load_row_regs(float (&reg)[64], const float* rowp, int lh) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 v = *(const f32x4*)(rowp + 8 * t + 4 * lh);
        reg[4 * t + 0] = v[0]; reg[4 * t + 1] = v[1]; reg[4 * t + 2] = v[2]; reg[4 * t + 3] = v[3];
    }
}

// T[32x32] = A_tile(rows from LDS, b128 fragments) x B_regs^T : acc row = LDS row, acc col = lane's fixed row
__device__ __forceinline__ f32x16 tile_x_regs(const float* tile, const float (&reg)[64], int l31, int lh) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* f = tile + l31 * LD + 4 * lh;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 a = *(const f32x4*)(f + 8 * t);
#pragma unroll
        for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], reg[4 * t + u], s, 0, 0, 0);
    }
    return s;
}

// acc^T[4 d-tiles] += sum_rows tile[row][d] * p[row][lane] : p's accumulator registers are the B operand as they stand
__device__ __forceinline__ void acc_tile_t_x_p(f32x16 (&o)[4], const float* tile, const f32x16& p, int l31, int lh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* vf = tile + ((r & 3) + 8 * (r >> 2) + 4 * lh) * LD + l31;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dt * 32], p[r], o[dt], 0, 0, 0);
    }
}

// write a wave's [4][32 d x 32 lane-rows] accumulators as 32 rows of 128 floats (row stride ld) through an LDS patch
__device__ __forceinline__ void store_rows(const f32x16 (&o)[4], float scale, float* patch, float* out, long ld, int l31, int lh) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c] * scale;
            *(f32x4*)(patch + l31 * LD + dt * 32 + 8 * g + 4 * lh) = v;
        }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = it * 2 + lh;
        *(f32x4*)(out + (long)row * ld + l31 * 4) = *(const f32x4*)(patch + row * LD + l31 * 4);
    }
}
}  // namespace attnbwd

// ------------------------------------------------------------------------------------------------- dQ (+ delta)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_kernel(const float* __restrict__ QKV, const float* __restrict__ O,
                                                                 const float* __restrict__ dO, const float* __restrict__ LSE,
                                                                 float* __restrict__ dQKV, float* __restrict__ DELTA, int N,
                                                                 int heads, int qgroups, float scale) {
    using namespace attnbwd;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                  // [32][132]
    float* Vs = smem + KT * LD;        // [32][132]
    // [r3] XCD-aware block order (as the forward and the bf16 kernels): the query blocks of one (batch, head) stream the same K / V
    // tiles; dealt round-robin over the 8 XCDs every L2 fetched them again (7.5 GB of L2-side reads per launch against 2.4 GB)
    const int lin_ = xcd_lin(blockIdx.x, gridDim.x);
    const int bh = lin_ / qgroups, qg = lin_ - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const float* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const long orow = ((long)b * N + q0 + l31) * D + h * DH;

    float qreg[64], doreg[64];
    load_row_regs(qreg, qkv + (long)(q0 + l31) * ld3, lh);
    load_row_regs(doreg, dO + orow, lh);
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 o4 = *(const f32x4*)(O + orow + 8 * t + 4 * lh);
#pragma unroll
        for (int u = 0; u < 4; ++u) delta += o4[u] * doreg[4 * t + u];
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse = LSE[(long)bh * N + q0 + l31];
    if (valid && lh == 0) DELTA[(long)bh * N + q0 + l31] = delta;
    const float c2 = scale * 1.4426950408889634f, lse2 = lse * 1.4426950408889634f;

    f32x16 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kt = 0; kt < N / KT; ++kt) {
        __syncthreads();
        stage_tile<64 * NW>(Ks, qkv + (long)(kt * KT) * ld3 + D, ld3, tid);
        stage_tile<64 * NW>(Vs, qkv + (long)(kt * KT) * ld3 + 2 * D, ld3, tid);
        __syncthreads();
        if (valid) {
            f32x16 s = tile_x_regs(Ks, qreg, l31, lh);          // S^T[key][q]
            const f32x16 dp = tile_x_regs(Vs, doreg, l31, lh);  // dP^T[key][q]
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = exp2f(fmaf(s[r], c2, -lse2)) * (dp[r] - delta) * scale;   // dS^T
            acc_tile_t_x_p(dq, Ks, s, l31, lh);                 // dQ^T[d][q] += K^T dS^T
        }
    }
    __syncthreads();
    if (valid) store_rows(dq, 1.0f, smem + wid * 32 * LD, dQKV + ((long)b * N + q0) * ld3 + h * DH, ld3, l31, lh);
}

// ------------------------------------------------------------------------------------------------- dK + dV
// One kernel owns both key-side gradients: S and P are computed once for the two of them (4 products: S, dP, dV, dK; with the
// dQ kernel's 3 that is 7 instead of the 8 of three separate kernels).  The next query tile (Q, dO rows) is requested into
// registers before the products of the current one and written to LDS after them, so the global latency is under the MFMAs.
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_dkv_kernel(const float* __restrict__ QKV, const float* __restrict__ dO,
                                                                  const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                  float* __restrict__ dQKV, int N, int heads, int kgroups,
                                                                  float scale) {
    using namespace attnbwd;
    constexpr int THREADS = 64 * NW, PER = KT * (DH / 4) / THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                              // [32][132]
    float* Ds = smem + KT * LD;                    // dO tile
    float* Ls = smem + 2 * KT * LD;                // [32] lse (log2 units), [32] delta
    float* Vw = smem + 2 * KT * LD + 64;           // per wave: V of its 32 keys [32][132]
    const int lin_ = xcd_lin(blockIdx.x, gridDim.x);          // [r3] the key blocks of one (batch, head) share their Q / dO tiles in one L2 (10.5 GB -> see profiles/r03)
    const int bh = lin_ / kgroups, kg = lin_ - bh * kgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const float* qkv = QKV + (long)b * N * ld3 + h * DH;
    const float* dop = dO + (long)b * N * D + h * DH;
    const int kb = kg * NW + wid;
    const bool valid = kb * 32 < N;
    const int k0 = min(kb * 32, N - 32);
    float kreg[64];
    load_row_regs(kreg, qkv + (long)(k0 + l31) * ld3 + D, lh);
    float* Vmine = Vw + wid * KT * LD;
    {   // this wave's V rows -> its private LDS image (read back as b128 fragments with the key on the lane)
        const float* vp = qkv + (long)k0 * ld3 + 2 * D;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = lane + i * 64, row = idx >> 5, c4 = idx & 31;
            *(f32x4*)(Vmine + row * LD + c4 * 4) = *(const f32x4*)(vp + (long)row * ld3 + c4 * 4);
        }
    }
    const float c2 = scale * 1.4426950408889634f;
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    f32x4 stq[PER], std_[PER];
    float stl = 0.f;
    auto request = [&](int qt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            stq[i] = *(const f32x4*)(qkv + (long)(qt * KT + row) * ld3 + c4 * 4);
            std_[i] = *(const f32x4*)(dop + (long)(qt * KT + row) * D + c4 * 4);
        }
        if (tid < 32) stl = LSE[(long)bh * N + qt * KT + tid] * 1.4426950408889634f;
        else if (tid < 64) stl = DELTA[(long)bh * N + qt * KT + tid - 32];
    };
    const int nq = N / KT;
    request(0);
    for (int qt = 0; qt < nq; ++qt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * THREADS, row = idx >> 5, c4 = idx & 31;
            *(f32x4*)(Qs + row * LD + c4 * 4) = stq[i];
            *(f32x4*)(Ds + row * LD + c4 * 4) = std_[i];
        }
        if (tid < 64) Ls[tid] = stl;
        __syncthreads();
        if (qt + 1 < nq) request(qt + 1);
        if (valid) {
            f32x16 s = tile_x_regs(Qs, kreg, l31, lh);          // S[q][key]
            // dP[q][key] = dO_tile V^T : B operand = V[key = lane][d] as b128 fragments of the wave's LDS image
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            {
                const float* af = Ds + l31 * LD + 4 * lh;
                const float* bf = Vmine + l31 * LD + 4 * lh;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const f32x4 a4 = *(const f32x4*)(af + 8 * t), b4 = *(const f32x4*)(bf + 8 * t);
#pragma unroll
                    for (int u = 0; u < 4; ++u) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u], dp, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;
                s[r] = exp2f(fmaf(s[r], c2, -Ls[q]));                        // P[q][key]
                dp[r] = s[r] * (dp[r] - Ls[32 + q]) * scale;                 // dS[q][key]
            }
            acc_tile_t_x_p(dv, Ds, s, l31, lh);                 // dV^T[d][key] += dO^T P
            acc_tile_t_x_p(dk, Qs, dp, l31, lh);                // dK^T[d][key] += Q^T dS
        }
    }
    __syncthreads();
    if (valid) {
        store_rows(dk, 1.0f, Vmine, dQKV + ((long)b * N + k0) * ld3 + D + h * DH, ld3, l31, lh);
        store_rows(dv, 1.0f, Vmine, dQKV + ((long)b * N + k0) * ld3 + 2 * D + h * DH, ld3, l31, lh);
    }
}

static hipError_t attention_bwd_f32_launch(const float* QKV, const float* O, const float* dO, const float* LSE, float* DELTA,
                                           float* dQKV, int B, int N, int heads, hipStream_t stream) {
    using namespace attnbwd;
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int NWQ = 4, NWK = 2;      // dK+dV: two waves per workgroup, two workgroups per CU (N = 576: 18 key blocks = 9 x 2, no idle wave; the
                                         // workgroups drift apart, so one's barrier and softmax phases sit under the other's MFMAs)
    const int qgroups = (N / 32 + NWQ - 1) / NWQ, kgroups = (N / 32 + NWK - 1) / NWK;
    const size_t lds_q = (size_t)(2 * KT * LD > NWQ * 32 * LD ? 2 * KT * LD : NWQ * 32 * LD) * 4;
    const size_t lds_k = (size_t)(2 * KT * LD + 64 + NWK * KT * LD) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NWQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NWK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel<NWQ>, dim3(B * heads * qgroups), dim3(64 * NWQ), lds_q, stream, QKV, O, dO, LSE, dQKV, DELTA,
                       N, heads, qgroups, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<NWK>, dim3(B * heads * kgroups), dim3(64 * NWK), lds_k, stream, QKV, dO, LSE, DELTA, dQKV, N,
                       heads, kgroups, scale);
    return hipGetLastError();
}
