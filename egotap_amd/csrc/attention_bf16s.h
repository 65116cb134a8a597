// Fused softmax attention with bf16 tensors in HBM (bf16-storage training mode, EGOTAP_PREC_BF16):
//   forward   ctx = softmax(Q K^T / sqrt(128)) V                                   (model/modeling_vit.py:226-252)
//   backward  flash style, P recomputed from Q, K and the forward's log-sum-exp:
//             dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta) / sqrt(dh),  dQ = dS K,  dK = dS^T Q,  delta = rowsum(dO * O)
// q | k | v are read in place from the fused bf16 [B*N, 3*heads*128] buffer and the three gradients are written in place into a
// bf16 buffer of the same layout (the operand of the QKV weight-gradient / input-gradient GEMMs).
// Same operand maps as attention_bf16.h / attention_bwd_bf16.h (v_mfma_f32_32x32x16_bf16; scores with the key on the accumulator
// row; probability accumulators used as the next product's B operand where they stand; transposed tiles through
// ds_read_b64_tr_b16) -- what changes: no conversion pass (tiles are staged as they lie, 16-byte loads, half the bytes), and the
// backward is TWO kernels instead of three:
//   attn_bwd_dq_bf16s_kernel   per 32-query block: S, dP, dQ (3 products) + delta
//   attn_bwd_dkv_bf16s_kernel  per 32-key block:   S, dV, dP, dK (4 products): S and dP are shared by the two gradient sums, the
//                              block's K rows stay in registers and its V rows in a per-wave LDS image, each gradient owns its accumulator (no float atomics,
//                              fixed summation order: bitwise reproducible).  7 products per tile pair instead of 8.
#pragma once
#include "attention_bwd_bf16.h"

namespace attns {
using namespace attnbf;

// stage a 32 x 128 bf16 tile (row stride ld elements) into a row image and / or a transposed-read image
template <int THREADS>
__device__ __forceinline__ void stage(__bf16* rowimg, __bf16* trimg, const __bf16* src, long ld, int tid) {
    constexpr int PER = KT * (DH / 8) / THREADS;          // 16-byte chunks per thread
    bf16x8 st[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 4, c8 = idx & 15;
        // wave-uniform base + 32-bit lane offset (global_load saddr form): a hoisted 64-bit pointer per lane and chunk spills in the
        // register-bound backward kernels, and a spill reload in the tile loop is a vmcnt-ordered memory operation
        st[i] = *(const bf16x8*)(src + (size_t)(unsigned)(row * (int)ld + c8 * 8));
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 4, c8 = idx & 15;
        if (rowimg) *(bf16x8*)(rowimg + row * RSTR + c8 * 8) = st[i];
        if (trimg) *(bf16x8*)(trimg + row * TSTR + c8 * 8) = st[i];
    }
}

// a ROWS x 128 bf16 tile in two steps (global -> registers, registers -> LDS images), so that the next tile's global loads are in
// flight during the current tile's MFMAs.  ROWS = 32 * SUB: SUB sub-tiles of 32 rows are staged per barrier pair.
template <int THREADS, int ROWS>
struct TileRegs { bf16x8 v[ROWS * (DH / 8) / THREADS]; };
template <int THREADS, int ROWS>
__device__ __forceinline__ void tile_load(TileRegs<THREADS, ROWS>& t, const __bf16* src, long ld, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * (DH / 8) / THREADS; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 4, c8 = idx & 15;
        t.v[i] = *(const bf16x8*)(src + (size_t)(unsigned)(row * (int)ld + c8 * 8));
    }
}
template <int THREADS, int ROWS>
__device__ __forceinline__ void tile_store(const TileRegs<THREADS, ROWS>& t, __bf16* rowimg, __bf16* trimg, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * (DH / 8) / THREADS; ++i) {
        const int idx = tid + i * THREADS, row = idx >> 4, c8 = idx & 15;
        if (rowimg) *(bf16x8*)(rowimg + row * RSTR + c8 * 8) = t.v[i];
        if (trimg) *(bf16x8*)(trimg + row * TSTR + c8 * 8) = t.v[i];
    }
}

// one 128-element bf16 row as 8 k-step fragments: lane half h of step s holds d = 16 s + 8 h + j
__device__ __forceinline__ void load_row_frags(Frags<1>& fr, const __bf16* rowp, int lh) {
#pragma unroll
    for (int s = 0; s < 8; ++s) fr.f[0][s] = *(const bf16x8*)(rowp + 16 * s + 8 * lh);
}

// [r4] "Use" every fragment of a row right after its loads, OUTSIDE the tile loop.  The tile loops below prefetch with LDS DMAs issued from
// inline asm and wait for them with hand-counted s_waitcnt: invisible to hipcc's waitcnt pass, which therefore still counted these plain
// loads as possibly pending at the loop header and put s_waitcnt vmcnt(7) ... vmcnt(0) in front of the first MFMAs of EVERY step (the
// first uses of the fragment registers).  vmcnt is one in-order counter: those waits drained the DMAs of the next tile, issued a few
// instructions earlier -- the double buffering overlapped nothing (all three kernels: MFMA busy 0.37-0.44 in profiles/r04_config3_summary.md).
__device__ __forceinline__ void settle_frags(const Frags<1>& fr) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(fr.f[0][s]));
}

// a wave's [4][32 d x 32 lane-rows] accumulators (times mul) as 32 rows of 128 bf16 (row stride ld) through an fp32 LDS patch.
// [r3] colpart (optional): 128 floats that receive the column sums of the 32 STORED (bf16) rows -- the per-block share of the
// q / k / v bias gradient, so that no kernel has to read the gradient tensor again just to sum its columns.
__device__ __forceinline__ void store_rows_bf16(const f32x16 (&o)[4], float mul, float* patch, __bf16* out, long ld, int lane, float* colpart = nullptr) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c] * mul;
            *(f32x4*)(patch + l31 * OLD + dt * 32 + 8 * g + 4 * lh) = v;
        }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), c8 = lane & 15;
        const f32x4 a = *(const f32x4*)(patch + row * OLD + c8 * 8), b = *(const f32x4*)(patch + row * OLD + c8 * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        store_bf16x8(out + (long)row * ld + c8 * 8, v);
        if (colpart != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[i] += (float)(__bf16)v[i];
        }
    }
    if (colpart != nullptr) {          // lanes l, l + 16, l + 32, l + 48 hold the same 8 columns for different rows
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            cs[i] += __shfl_xor(cs[i], 16, 64);
            cs[i] += __shfl_xor(cs[i], 32, 64);
        }
        if (lane < 16) {
            *(f32x4*)(colpart + lane * 8) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *(f32x4*)(colpart + lane * 8 + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
        }
    }
}
}  // namespace attns

// ------------------------------------------------------------------------------------------------- forward
// SUB sub-tiles of 32 keys per barrier pair.  The running maximum is only raised (and the output accumulators rescaled) when a
// query's scores exceed it by more than 2^RESC in the softmax's base-2 units (cdna_hip_programming.md T13): probabilities then
// stay below 2^RESC, harmless in fp32 sums and bf16 operands, and the 64 multiplies per sub-tile disappear from almost every tile.
template <int NW, int SUB>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_bf16s_kernel(const __bf16* __restrict__ QKV, const __bf16* __restrict__ O,
                                                                      const __bf16* __restrict__ dO, const float* __restrict__ LSE,
                                                                      __bf16* __restrict__ dQKV, float* __restrict__ DELTA, int N, int heads,
                                                                      int qgroups, float scale) {
    using namespace attns;
    constexpr int THREADS = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) __bf16 bsm_s[];
    constexpr int ROWS = 32 * SUB;
    __bf16* Krow = bsm_s;
    __bf16* Ktr = Krow + SUB * RIMG;
    __bf16* Vrow = Ktr + SUB * TIMG;
    // XCD-aware block order (as the forward): the workgroups of one (batch, head) stream the same K / V tiles; dealt round-robin over
    // the 8 XCDs each would fetch them into its own L2 (measured: 15 GB of L2-side reads per launch against 6 GB algorithmic)
    const int lin = xcd_lin(blockIdx.x, gridDim.x);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const long orow = ((long)b * N + q0 + l31) * D + h * DH;
    Frags<1> qf, dof;
    load_row_frags(qf, qkv + (long)(q0 + l31) * ld3, lh);
    load_row_frags(dof, dO + orow, lh);
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
    f32x16 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    TileRegs<THREADS, ROWS> tk, tv;
    tile_load(tk, qkv + D, ld3, tid);
    tile_load(tv, qkv + 2 * D, ld3, tid);
    const int ntiles = N / ROWS;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        tile_store(tk, Krow, Ktr, tid);
        tile_store(tv, Vrow, nullptr, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {
            tile_load(tk, qkv + (long)((kt + 1) * ROWS) * ld3 + D, ld3, tid);
            tile_load(tv, qkv + (long)((kt + 1) * ROWS) * ld3 + 2 * D, ld3, tid);
        }
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            f32x16 s = tile_x_frags<1>(Krow + sub * RIMG, qf, l31, lh);             // S^T[key][q]
            const f32x16 dp = tile_x_frags<1>(Vrow + sub * RIMG, dof, l31, lh);     // dP^T[key][q]
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse2)) * (dp[r] - delta) * scale;   // dS^T
            acc_tile_t_x_p<1>(dq, Ktr + sub * TIMG, s, lane);                       // dQ^T[d][q] += K^T dS^T
        }
    }
    __syncthreads();
    if (valid) store_rows_bf16(dq, 1.0f, (float*)bsm_s + wid * 32 * OLD, dQKV + ((long)b * N + q0) * ld3 + h * DH, ld3, lane);
}

// ------------------------------------------------------------------------------------------------- dK and dV
static hipError_t attention_bf16s3_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream);      // attention_bf16s2.h
static hipError_t attention_bf16s_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    return attention_bf16s3_fwd_launch(QKV, CTX, LSE, B, N, heads, stream);      // 32-key steps, three workgroups per CU
}

static hipError_t attention_bf16s2_dkv_launch(const __bf16* QKV, const __bf16* dO, const float* LSE, const float* DELTA, __bf16* dQKV, int B, int N,
                                              int heads, hipStream_t stream, float* colpart);      // attention_bf16s2.h
static hipError_t attention_bf16s2_dq_launch(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                             int heads, hipStream_t stream, float* colpart);

static hipError_t attention_bf16s_bwd_launch(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                             int heads, hipStream_t stream, float* colpart = nullptr) {
    // colpart (N % 64 == 0 only): fp32 [B * N / 32][3 * heads * 128] partial column sums of dQKV, one row per 32-row block -- summed over
    // the rows they give the q | k | v bias gradients
    using namespace attns;
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    if (colpart != nullptr && N % 64 != 0) return hipErrorInvalidValue;
    if (N % 64 == 0) {       // the DMA-staged dQ kernel walks the keys 64 at a time
        hipError_t e = attention_bf16s2_dq_launch(QKV, O, dO, LSE, DELTA, dQKV, B, N, heads, stream, colpart);
        if (e != hipSuccess) return e;
    } else {                 // other multiples of 32: the register-staged dQ kernel on 32-key tiles
        constexpr int NW = 4, SUB = 1;
        const int groups = (N / 32 + NW - 1) / NW;
        constexpr size_t patch = (size_t)NW * 32 * OLD * 4, img_q = (size_t)SUB * (2 * RIMG + TIMG) * 2;
        constexpr size_t lds_q = img_q > patch ? img_q : patch;
        static_assert(2 * lds_q <= 160 * 1024, "two workgroups per CU");
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_bf16s_kernel<NW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_bf16s_kernel<NW, SUB>), dim3(B * heads * groups), dim3(64 * NW), lds_q, stream, QKV, O, dO, LSE, dQKV, DELTA, N, heads, groups,
                           1.0f / sqrtf((float)DH));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return attention_bf16s2_dkv_launch(QKV, dO, LSE, DELTA, dQKV, B, N, heads, stream, colpart);
}
