// RETIRED from the shipped library in round 4 (kept for the record; not compiled).  Round 2's register-staged bf16-storage attention kernels:
// forward (`attention_bf16s_kernel`), dQ (`attn_bwd_dq_bf16s_kernel`, whose 32-key instantiation still ships as the fallback for sequence
// lengths that are not a multiple of 64) and dK + dV (`attn_bwd_dkv_bf16s_kernel`).  Same-box A/B against the DMA-staged kernels of
// attention_bf16s2.h at B = 1024, N = 576 (round 3, profiles/r03_config3_summary.md): forward 2.35 -> 1.74-1.80 ms,
// dQ 3.0 -> 2.6 ms, dK + dV 4.87 -> 3.25 ms per layer; bit-identical outputs (max |gen 1 - gen 2| = 0 on 450 M elements).
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
__global__ __launch_bounds__(64 * NW, 2) void attention_bf16s_kernel(const __bf16* __restrict__ QKV, __bf16* __restrict__ CTX, int N, int heads,
                                                                    int qgroups, float scale_log2e, float* __restrict__ LSE) {
    using namespace attns;
    constexpr int THREADS = 64 * NW, ROWS = 32 * SUB;
    constexpr float RESC = 6.0f;
    extern __shared__ __attribute__((aligned(16))) __bf16 simg_s[];
    __bf16* Kimg = simg_s;                     // [ROWS][RSTR]
    __bf16* Vimg = simg_s + SUB * RIMG;        // [ROWS][TSTR]
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = bid & 7;
    const int lin = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (bid >> 3);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld = 3L * D;
    const __bf16* base = QKV + (long)b * N * ld + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;            // invalid waves run on clamped rows (EXEC all ones around the transposing reads) and skip the store
    const int q0 = min(qb * 32, N - 32);
    Frags<1> qf;
    load_row_frags(qf, base + (long)(q0 + l31) * ld, lh);
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // m_run in base-2 units (score * scale_log2e)
    TileRegs<THREADS, ROWS> tk, tv;
    tile_load(tk, base + D, ld, tid);
    tile_load(tv, base + 2 * D, ld, tid);
    const int ntiles = N / ROWS;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                       // every wave is done with the previous tile's images
        tile_store(tk, Kimg, nullptr, tid);
        tile_store(tv, nullptr, Vimg, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {                 // next tile's loads fly during this tile's MFMAs
            tile_load(tk, base + (long)((kt + 1) * ROWS) * ld + D, ld, tid);
            tile_load(tv, base + (long)((kt + 1) * ROWS) * ld + 2 * D, ld, tid);
        }
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            f32x16 s = tile_x_frags<1>(Kimg + sub * RIMG, qf, l31, lh);     // S^T[key][q]
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
            const bool raise = mx > m_run + RESC;
            if (__builtin_amdgcn_ballot_w64(raise) != 0) {                   // rare after the first tile: wave-uniform branch
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
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
                psum += s[r];
            }
            l_run += psum;
            acc_tile_t_x_p<1>(o, Vimg + sub * TIMG, s, lane);                // O^T[d][q] += V^T P^T
        }
    }
    __syncthreads();
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (LSE != nullptr && lh == 0) LSE[(long)bh * N + q0 + l31] = m_run * 0.6931471805599453f + logf(l_tot);
        store_rows_bf16(o, 1.0f / l_tot, (float*)simg_s + wid * 32 * OLD, CTX + ((long)b * N + q0) * D + h * DH, D, lane);
    }
}

// ------------------------------------------------------------------------------------------------- dQ (+ delta)
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
template <int NW, int SUB>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_bf16s_kernel(const __bf16* __restrict__ QKV, const __bf16* __restrict__ dO,
                                                                       const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                       __bf16* __restrict__ dQKV, int N, int heads, int kgroups, float scale) {
    using namespace attns;
    constexpr int THREADS = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) __bf16 bsm_s[];
    constexpr int ROWS = 32 * SUB;
    __bf16* Qrow = bsm_s;
    __bf16* Qtr = Qrow + SUB * RIMG;
    __bf16* Drow = Qtr + SUB * TIMG;
    __bf16* Dtr = Drow + SUB * RIMG;
    float* Ls = (float*)(Dtr + SUB * TIMG);    // [ROWS] lse (log2 units), [ROWS] delta
    __bf16* Vw = (__bf16*)(Ls + 2 * ROWS);     // per wave: row image of the V rows of its 32 keys (registers hold K, dK, dV)
    const int lin = xcd_lin(blockIdx.x, gridDim.x);          // XCD-aware order: the key blocks of one (batch, head) share Q / dO tiles in one L2
    const int bh = lin / kgroups, kg = lin - bh * kgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int kb = kg * NW + wid;
    const bool valid = kb * 32 < N;
    const int k0 = min(kb * 32, N - 32);
    Frags<1> kf;
    load_row_frags(kf, qkv + (long)(k0 + l31) * ld3 + D, lh);
    __bf16* Vmine = Vw + wid * RIMG;
    stage<64>(Vmine, nullptr, qkv + (long)k0 * ld3 + 2 * D, ld3, lane);
    const float c2 = scale * 1.4426950408889634f;
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
    // no register prefetch of the next tile here: dK + dV hold 128 accumulators, the K fragments 32 more, and 16 staging registers
    // on top spill in the tile loop (measured 5.3 -> 7.2 ms); the CU's second workgroup covers the load latency instead
    const __bf16* dob = dO + (long)b * N * D + h * DH;
    const int ntiles = N / ROWS;
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            stage<THREADS>(Qrow + sub * RIMG, Qtr + sub * TIMG, qkv + (long)(qt * ROWS + 32 * sub) * ld3, ld3, tid);
            stage<THREADS>(Drow + sub * RIMG, Dtr + sub * TIMG, dob + (long)(qt * ROWS + 32 * sub) * D, D, tid);
        }
        {
            const float* lp = LSE + (long)bh * N + qt * ROWS;
            const float* dp_ = DELTA + (long)bh * N + qt * ROWS;
            if (tid < ROWS) Ls[tid] = lp[(unsigned)tid] * 1.4426950408889634f;
            else if (tid < 2 * ROWS) Ls[tid] = dp_[(unsigned)(tid - ROWS)];
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            f32x16 p = tile_x_frags<1>(Qrow + sub * RIMG, kf, l31, lh);              // S[q][key]
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(fmaf(p[r], c2, -Ls[32 * sub + (r & 3) + 8 * (r >> 2) + 4 * lh]));
            acc_tile_t_x_p<1>(dv, Dtr + sub * TIMG, p, lane);                        // dV^T[d][key] += dO^T P
            const f32x16 dp = tile_x_tile<1>(Drow + sub * RIMG, Vmine, l31, lh);     // dP[q][key] = dO V^T
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = p[r] * (dp[r] - Ls[ROWS + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * lh]) * scale;   // dS[q][key]
            acc_tile_t_x_p<1>(dk, Qtr + sub * TIMG, p, lane);                        // dK^T[d][key] += Q^T dS
        }
    }
    __syncthreads();
    if (valid) {
        float* patch = (float*)bsm_s + wid * 32 * OLD;
        __bf16* dst = dQKV + ((long)b * N + k0) * ld3 + h * DH;
        store_rows_bf16(dk, 1.0f, patch, dst + D, ld3, lane);
        store_rows_bf16(dv, 1.0f, patch, dst + 2 * D, ld3, lane);    // same wave, same patch: program order
    }
}

template <int SUB>
static hipError_t attention_bf16s_fwd_launch_t(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream) {
    using namespace attns;
    constexpr int NW = 4;
    constexpr size_t img = (size_t)SUB * (RIMG + TIMG) * 2, patch = (size_t)NW * 32 * OLD * 4;
    constexpr size_t lds = img > patch ? img : patch;
    auto kern = attention_bf16s_kernel<NW, SUB>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int qgroups = (N / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3(B * heads * qgroups), dim3(64 * NW), lds, stream, QKV, CTX, N, heads, qgroups, 1.4426950408889634f / sqrtf(128.0f), LSE);
    return hipGetLastError();
}
static hipError_t attention_bf16s2_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream);      // attention_bf16s2.h
static hipError_t attention_bf16s3_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream);      // attention_bf16s2.h
static hipError_t attention_bf16s_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream, int gen = 3) {
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    if (gen == 3) return attention_bf16s3_fwd_launch(QKV, CTX, LSE, B, N, heads, stream);      // 32-key steps, three workgroups per CU
    if (gen >= 2 && N % 64 == 0) return attention_bf16s2_fwd_launch(QKV, CTX, LSE, B, N, heads, stream);
    return N % 64 == 0 ? attention_bf16s_fwd_launch_t<2>(QKV, CTX, LSE, B, N, heads, stream) : attention_bf16s_fwd_launch_t<1>(QKV, CTX, LSE, B, N, heads, stream);
}

static hipError_t attention_bf16s2_dkv_launch(const __bf16* QKV, const __bf16* dO, const float* LSE, const float* DELTA, __bf16* dQKV, int B, int N,
                                              int heads, hipStream_t stream, float* colpart);      // attention_bf16s2.h
static hipError_t attention_bf16s2_dq_launch(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                             int heads, hipStream_t stream, float* colpart);

template <int SUB>
static hipError_t attention_bf16s_bwd_launch_t(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                               int heads, hipStream_t stream, int gen, float* colpart) {
    using namespace attns;
    constexpr int NW = 4;
    const float scale = 1.0f / sqrtf((float)DH);
    const int groups = (N / 32 + NW - 1) / NW;
    constexpr size_t patch = (size_t)NW * 32 * OLD * 4;
    constexpr int SKV = 1;       // the dK + dV kernel keeps four images + a per-wave V image: 64-row tiles would leave one workgroup per CU
    constexpr size_t img_q = (size_t)SUB * (2 * RIMG + TIMG) * 2, img_kv = (size_t)SKV * (2 * RIMG + 2 * TIMG) * 2 + SKV * 256 + (size_t)NW * RIMG * 2;
    constexpr size_t lds_q = img_q > patch ? img_q : patch, lds_kv = img_kv > patch ? img_kv : patch;
    static_assert(2 * lds_kv <= 160 * 1024 && 2 * lds_q <= 160 * 1024, "two workgroups per CU");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_bf16s_kernel<NW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dkv_bf16s_kernel<NW, SKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (gen >= 2 && SUB == 2) {       // N % 64 == 0: the DMA-staged dQ kernel walks the keys 64 at a time
        hipError_t e = attention_bf16s2_dq_launch(QKV, O, dO, LSE, DELTA, dQKV, B, N, heads, stream, colpart);
        if (e != hipSuccess) return e;
    } else
    hipLaunchKernelGGL((attn_bwd_dq_bf16s_kernel<NW, SUB>), dim3(B * heads * groups), dim3(64 * NW), lds_q, stream, QKV, O, dO, LSE, dQKV, DELTA, N, heads, groups, scale);
    if (gen >= 2) return attention_bf16s2_dkv_launch(QKV, dO, LSE, DELTA, dQKV, B, N, heads, stream, colpart);
    hipLaunchKernelGGL((attn_bwd_dkv_bf16s_kernel<NW, SKV>), dim3(B * heads * groups), dim3(64 * NW), lds_kv, stream, QKV, dO, LSE, DELTA, dQKV, N, heads, groups, scale);
    return hipGetLastError();
}
static hipError_t attention_bf16s_bwd_launch(const __bf16* QKV, const __bf16* O, const __bf16* dO, const float* LSE, float* DELTA, __bf16* dQKV, int B, int N,
                                             int heads, hipStream_t stream, int gen = 2, float* colpart = nullptr) {
    // colpart (generation 2, N % 64 == 0 only; the caller checks with attention_bf16s_bwd_colsums()): fp32 [B * N / 32][3 * heads * 128]
    // partial column sums of dQKV, one row per 32-row block -- summed over the rows they give the q | k | v bias gradients
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    if (colpart != nullptr && !(gen >= 2 && N % 64 == 0)) return hipErrorInvalidValue;
    return N % 64 == 0 ? attention_bf16s_bwd_launch_t<2>(QKV, O, dO, LSE, DELTA, dQKV, B, N, heads, stream, gen, colpart)
                       : attention_bf16s_bwd_launch_t<1>(QKV, O, dO, LSE, DELTA, dQKV, B, N, heads, stream, gen, colpart);
}
