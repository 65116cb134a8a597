// Backward of the fused softmax attention on the bf16 matrix cores (opt-in modes of egotap_set_precision), fp32 tensors in HBM.
// Same three kernels, same math and the same fixed summation orders as attention_bwd_f32.h (flash style: P is recomputed from
// Q, K and the forward's log-sum-exp; dQ, dV, dK each own their accumulator, no float atomics):
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta) / sqrt(dh),  dQ = dS K,  dK = dS^T Q.
// Every product runs on v_mfma_f32_32x32x16_bf16 with operands made in registers (NP = 3: hi + lo split, three MFMAs per
// product; NP = 1: bf16 rounding).  A 32 x 128 tile that is contracted over d ("T x regs^T": scores, dP) is staged as a
// ROW image (272-byte rows, one ds_read_b128 per 16-d step); a tile that is contracted over its rows ("T^T x P": the
// gradient accumulations) is staged as a TRANSPOSED-READ image (320-byte rows, ds_read_b64_tr_b16 blocks in the key order
// of the probability accumulator: 16s + 8(j>>2) + 4h + (j&3)); K in dQ and Q in dK are used both ways and get both images.
#pragma once
#include "attention_bf16.h"
#include "attention_bwd_f32.h"

namespace attnbf {
constexpr int DH = 128, KT = 32, RSTR = 136, TSTR = 160;      // image row strides in bf16
constexpr int RIMG = KT * RSTR, TIMG = KT * TSTR;              // bf16 per image
constexpr int OLD = DH + 4;                                    // fp32 output patch row

template <int NP> struct Frags { bf16x8 f[NP == 3 ? 2 : 1][8]; };

// stage a 32 x 128 fp32 tile (row stride ld) as bf16 images: row image (or nullptr) and transposed-read image (or nullptr);
// each is [hi][lo] for NP = 3
template <int THREADS, int NP>
__device__ __forceinline__ void stage(__bf16* rowimg, __bf16* trimg, const float* src, long ld, int tid) {
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
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (__bf16)st[i][e];
            lo[e] = (__bf16)(st[i][e] - (float)hi[e]);
        }
        if (rowimg) {
            *(bf16x4*)(rowimg + row * RSTR + c4 * 4) = hi;
            if (NP == 3) *(bf16x4*)(rowimg + RIMG + row * RSTR + c4 * 4) = lo;
        }
        if (trimg) {
            *(bf16x4*)(trimg + row * TSTR + c4 * 4) = hi;
            if (NP == 3) *(bf16x4*)(trimg + TIMG + row * TSTR + c4 * 4) = lo;
        }
    }
}

// one 128-float row as 8 k-step fragments: lane half h of step s holds d = 16 s + 8 h + j
template <int NP>
__device__ __forceinline__ void load_row_frags(Frags<NP>& fr, const float* rowp, int lh) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const f32x4 v0 = *(const f32x4*)(rowp + 16 * s + 8 * lh), v1 = *(const f32x4*)(rowp + 16 * s + 8 * lh + 4);
        if (NP == 3) bf16_split8(v0, v1, fr.f[0][s], fr.f[NP == 3 ? 1 : 0][s]);
        else fr.f[0][s] = bf16_round8(v0, v1);
    }
}

// T[32 x 32] = rowimg (rows on the accumulator row) x frags^T (the lane's fixed row on the accumulator column)
template <int NP>
__device__ __forceinline__ f32x16 tile_x_frags(const __bf16* rowimg, const Frags<NP>& fr, int l31, int lh) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const __bf16* p = rowimg + l31 * RSTR + 8 * lh;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bf16x8 ah = *(const bf16x8*)(p + 16 * t);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fr.f[0][t], s, 0, 0, 0);
        if (NP == 3) {
            const bf16x8 al = *(const bf16x8*)(p + RIMG + 16 * t);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fr.f[NP == 3 ? 1 : 0][t], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, fr.f[0][t], s, 0, 0, 0);
        }
    }
    return s;
}

// same with the second operand read from a (per-wave) row image instead of registers: T = A_img x B_img^T
template <int NP>
__device__ __forceinline__ f32x16 tile_x_tile(const __bf16* aimg, const __bf16* bimg, int l31, int lh) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const __bf16* pa = aimg + l31 * RSTR + 8 * lh;
    const __bf16* pb = bimg + l31 * RSTR + 8 * lh;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bf16x8 ah = *(const bf16x8*)(pa + 16 * t), bh = *(const bf16x8*)(pb + 16 * t);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, s, 0, 0, 0);
        if (NP == 3) {
            const bf16x8 al = *(const bf16x8*)(pa + RIMG + 16 * t), bl = *(const bf16x8*)(pb + RIMG + 16 * t);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, s, 0, 0, 0);
        }
    }
    return s;
}

// acc^T[4 d-tiles] += sum_rows tile[row][d] * p[row][lane]: p's accumulator registers (split in place) are the B operand,
// the tile comes back transposed from its row-major image through ds_read_b64_tr_b16
template <int NP>
__device__ __forceinline__ void acc_tile_t_x_p(f32x16 (&o)[4], const __bf16* trimg, const f32x16& p, int lane) {
    const int lh = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const __bf16* base = trimg + (4 * lh + tq) * TSTR + 16 * tg + 4 * tp;
    bf16x8 ph[2], pl[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pv = p[8 * ss + e];
            const __bf16 h = (__bf16)pv;
            ph[ss][e] = h;
            pl[ss][e] = (__bf16)(pv - (float)h);
        }
    auto frag = [&](const __bf16* q) __attribute__((always_inline)) {
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(q));
        const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(q + 8 * TSTR));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = c[e];
        }
        return f;
    };
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 th = frag(base + 16 * ss * TSTR + 32 * dt);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, ph[ss], o[dt], 0, 0, 0);
            if (NP == 3) {
                const bf16x8 tl = frag(base + TIMG + 16 * ss * TSTR + 32 * dt);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, pl[ss], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, ph[ss], o[dt], 0, 0, 0);
            }
        }
}

// write a wave's [4][32 d x 32 lane-rows] accumulators as 32 rows of 128 floats (row stride ld) through an fp32 LDS patch
__device__ __forceinline__ void store_rows(const f32x16 (&o)[4], float* patch, float* out, long ld, int l31, int lh) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = o[dt][4 * g + c];
            *(f32x4*)(patch + l31 * OLD + dt * 32 + 8 * g + 4 * lh) = v;
        }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = it * 2 + lh;
        *(f32x4*)(out + (long)row * ld + l31 * 4) = *(const f32x4*)(patch + row * OLD + l31 * 4);
    }
}
template <int NP> constexpr int rimg_pair() { return (NP == 3 ? 2 : 1) * RIMG; }
template <int NP> constexpr int timg_pair() { return (NP == 3 ? 2 : 1) * TIMG; }
}  // namespace attnbf

// ------------------------------------------------------------------------------------------------- dQ (+ delta)
// Invalid waves (query block past N) run the whole loop on clamped rows and skip only the final store: EXEC stays all
// ones around the transposing reads.
template <int NW, int NP>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_bf16_kernel(const float* __restrict__ QKV, const float* __restrict__ O,
                                                                      const float* __restrict__ dO, const float* __restrict__ LSE,
                                                                      float* __restrict__ dQKV, float* __restrict__ DELTA, int N,
                                                                      int heads, int qgroups, float scale) {
    using namespace attnbf;
    extern __shared__ __attribute__((aligned(16))) __bf16 bsm[];
    __bf16* Krow = bsm;
    __bf16* Ktr = Krow + rimg_pair<NP>();
    __bf16* Vrow = Ktr + timg_pair<NP>();
    const int bh = blockIdx.x / qgroups, qg = blockIdx.x - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const float* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const long orow = ((long)b * N + q0 + l31) * D + h * DH;

    Frags<NP> qf, dof;
    load_row_frags<NP>(qf, qkv + (long)(q0 + l31) * ld3, lh);
    load_row_frags<NP>(dof, dO + orow, lh);
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 o4 = *(const f32x4*)(O + orow + 8 * t + 4 * lh), d4 = *(const f32x4*)(dO + orow + 8 * t + 4 * lh);
#pragma unroll
        for (int u = 0; u < 4; ++u) delta += o4[u] * d4[u];
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
        stage<64 * NW, NP>(Krow, Ktr, qkv + (long)(kt * KT) * ld3 + D, ld3, tid);
        stage<64 * NW, NP>(Vrow, nullptr, qkv + (long)(kt * KT) * ld3 + 2 * D, ld3, tid);
        __syncthreads();
        f32x16 s = tile_x_frags<NP>(Krow, qf, l31, lh);            // S^T[key][q]
        const f32x16 dp = tile_x_frags<NP>(Vrow, dof, l31, lh);    // dP^T[key][q]
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = exp2f(fmaf(s[r], c2, -lse2)) * (dp[r] - delta) * scale;   // dS^T
        acc_tile_t_x_p<NP>(dq, Ktr, s, lane);                      // dQ^T[d][q] += K^T dS^T
    }
    __syncthreads();
    if (valid) store_rows(dq, (float*)bsm + wid * 32 * OLD, dQKV + ((long)b * N + q0) * ld3 + h * DH, ld3, l31, lh);
}

// ------------------------------------------------------------------------------------------------- dV
template <int NW, int NP>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dv_bf16_kernel(const float* __restrict__ QKV, const float* __restrict__ dO,
                                                                      const float* __restrict__ LSE, float* __restrict__ dQKV, int N,
                                                                      int heads, int kgroups, float scale) {
    using namespace attnbf;
    extern __shared__ __attribute__((aligned(16))) __bf16 bsm[];
    __bf16* Qrow = bsm;
    __bf16* Dtr = Qrow + rimg_pair<NP>();
    float* Ls = (float*)(Dtr + timg_pair<NP>());      // lse of the tile's 32 queries (log2 units)
    const int bh = blockIdx.x / kgroups, kg = blockIdx.x - bh * kgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const float* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int kb = kg * NW + wid;
    const bool valid = kb * 32 < N;
    const int k0 = min(kb * 32, N - 32);
    Frags<NP> kf;
    load_row_frags<NP>(kf, qkv + (long)(k0 + l31) * ld3 + D, lh);
    const float c2 = scale * 1.4426950408889634f;
    f32x16 dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[dt][r] = 0.f;
    for (int qt = 0; qt < N / KT; ++qt) {
        __syncthreads();
        stage<64 * NW, NP>(Qrow, nullptr, qkv + (long)(qt * KT) * ld3, ld3, tid);
        stage<64 * NW, NP>(nullptr, Dtr, dO + ((long)b * N + qt * KT) * D + h * DH, D, tid);
        if (tid < 32) Ls[tid] = LSE[(long)bh * N + qt * KT + tid] * 1.4426950408889634f;
        __syncthreads();
        f32x16 p = tile_x_frags<NP>(Qrow, kf, l31, lh);             // S[q][key]
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = exp2f(fmaf(p[r], c2, -Ls[(r & 3) + 8 * (r >> 2) + 4 * lh]));
        acc_tile_t_x_p<NP>(dv, Dtr, p, lane);                       // dV^T[d][key] += dO^T P
    }
    __syncthreads();
    if (valid) store_rows(dv, (float*)bsm + wid * 32 * OLD, dQKV + ((long)b * N + k0) * ld3 + 2 * D + h * DH, ld3, l31, lh);
}

// ------------------------------------------------------------------------------------------------- dK
template <int NW, int NP>
__global__ __launch_bounds__(64 * NW, NP == 1 ? 2 : 1) void attn_bwd_dk_bf16_kernel(const float* __restrict__ QKV, const float* __restrict__ dO,
                                                                      const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                      float* __restrict__ dQKV, int N, int heads, int kgroups,
                                                                      float scale) {
    using namespace attnbf;
    extern __shared__ __attribute__((aligned(16))) __bf16 bsm[];
    __bf16* Qrow = bsm;
    __bf16* Qtr = Qrow + rimg_pair<NP>();
    __bf16* Drow = Qtr + timg_pair<NP>();
    float* Ls = (float*)(Drow + rimg_pair<NP>());     // [32] lse (log2 units), [32] delta
    __bf16* Vw = (__bf16*)(Ls + 64);                  // per wave: row image (hi [+ lo]) of the V rows of its 32 keys
    const int bh = blockIdx.x / kgroups, kg = blockIdx.x - bh * kgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const long ld3 = 3L * D;
    const float* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int kb = kg * NW + wid;
    const bool valid = kb * 32 < N;
    const int k0 = min(kb * 32, N - 32);
    Frags<NP> kf;
    load_row_frags<NP>(kf, qkv + (long)(k0 + l31) * ld3 + D, lh);
    __bf16* Vmine = Vw + wid * rimg_pair<NP>();
    stage<64, NP>(Vmine, nullptr, qkv + (long)k0 * ld3 + 2 * D, ld3, lane);     // this wave's V rows -> its private row image
    const float c2 = scale * 1.4426950408889634f;
    f32x16 dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = 0.f;
    for (int qt = 0; qt < N / KT; ++qt) {
        __syncthreads();
        stage<64 * NW, NP>(Qrow, Qtr, qkv + (long)(qt * KT) * ld3, ld3, tid);
        stage<64 * NW, NP>(Drow, nullptr, dO + ((long)b * N + qt * KT) * D + h * DH, D, tid);
        if (tid < 32) Ls[tid] = LSE[(long)bh * N + qt * KT + tid] * 1.4426950408889634f;
        else if (tid < 64) Ls[tid] = DELTA[(long)bh * N + qt * KT + tid - 32];
        __syncthreads();
        f32x16 s = tile_x_frags<NP>(Qrow, kf, l31, lh);              // S[q][key]
        const f32x16 dp = tile_x_tile<NP>(Drow, Vmine, l31, lh);     // dP[q][key] = dO_tile V^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;
            s[r] = exp2f(fmaf(s[r], c2, -Ls[q])) * (dp[r] - Ls[32 + q]) * scale;     // dS[q][key]
        }
        acc_tile_t_x_p<NP>(dk, Qtr, s, lane);                        // dK^T[d][key] += Q^T dS
    }
    __syncthreads();
    // every image is dead after the barrier: the fp32 output patches lie over them from the base (plain bf16: 70 KB of images,
    // 68 KB of patches -- two workgroups per CU instead of one)
    if (valid) store_rows(dk, (float*)bsm + wid * 32 * OLD, dQKV + ((long)b * N + k0) * ld3 + D + h * DH, ld3, l31, lh);
}

template <int NP>
static hipError_t attention_bwd_bf16_launch(const float* QKV, const float* O, const float* dO, const float* LSE, float* DELTA,
                                            float* dQKV, int B, int N, int heads, hipStream_t stream) {
    using namespace attnbf;
    if (B <= 0) return hipSuccess;
    if (N % 32 != 0) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int NW = 4;
    const int groups = (N / 32 + NW - 1) / NW;
    constexpr size_t patch = (size_t)NW * 32 * OLD * 4;
    constexpr size_t img_q = (size_t)(2 * rimg_pair<NP>() + timg_pair<NP>()) * 2;
    constexpr size_t img_v = (size_t)(rimg_pair<NP>() + timg_pair<NP>()) * 2 + 128;
    constexpr size_t lds_q = img_q > patch ? img_q : patch, lds_v = img_v > patch ? img_v : patch;
    constexpr size_t img_k = (size_t)(2 * rimg_pair<NP>() + timg_pair<NP>()) * 2 + 256 + (size_t)NW * rimg_pair<NP>() * 2;
    constexpr size_t lds_k = img_k > patch ? img_k : patch;
    static_assert(lds_k <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<NW, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dv_bf16_kernel<NW, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dk_bf16_kernel<NW, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<NW, NP>), dim3(B * heads * groups), dim3(64 * NW), lds_q, stream, QKV, O, dO, LSE, dQKV,
                       DELTA, N, heads, groups, scale);
    hipLaunchKernelGGL((attn_bwd_dv_bf16_kernel<NW, NP>), dim3(B * heads * groups), dim3(64 * NW), lds_v, stream, QKV, dO, LSE, dQKV, N,
                       heads, groups, scale);
    hipLaunchKernelGGL((attn_bwd_dk_bf16_kernel<NW, NP>), dim3(B * heads * groups), dim3(64 * NW), lds_k, stream, QKV, dO, LSE, DELTA, dQKV,
                       N, heads, groups, scale);
    return hipGetLastError();
}
