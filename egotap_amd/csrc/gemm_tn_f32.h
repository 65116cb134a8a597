// Weight-gradient GEMM ("TN"): dW[N,K] = sum_m dY[m,n] * X[m,k]  -- the backward of y = x W^T + b w.r.t. W
// (autograd of nn.Linear / the patch-embedding conv; training step of egotap_autoencoder_model.py:299-323).
//
// Both operands are row-major with the CONTRACTED index m as the row, so a slab of BKM rows of dY and of X is staged
// exactly as it lies in memory (coalesced float4 rows, no transposes of the big activations) and the MFMA fragments
// are ds_read_b32 with the output row/column on the lane: A[i = n][k = m] = dY[m][n0 + lane], B[k = m][j] = X[m][k0 + lane].
// X goes through the same loader functors as the forward GEMM, so the gathers (ViT patch tiling, per-heatmap token
// regroup, stereo cos/sin interleave) cost nothing in the backward either.
// M is huge (147 k tokens at B = 256) and N x K small, so the M range is SPLIT over blocks: grid = tiles x splits,
// each block writes a partial [N,K] slab; reduce_slabs_kernel sums the slabs in a fixed order (bitwise reproducible,
// no float atomics) and optionally accumulates into the existing gradient.
#pragma once
#include "common.h"
#include "gemm_f32.h"
#include "lds_dma.h"

template <int BN_, int BK_, int BKM_, int WN_, int WK_>
struct TnCfg {
    static constexpr int BN = BN_, BK = BK_, BKM = BKM_, WN = WN_, WK = WK_;
    static constexpr int THREADS = 64 * WN * WK;
    static constexpr int LDA = BN + 4, LDB = BK + 4;          // padded rows (floats): the two lane halves read rows m, m+1
    static constexpr int TN = BN / WN / 32, TK = BK / WK / 32;
    static constexpr int A_V4 = BKM * BN / 4 / THREADS, B_V4 = BKM * BK / 4 / THREADS;
    static constexpr int LDS_BYTES = 2 * BKM * (LDA + LDB) * 4;
    static_assert(BKM % 2 == 0 && (BKM * BN / 4) % THREADS == 0 && (BKM * BK / 4) % THREADS == 0, "staging must divide evenly");
};

template <class Cfg, class XLoad>
__global__ __launch_bounds__(Cfg::THREADS, 2) void gemm_tn_f32_kernel(const float* __restrict__ dY, long ldy, XLoad xl,
                                                                        float* __restrict__ slabs, int M, int N, int K,
                                                                        int tiles_n, int tiles_k, int splits) {
    constexpr int BN = Cfg::BN, BK = Cfg::BK, BKM = Cfg::BKM, LDA = Cfg::LDA, LDB = Cfg::LDB;
    constexpr int TN = Cfg::TN, TK = Cfg::TK, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4, THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BKM][LDA]   dY rows
    float* Bs = smem + 2 * BKM * LDA;          // [2][BKM][LDB]   X rows
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / Cfg::WK, wk = wid % Cfg::WK;
    const int l31 = lane & 31, lh = lane >> 5;

    // block -> (tile, split); splits of one tile are consecutive so they share the L2-resident slab region
    const int bid = blockIdx.x;
    const int split = bid % splits, tile = bid / splits;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int n0 = tn * BN, k0 = tk * BK;
    const int rows_per = ((M + splits - 1) / splits + BKM - 1) / BKM * BKM;
    const int m_lo = split * rows_per, m_hi = min(M, m_lo + rows_per);
    const int nslab = m_hi > m_lo ? (m_hi - m_lo + BKM - 1) / BKM : 0;

    // staging: thread -> (row r, float4 column c) of each operand slab
    constexpr int A_C4 = BN / 4, B_C4 = BK / 4;
    f32x4 pa[A_V4], pb[B_V4];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int mb = m_lo + s * BKM;
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / A_C4, c = idx - r * A_C4;
            const int m = mb + r;
            pa[i] = m < m_hi ? *(const f32x4*)(dY + (long)m * ldy + n0 + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / B_C4, c = idx - r * B_C4;
            const int m = mb + r;
            pb[i] = m < m_hi ? xl.load(xl.row(m), k0 + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / A_C4, c = idx - r * A_C4;
            *(f32x4*)(As + (buf * BKM + r) * LDA + c * 4) = pa[i];
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / B_C4, c = idx - r * B_C4;
            *(f32x4*)(Bs + (buf * BKM + r) * LDB + c * 4) = pb[i];
        }
    };

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nslab > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(s + 1);
        const float* Ab = As + (buf * BKM + lh) * LDA + wn * (TN * 32) + l31;
        const float* Bb = Bs + (buf * BKM + lh) * LDB + wk * (TK * 32) + l31;
#pragma unroll
        for (int p = 0; p < BKM / 2; ++p) {
            float a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = Ab[2 * p * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = Bb[2 * p * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
    float* out = slabs + (long)split * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int kk = k0 + wk * (TK * 32) + j * 32 + l31;
            const int nb = n0 + wn * (TN * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + (r & 3) + 8 * (r >> 2);
                if (n < N && kk < K) out[(long)n * K + kk] = acc[i][j][r];
            }
        }
}

// dst[i] = (accumulate ? dst[i] : 0) + sum_s slabs[s][i], fixed summation order
static __global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ dst, long n,
                                                           int splits, int accumulate) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = accumulate ? *(const f32x4*)(dst + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) s += *(const f32x4*)(slabs + (long)k * n + i);
    *(f32x4*)(dst + i) = s;
}

// column sums (bias gradients): out[n] = (accumulate ? out[n] : 0) + sum_m Y[m][n].  Two stages, fixed order:
// stage 1 writes one partial row per block into part[gridDim.y][N]; stage 2 is reduce_slabs_kernel over those rows.
static __global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ Y, long ldy, float* __restrict__ part,
                                                             int M, int N, int rows_per_block) {
    __shared__ f32x4 red[4][64];
    const int c4 = blockIdx.x * 64 + (threadIdx.x & 63);      // float4 column
    const int q = threadIdx.x >> 6;                            // 4 row phases per block
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < N)
        for (int m = m_lo + q; m < m_hi; m += 4) s += *(const f32x4*)(Y + (long)m * ldy + c4 * 4);
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c4 * 4 < N) {
        const int l = threadIdx.x & 63;
        *(f32x4*)(part + (long)blockIdx.y * N + c4 * 4) = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    }
}

// out[C, ldo >= R] = in[R,C]^T (weights for the input-gradient GEMM; 32x32 tiles through LDS)
static __global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C, long ldo) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = by + ty + i * 8, c = bx + tx;
        t[ty + i * 8][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = bx + ty + i * 8, r = by + tx;
        if (r < R && c < C) out[(long)c * ldo + r] = t[tx][ty + i * 8];
    }
}

template <class Cfg, class XLoad>
static hipError_t gemm_tn_f32_launch(const float* dY, long ldy, const XLoad& xl, float* dW, float* slabs, size_t slab_bytes,
                                     int M, int N, int K, int num_cu, int accumulate, hipStream_t stream) {
    if (N % 4 != 0 || K % 4 != 0) return hipErrorInvalidValue;
    const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN, tiles_k = (K + Cfg::BK - 1) / Cfg::BK;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0) return hipErrorInvalidValue;
    const int tiles = tiles_n * tiles_k;
    int splits = (2 * num_cu + tiles - 1) / tiles;               // about two blocks per CU
    const int max_by_rows = (M + 4 * Cfg::BKM - 1) / (4 * Cfg::BKM);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    while ((size_t)splits * N * K * 4 > slab_bytes && splits > 1) --splits;
    const bool direct = splits == 1 && !accumulate;              // a single slab that is not added to anything IS the gradient
    if (!direct && (size_t)splits * N * K * 4 > slab_bytes) return hipErrorOutOfMemory;
    auto kern = gemm_tn_f32_kernel<Cfg, XLoad>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, dY, ldy, xl, direct ? dW : slabs, M, N, K,
                       tiles_n, tiles_k, splits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || direct) return e;
    const long n = (long)N * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, dW, n, splits,
                       accumulate);
    return hipGetLastError();
}

// [r3] How many splits of a weight-gradient launch (the convolution weight gradients of hm_train.h and the DMA-staged GEMM below; a slab = the
// rows staged between two barriers): workgroups go to the CUs round-robin, so a launch takes ceil(tiles * S / CUs) workgroups in a row on the busiest CU, each
// ceil(slabs / S) slabs long plus a fixed part (LDS zeroing, first loads, the slab store), and the reduction reads S slabs afterwards.  Round 2
// took S = 2 CUs / tiles over whole images: 560 workgroups of 5 images on 256 CUs for conv_up1 (three rounds for 2.2 rounds of work), 128
// workgroups for the 64-channel layers (half the chip idle).  Times in microseconds; only their ratios matter.
// Two workgroups fit on a CU where the LDS allows (the registers never allow more): they share the matrix pipe, and one's staging and barriers
// hide under the other's MFMAs.
static int wgrad_pick_splits(int tiles, long slabs, long n_floats, size_t slab_bytes, int num_cu, int lds_bytes, double slab_us, double fixed_us, int* per_out) {
    const int wpc = 2 * lds_bytes <= 160 * 1024 ? 2 : 1;
    long max_s = (long)(slab_bytes / ((size_t)n_floats * 4));
    if (max_s < 1) return 0;
    if (max_s > slabs) max_s = slabs;
    if (max_s > 4096) max_s = 4096;
    // a training step asks the same few questions every time: remember the last answers (per host thread; the scan below is up to 4096 candidates)
    struct Memo { int tiles; long slabs, n_floats, max_s; int num_cu, wpc; double slab_us, fixed_us; int s, per; };
    static thread_local Memo memo[32];
    static thread_local int memo_n = 0, memo_next = 0;
    for (int i = 0; i < memo_n; ++i) {
        const Memo& m = memo[i];
        if (m.tiles == tiles && m.slabs == slabs && m.n_floats == n_floats && m.max_s == max_s && m.num_cu == num_cu && m.wpc == wpc && m.slab_us == slab_us &&
            m.fixed_us == fixed_us) {
            *per_out = m.per;
            return m.s;
        }
    }
    double best = 1e30;
    int best_s = 1;
    for (long s = 1; s <= max_s; ++s) {
        const long per = (slabs + s - 1) / s, s_eff = (slabs + per - 1) / per;
        if (s_eff != s) continue;                           // the same launch as a smaller S
        const long in_a_row = ((long)tiles * s + num_cu - 1) / num_cu;
        const double alone_us = wpc == 2 && in_a_row >= 2 ? 0.0 : 0.25;      // a slab's barrier and LDS stores, exposed when nobody shares the CU
        const double t = in_a_row * per * (slab_us + alone_us) + (in_a_row + wpc - 1) / wpc * fixed_us + (double)s * n_floats * 4.0 / 3.0e6 + 0.02 * s;
        if (t < best * 0.995) { best = t; best_s = (int)s; }
    }
    *per_out = (int)((slabs + best_s - 1) / best_s);
    memo[memo_next] = Memo{tiles, slabs, n_floats, max_s, num_cu, wpc, slab_us, fixed_us, best_s, *per_out};
    memo_next = (memo_next + 1) % 32;
    if (memo_n < 32) ++memo_n;
    return best_s;
}

// [r3] The same weight-gradient GEMM with global -> LDS DMA staging, for operands that are plain row-major matrices (every weight gradient of
// the ViT layers and fc2 / fc3: 83 of the fp32 training step's 324 ms).  A slab is 32 rows of dY and 32 rows of X as they lie in memory: one
// 1 KiB row segment per DMA instruction (lane = 16 bytes), written to the padded LDS rows of the register-staged kernel above, so fragment
// addressing, MFMA order and the summation order per output element are unchanged.  Two stages: the slab after the current one is in
// flight during the current one's 128 MFMAs per wave; no staging registers, no ds_write, half the barriers (32-row slabs).
// Needs M % 32 == 0 (every slab full: the DMA cannot write the zero rows of a ragged tail) and 16-byte aligned rows.
struct TnDmaCfg {
    static constexpr int BN = 256, BK = 256, BKM = 32, WN = 4, WK = 2, THREADS = 512, TN = 2, TK = 4, NS = 2;
    static constexpr int LDA = BN + 4, LDB = BK + 4;
    static constexpr int STAGE_FLOATS = BKM * (LDA + LDB);
    static constexpr int LDS_BYTES = NS * STAGE_FLOATS * 4;             // 133 120
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

static __global__ __launch_bounds__(TnDmaCfg::THREADS) void gemm_tn_f32_dma_kernel(const float* __restrict__ dY, long ldy, const float* __restrict__ X,
                                                                                  long ldx, float* __restrict__ slabs, int M, int N, int K,
                                                                                  int tiles_n, int tiles_k, int splits, int rows_per,
                                                                                  float* __restrict__ csum) {
    using Cfg = TnDmaCfg;
    constexpr int BN = Cfg::BN, BK = Cfg::BK, BKM = Cfg::BKM, LDA = Cfg::LDA, LDB = Cfg::LDB, TN = Cfg::TN, TK = Cfg::TK, SF = Cfg::STAGE_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem_tnd[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / Cfg::WK, wk = wid % Cfg::WK;
    const int l31 = lane & 31, lh = lane >> 5;
    const int bid = blockIdx.x;
    const int split = bid % splits, tile = bid / splits;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int n0 = tn * BN, k0 = tk * BK;
    const int m_lo = split * rows_per, m_hi = min(M, m_lo + rows_per);
    const int nslab = m_hi > m_lo ? (m_hi - m_lo) / BKM : 0;        // rows_per % BKM == 0 and M % BKM == 0: every slab is full

    // the DMA goes through inline asm and vmcnt is waited for by hand (see gemm_f32_dma.h for why)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_tnd;
    auto dma1 = [&](const float* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    // wave w stages rows w, w + 8, w + 16, w + 24 of both operands: 8 DMA instructions per wave and slab
    // [r5] a staged row is one wave instruction: wave-uniform row address (scalar registers) + 16 bytes per lane -- the global_load_lds s[base]
    // form instead of a 64-bit pointer per lane (-DEGOTAP_TNF32_DMA_FLAT keeps the pointer form for the A/B)
    auto dma1s = [&](unsigned voff, const float* base, unsigned lds_addr) __attribute__((always_inline)) {
#ifdef EGOTAP_TNF32_DMA_FLAT
        dma1((const float*)((const char*)base + voff), lds_addr);
#else
        const unsigned long long sb = lds_dma_base(base);      // lds_dma.h
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
#endif
    };
    const float* ga = dY + (long)(m_lo + wid) * ldy + n0;
    const float* gb = X + (long)(m_lo + wid) * ldx + k0;
    const unsigned lo16 = lane * 16;
    auto dma_slab = [&](int s, int buf) __attribute__((always_inline)) {
        const unsigned sa = lds0 + (unsigned)(buf * SF + wid * LDA) * 4u, sb = lds0 + (unsigned)(buf * SF + BKM * LDA + wid * LDB) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dma1s(lo16, ga + ((long)s * BKM + 8 * q) * ldy, sa + (unsigned)(8 * q * LDA) * 4u);
            dma1s(lo16, gb + ((long)s * BKM + 8 * q) * ldx, sb + (unsigned)(8 * q * LDB) * 4u);
        }
    };

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // csum != nullptr: the workgroups of the first k-tile column also sum the dY rows they stage anyway -- csum[split][N] = column sums of dY over
    // the split's rows = the bias gradient of the same Linear layer, which used to be a pass of its own over dY (thread = column tid & 255, rows
    // 16 (tid >> 8) .. + 15 of every slab, fixed order)
    const bool do_cs = csum != nullptr && tk == 0;
    float cs = 0.f;
    if (nslab > 0) dma_slab(0, 0);
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's rows of slab s have landed ...
        __syncthreads();                                          // ... everybody's have, and nobody reads the other stage any more
        if (s + 1 < nslab) dma_slab(s + 1, buf ^ 1);
        if (do_cs) {
            const float* cp = smem_tnd + buf * SF + (tid >> 8) * 16 * LDA + (tid & 255);
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += cp[r * LDA];
        }
        const float* Ab = smem_tnd + buf * SF + lh * LDA + wn * (TN * 32) + l31;
        const float* Bb = smem_tnd + buf * SF + BKM * LDA + lh * LDB + wk * (TK * 32) + l31;
#pragma unroll
        for (int p = 0; p < BKM / 2; ++p) {
            float a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = Ab[2 * p * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = Bb[2 * p * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = slabs + (long)split * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int kk = k0 + wk * (TK * 32) + j * 32 + l31;
            const int nb = n0 + wn * (TN * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + (r & 3) + 8 * (r >> 2);
                out[(long)n * K + kk] = acc[i][j][r];
            }
        }
    if (do_cs) {                     // (uniform over the workgroup)
        __syncthreads();             // everybody is done with the stages: the first 2 KB become the hand-over of the two row halves
        if (tid >= 256) smem_tnd[tid & 255] = cs;
        __syncthreads();
        if (tid < 256) csum[(long)split * N + n0 + tid] = cs + smem_tnd[tid];
    }
}

static inline bool gemm_tn_f32_dma_ok(const float* dY, long ldy, const float* X, long ldx, int M, int N, int K) {
    return M % TnDmaCfg::BKM == 0 && N % TnDmaCfg::BN == 0 && K % TnDmaCfg::BK == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dY & 15) == 0 &&
           ((uintptr_t)X & 15) == 0;
}
// db != nullptr: also db[N] (+)= column sums of dY (the layer's bias gradient), summed by the workgroups that stage dY anyway
static hipError_t gemm_tn_f32_dma_launch(const float* dY, long ldy, const float* X, long ldx, float* dW, float* slabs, size_t slab_bytes, int M, int N,
                                         int K, int num_cu, int accumulate, hipStream_t stream, float* db = nullptr) {
    using Cfg = TnDmaCfg;
    if (!gemm_tn_f32_dma_ok(dY, ldy, X, ldx, M, N, K)) return hipErrorInvalidValue;
    const int tiles_n = N / Cfg::BN, tiles_k = K / Cfg::BK, tiles = tiles_n * tiles_k;
    // split count from the model shared with the convolution weight gradients: the q | k | v gradient (48 tiles) took 11 splits = 528 workgroups =
    // three rounds for 2.06 rounds of work under the "two blocks per CU" rule; 16 splits are three full rounds
    int per = 0;
    // with db every split needs N more floats (its column-sum partial row) behind its N x K slab
    const size_t avail = db ? slab_bytes / (size_t)(K + 1) * (size_t)K : slab_bytes;
    int splits = wgrad_pick_splits(tiles, M / Cfg::BKM, (long)N * K, avail, num_cu, Cfg::LDS_BYTES, Cfg::BKM / 2 * Cfg::TN * Cfg::TK * 64 / 2400.0, 6.0, &per);
    if (splits < 1) { splits = 1; per = M / Cfg::BKM; }
    const bool direct = splits == 1 && !accumulate;
    const size_t slab_floats = direct ? 0 : (size_t)splits * N * K;
    if ((slab_floats + (db ? (size_t)splits * N : 0)) * 4 > slab_bytes) return hipErrorOutOfMemory;
    const int rows_per = per * Cfg::BKM;
    float* csum = db ? slabs + slab_floats : nullptr;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_f32_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_f32_dma_kernel, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, dY, ldy, X, ldx, direct ? dW : slabs, M, N,
                       K, tiles_n, tiles_k, splits, rows_per, csum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (db) hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)csum, db, (long)N, splits, accumulate);
    if (direct) return hipGetLastError();
    const long n = (long)N * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, dW, n, splits, accumulate);
    return hipGetLastError();
}

static hipError_t colsum_f32_launch(const float* Y, long ldy, float* out, float* part, size_t part_bytes, int M, int N,
                                    int accumulate, hipStream_t stream) {
    if (N % 4 != 0) return hipErrorInvalidValue;
    int rows_per_block = 512;
    int gy = (M + rows_per_block - 1) / rows_per_block;
    while ((size_t)gy * N * 4 > part_bytes && rows_per_block < (1 << 30)) {
        rows_per_block *= 2;
        gy = (M + rows_per_block - 1) / rows_per_block;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N / 4 + 63) / 64, gy), dim3(256), 0, stream, Y, ldy, part, M, N, rows_per_block);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, stream, part, out, (long)N, gy,
                       accumulate);
    return hipGetLastError();
}
