// [r3] 3x3 stride-1 convolution 64 -> 64 channels + BatchNorm(eval) (+ residual) + ReLU on bf16 channels-last eye-interleaved maps:
// the four convolutions of ResNet-18's layer1 (torchvision BasicBlock x 2 via net_architecture.py:70) in the bf16 estimators.
//
// On the implicit-GEMM kernel (gemm_bf16s.h, 256-row tile, NI = 1) these ran at 8 % matrix-pipe utilisation: with N = 64 a 32-deep
// K-tile carries 8 MFMAs per wave but costs the same DMA + two-barrier cadence as one that carries 32, and every input pixel is
// fetched nine times (once per tap; eight of them from L2).  This is the direct form:
//   * a workgroup (4 waves, one per SIMD with the whole register file) owns 8 x 32 output pixels of one image; the 10 x 34 input halo (64 channels = 128 bytes per pixel) is
//     fetched ONCE by global_load_lds_dwordx4 (zero page outside the image), double buffered: the next tile's halo lands while this
//     tile multiplies.  All nine taps read it in place: a tap is an address offset.
//   * v_mfma_f32_32x32x16_bf16 with A = weights, B = pixels: wave (mh, rg) owns output channels 32 mh .. +31 and tile rows 4 rg .. 4 rg + 3.
//     Its A fragments -- 32 channels x 576 k -- stay in registers for the whole kernel (144 VGPRs, read once from the packed weights
//     [co][ci / 32][tap][ci % 32] of pack_conv3x3_bf16s_kernel); the only LDS traffic of the loop is one ds_read_b128 per MFMA.
//     A pixel's eight 16-byte channel chunks sit at chunk ^ ((x >> 1) & 7): the 16 lanes a ds_read_b128 serves together are 16
//     consecutive x of one row, i.e. 16 distinct 16-byte slots of the 256-byte bank row (conflict-free); the DMA applies the swizzle
//     on the global side.
//   * epilogue through a 64 KB fp32 patch in LDS ([pixel][64 channels], chunk ^ (x & 15)): acc * scale + shift per channel on the
//     way in; on the way out one thread takes 8 channels of a pixel -- 16-byte residual load, ReLU, one rounding to bf16, 16-byte
//     store: 8 lanes cover a pixel's 128 bytes.  Same fp32 formula as SEpiBnBf16 (conv_bf16s.h).
#pragma once
#include "conv_bf16s.h"

struct Conv64Cfg {
    static constexpr int TR = 8, TC = 32, HR = TR + 2, HC = TC + 2, HPIX = HR * HC;            // tile, halo (340 pixels)
    static constexpr int NDMA = (HPIX + 7) / 8;                                               // wave-instructions per halo: 43 (8 pixels each)
    static constexpr int HALO_BYTES = NDMA * 1024;                                            // 44 032
    static constexpr int STAGE_BYTES = TR * TC * 64 * 4;                                      // 65 536
    static constexpr int OFF_STAGE = 2 * HALO_BYTES, OFF_BN = OFF_STAGE + STAGE_BYTES;
    static constexpr int LDS_BYTES = OFF_BN + 2 * 64 * 4;
    static constexpr int THREADS = 256, WAVES = THREADS / 64, RPW = 2 * TR / WAVES;      // rows of the tile per wave (two waves share a row group: one per 32-channel half)
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(((HR - 1) * 128 + HC) * 128 + 64 < (1 << 20), "element offset of a halo pixel fits 20 bits at 128 x 128 maps");
};

// in / out / res: [B * S * S, 128] bf16 (pixel (b, y, x): [left 64 | right 64]); image n = 2 b + eye; wp: packed weights [64][2][9][32] bf16
static __global__ __launch_bounds__(Conv64Cfg::THREADS, 1) void conv64_direct_bf16s_kernel(
    const __bf16* __restrict__ in, const __bf16* __restrict__ zero, const __bf16* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const __bf16* __restrict__ res, __bf16* __restrict__ out, int log2S, int nimg, int relu) {
    using Cfg = Conv64Cfg;
    constexpr int TR = Cfg::TR, TC = Cfg::TC, HC = Cfg::HC, HPIX = Cfg::HPIX, NDMA = Cfg::NDMA, RPW = Cfg::RPW;
    extern __shared__ __attribute__((aligned(16))) char c64_sm[];
    char* stage = c64_sm + Cfg::OFF_STAGE;
    float* bn_sc = (float*)(c64_sm + Cfg::OFF_BN);
    float* bn_sh = bn_sc + 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), xl = lane & 31, h = lane >> 5;
    const int mh = wid & 1, rg = wid >> 1;
    constexpr int STORES = (TR * TC * 8) / Cfg::THREADS;       // output pieces (8 channels of a pixel) per thread and tile
    const int S = 1 << log2S, tx_n = S / TC, ty_n = S / TR;
    const long ntiles = (long)nimg * ty_n * tx_n;
    if (tid < 64) { bn_sc[tid] = scale[tid]; bn_sh[tid] = shift[tid]; }

    // ---- A fragments: channel 32 mh + xl; step (tap, sl, q) covers input channels 32 sl + 16 q + 8 h .. + 7 of tap
    bf16x8 af[9][2][2];
    {
        const __bf16* wrow = wp + (long)(mh * 32 + xl) * 576;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 2; ++q) af[tap][sl][q] = *(const bf16x8*)(wrow + (sl * 9 + tap) * 32 + 16 * q + 8 * h);
    }
    // ---- B fragment offsets inside a halo buffer, without the row term: column xl + dx, chunk 4 sl + 2 q + h
    int boff[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            const int hx = xl + dx;
            boff[dx][c2] = hx * 128 + (((2 * c2 + h) ^ ((hx >> 1) & 7)) << 4);
        }

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)c64_sm;
    auto tile_of = [&](long t, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        const int txi = (int)(t % tx_n), tyi = (int)((t / tx_n) % ty_n);
        n = (int)(t / ((long)tx_n * ty_n));
        y0 = tyi * TR;
        x0 = txi * TC;
    };
    // halo of tile t -> buffer b: wave w issues pieces w, w + 4, ...; lane i of piece j: pixel 8 j + (i >> 3), slot i & 7.  What does not
    // depend on the tile is kept per piece: the pixel's halo coordinates and its element offset from the halo's corner pixel.
    constexpr int PPW = (NDMA + Cfg::WAVES - 1) / Cfg::WAVES;       // pieces per wave: 11
    int dma_info[PPW];          // bits 0-19: element offset; 20-23: halo row; 24-29: halo column; 30: past the halo (the last piece's tail)
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
        const int p = 8 * (wid + jj * Cfg::WAVES) + (lane >> 3), pos = lane & 7;
        const int hy = p / HC, hx = p - hy * HC;
        dma_info[jj] = (((hy << log2S) + hx) * 128 + ((pos ^ ((hx >> 1) & 7)) << 3)) | (hy << 20) | (hx << 24) | (p < HPIX ? 0 : 1 << 30);
    }
    auto issue_halo = [&](long t, int b) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_of(t, n, y0, x0);
        const __bf16* corner = in + ((long)(n >> 1) << (2 * log2S)) * 128 + (n & 1) * 64 + ((((long)(y0 - 1)) << log2S) + (x0 - 1)) * 128;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int j = wid + jj * Cfg::WAVES;
            if (j < NDMA) {                                          // wave-uniform
                const int gy = y0 - 1 + ((dma_info[jj] >> 20) & 15), gx = x0 - 1 + ((dma_info[jj] >> 24) & 63);
                const bool ok = !(dma_info[jj] >> 30) && gy >= 0 && gy < S && gx >= 0 && gx < S;
                const __bf16* g = ok ? corner + (dma_info[jj] & 0xfffff) : zero;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds0 + b * Cfg::HALO_BYTES + j * 1024)) : "memory");
            }
        }
    };

    if ((long)blockIdx.x < ntiles) issue_halo(blockIdx.x, 0);
    int buf = 0;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int n, y0, x0;
        tile_of(t, n, y0, x0);
        // this wave's pieces of the tile's halo have landed: vmcnt is one in-order counter and the only younger operations are the
        // previous tile's stores (exactly STORES per thread, every lane active), which may stay in flight
        if (t == (long)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES) : "memory");
        __syncthreads();                                       // ... everyone's; the other buffer and the patch are free again
        if (t + gridDim.x < ntiles) issue_halo(t + gridDim.x, buf ^ 1);

        // this thread's output pieces: element offset of piece k = obase + (row_k * S + x_k) * 128; the residual (same layout as the output)
        // is requested now and used after the MFMAs
        const long obase = ((long)(n >> 1) << (2 * log2S)) * 128 + (n & 1) * 64 + (((long)y0 << log2S) + x0) * 128 + 8 * (tid & 7);
        bf16x8 rres[STORES];
        if (res != nullptr) {
#pragma unroll
            for (int k = 0; k < STORES; ++k) {
                const int tp = (tid + k * Cfg::THREADS) >> 3;
                rres[k] = *(const bf16x8*)(res + obase + (((long)(tp >> 5) << log2S) + (tp & (TC - 1))) * 128);
            }
        }
        f32x16 acc[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const char* hb = c64_sm + buf * Cfg::HALO_BYTES;
        // 36 k-steps (tap, c2) in 18 groups of two: the fragment reads of group g + 1 are issued before the MFMAs of group g (one wave per
        // SIMD: nobody else hides the LDS latency); the scheduling barrier per group keeps the compiler from hoisting the whole tile's reads
        // above the first MFMA (it then spills the weights)
        auto frag = [&](int step, int i) __attribute__((always_inline)) {
            const int tap = step >> 2, c2 = step & 3;
            return *(const bf16x8*)(hb + (RPW * rg + tap / 3 + i) * (HC * 128) + boff[tap % 3][c2]);
        };
        bf16x8 bq[2][2][RPW];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < RPW; ++i) bq[0][u][i] = frag(u, i);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (g + 1 < 18) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < RPW; ++i) bq[(g + 1) & 1][u][i] = frag(2 * (g + 1) + u, i);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int step = 2 * g + u, tap = step >> 2, c2 = step & 3;
#pragma unroll
                for (int i = 0; i < RPW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap][c2 >> 1][c2 & 1], bq[g & 1][u][i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- accumulators -> fp32 patch [pixel][64 channels], BatchNorm applied; 16-byte chunk ch of pixel (row, x) at ch ^ (x & 15)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            char* prow = stage + (((RPW * rg + i) * TC + xl) << 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co0 = mh * 32 + 8 * g + 4 * h;
                const f32x4 sc = *(const f32x4*)(bn_sc + co0), sh = *(const f32x4*)(bn_sh + co0);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e] * sc[e] + sh[e];
                *(f32x4*)(prow + (((co0 >> 2) ^ (xl & 15)) << 4)) = v;
            }
        }
        __syncthreads();
        // ---- patch -> global: item = (pixel, 8 channels); 8 consecutive lanes cover the 128 bytes of a pixel's eye slice
#pragma unroll
        for (int k = 0; k < STORES; ++k) {
            const int item = tid + k * Cfg::THREADS;
            const int u = item & 7, tp = item >> 3, x = tp & (TC - 1), row = tp >> 5;
            const char* pp = stage + (tp << 8);
            const f32x4 v0 = *(const f32x4*)(pp + (((2 * u) ^ (x & 15)) << 4)), v1 = *(const f32x4*)(pp + (((2 * u + 1) ^ (x & 15)) << 4));
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (res != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rres[k][e];
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *(bf16x8*)(out + obase + (((long)row << log2S) + x) * 128) = o;
        }
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static inline hipError_t conv64_direct_bf16s_launch(const __bf16* in, const __bf16* zero, const __bf16* wp, const float* scale, const float* shift,
                                                    const __bf16* res, __bf16* out, int log2S, int nimg, int relu, int num_cu, hipStream_t s) {
    using Cfg = Conv64Cfg;
    const int S = 1 << log2S;
    if (S % Cfg::TC != 0 || S % Cfg::TR != 0 || nimg <= 0) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv64_direct_bf16s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const long ntiles = (long)nimg * (S / Cfg::TR) * (S / Cfg::TC);
    const long grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(conv64_direct_bf16s_kernel, dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, in, zero, wp, scale, shift, res, out, log2S,
                       nimg, relu);
    return hipGetLastError();
}
