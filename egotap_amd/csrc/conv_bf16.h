// 3x3 stride-1 NCHW convolution on the bf16 matrix cores with fp32 tensors in HBM (opt-in modes of egotap_set_precision):
// the U-Net decoder convs and the stride-1 BasicBlock convs of HeatMap_UnrealEgo_Shared (model/net_architecture.py:53-173,
// model/network_utils.py:144-148) at map widths 64 / 32 / 16 / 8 -- 97 % of the estimators' FLOPs.  Arithmetic as gemm_bf16.h:
// NP = 3 takes every product as hi*hi + hi*lo + lo*hi of operands split into hi + lo bf16, NP = 1 rounds to bf16.
//
//   D[co][pixel] = sum_{tap, ci} W[co][ci][tap] * X[ci][pixel + tap]        (implicit GEMM, never im2col'ed)
// A v_mfma_f32_32x32x16_bf16 fragment is 8 consecutive k per lane, so the contraction order is (tap, ci) with ci innermost:
// * input: the block stages the raw rows of its 256-pixel tile (+ halo) for a slab of 16 channels ONCE, transposed to
//   channels-last in LDS -- [row][x][hi(16 ci) | lo(16 ci) | pad] = 80 bytes per pixel -- and every tap of every output
//   pixel is one ds_read_b128 per image at (pixel + tap) * 80 + half * 16: the halo re-reads stay inside LDS.
// * weights: PyTorch's [Cout][Cin][3][3] has the tap innermost; conv_pack_w_kernel rewrites them per launch (the tensors stay
//   the caller's live parameters; 57 MB for the largest conv = ~25 us) into [co][ci/16][tap][hi 16 | lo 16], so a (slab, ky)
//   sub-slab is 192 contiguous bytes per output channel and goes global -> LDS without touching the VALU.
// * tile 128 (or 64) co x 256 pixels = whole rows of one image (whole 8x8 images at width 8), 8 waves, 64 co each; per 16-channel slab the input image is staged once and reused
//   by three weight sub-slabs (ky = 0, 1, 2), both double buffered; one barrier per sub-slab (36 MFMAs per wave at NP = 3).
// * epilogue = conv_f32.h's: BatchNorm(eval) or bias, residual, ReLU, 128-byte NCHW row segments with a caller-given image
//   stride (concat-free).
#pragma once
#include "conv_f32.h"
#include "gemm_bf16.h"

// W[Cout][Cin][9] fp32 -> Wp[Cout][S][9][hi 16 | lo 16] bf16, S = ceil(Cin / 16); channels past Cin are zero
static __global__ __launch_bounds__(256) void conv_pack_w_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Cout, int Cin,
                                                          int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * S * 9) return;
    const int tap = (int)(i % 9);
    const long cs = i / 9;
    const int s = (int)(cs % S), co = (int)(cs / S);
    const float* src = w + ((long)co * Cin + 16 * s) * 9 + tap;
    bf16x8 hi[2], lo[2];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float v = 16 * s + c < Cin ? src[c * 9] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[c >> 3][c & 7] = h;
        lo[c >> 3][c & 7] = (__bf16)(v - (float)h);
    }
    bf16x8* dst = (bf16x8*)(wp + i * 32);
    dst[0] = hi[0]; dst[1] = hi[1]; dst[2] = lo[0]; dst[3] = lo[1];
}

template <int LOG2W_, int NP_, int CO_T_ = 128>
struct ConvBfCfg {
    static constexpr int LOG2W = LOG2W_, NP = NP_, W = 1 << LOG2W_, CO_T = CO_T_;
    static constexpr int PX_T = 256, R = PX_T / W;                         // output rows per tile
    static constexpr int G = R > W ? R / W : 1, RSEG = R > W ? W : R;      // whole images per tile when the map is small (W = 8)
    static constexpr int WCO = CO_T / 64, WPX = 8 / WCO, THREADS = 512;    // every wave owns 64 output channels
    static constexpr int TCO = 2, TPX = PX_T / WPX / 32;
    static constexpr int XROW = W + 2, XROWS = RSEG + 2, PXE = 40;         // staged pixels per row / rows per image; bf16 per pixel (80 B)
    static constexpr int XBUF = G * XROWS * XROW * PXE;                    // bf16 per input buffer
    static constexpr int WROW = 104, WBUF = CO_T * WROW;                   // 208-byte weight rows (3 taps x 64 B + pad)
    static constexpr int LDS_BYTES = 2 * (XBUF + WBUF) * 2;
    static constexpr int NQ = 8 * G * XROWS * (W / 4);                     // (channel pair, image, row, float4 column) staging items
    static constexpr int X_IT = (NQ + THREADS - 1) / THREADS;
    static constexpr int W_IT = (CO_T * 12 + THREADS - 1) / THREADS;       // 16-byte chunks of a weight sub-slab per thread
    static_assert(CO_T == 64 || CO_T == 128, "output-channel tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void conv_bf16_kernel(ConvArgs a, const __bf16* __restrict__ wp, int S) {
    constexpr int W = Cfg::W, LOG2W = Cfg::LOG2W, R = Cfg::R, XROW = Cfg::XROW, XROWS = Cfg::XROWS, PXE = Cfg::PXE;
    constexpr int G = Cfg::G, RSEG = Cfg::RSEG;
    constexpr int XBUF = Cfg::XBUF, WROW = Cfg::WROW, WBUF = Cfg::WBUF, THREADS = Cfg::THREADS, NP = Cfg::NP;
    constexpr int TCO = Cfg::TCO, TPX = Cfg::TPX, CO_T = Cfg::CO_T, X_IT = Cfg::X_IT, W_IT = Cfg::W_IT, NQ = Cfg::NQ;
    extern __shared__ __attribute__((aligned(16))) __bf16 csm[];
    __bf16* Xs = csm;                    // [2][XBUF]
    __bf16* Ws = csm + 2 * XBUF;         // [2][WBUF]

    int tpx, tco;
    xcd_tile(blockIdx.x, gridDim.x, a.tiles_px, a.tiles_co, 32, tpx, tco);
    const int co0 = tco * CO_T;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wco = wid / Cfg::WPX, wpx = wid % Cfg::WPX;
    const int l31 = lane & 31, lh = lane >> 5;
    int n0, y0;               // tile origin: image n0 (segment g -> image n0 + g), first output row y0
    if (G == 1) {
        const int gr0 = tpx * R;
        n0 = gr0 >> LOG2W;
        y0 = gr0 - (n0 << LOG2W);
    } else {
        n0 = tpx * G;
        y0 = 0;
    }
    const long ch_in = (long)W * W;

    // zero both input buffers once: x halos and rows outside the image are never written again
    for (int i = tid; i < 2 * XBUF / 8; i += THREADS) ((bf16x8*)Xs)[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

    f32x4 wreg[W_IT];
    f32x4 xreg[X_IT][2];
    auto wload = [&](int s, int ky) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int c = min(tid + it * THREADS, CO_T * 12 - 1), cr = c / 12, chk = c - cr * 12;
            const int co = min(co0 + cr, a.Cout - 1);
            wreg[it] = *(const f32x4*)(wp + (((long)co * S + s) * 9 + ky * 3) * 32 + chk * 8);
        }
    };
    auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int c = tid + it * THREADS, cr = c / 12, chk = c - cr * 12;
            if (c < CO_T * 12) *(f32x4*)(Ws + buf * WBUF + cr * WROW + chk * 8) = wreg[it];
        }
    };
    auto xload = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int q = tid + it * THREADS;
            const int cp = q & 7, x4 = (q >> 3) % (W / 4), gr = (q >> 3) / (W / 4), g = gr / XROWS, rr = gr - g * XROWS;
            const int y = y0 - 1 + rr, ci = 16 * s + 2 * cp, n = n0 + g;
            const bool ok = q < NQ && y >= 0 && y < W && n < a.Nimg;
            const float* p = a.in + (long)n * a.in_istride + (long)ci * ch_in + (long)y * W + x4 * 4;
            xreg[it][0] = ok && ci < a.Cin ? *(const f32x4*)p : f32x4{0.f, 0.f, 0.f, 0.f};
            xreg[it][1] = ok && ci + 1 < a.Cin ? *(const f32x4*)(p + ch_in) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto xstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int q = tid + it * THREADS;
            const int cp = q & 7, x4 = (q >> 3) % (W / 4), gr = (q >> 3) / (W / 4), g = gr / XROWS, rr = gr - g * XROWS;
            const int y = y0 - 1 + rr;
            if (q < NQ && y >= 0 && y < W) {            // images past Nimg are staged as zeros (xload)
                __bf16* dst = Xs + buf * XBUF + (gr * XROW + 1 + 4 * x4) * PXE + 2 * cp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 h, l;
                    h[0] = (__bf16)xreg[it][0][e];
                    h[1] = (__bf16)xreg[it][1][e];
                    l[0] = (__bf16)(xreg[it][0][e] - (float)h[0]);
                    l[1] = (__bf16)(xreg[it][1][e] - (float)h[1]);
                    *(bf16x2*)(dst + e * PXE) = h;
                    if (NP == 3) *(bf16x2*)(dst + e * PXE + 16) = l;
                }
            }
        }
    };

    int pixb[TPX];            // lane's pixel inside the staged slab (row yy, column x of the halo-less tile), in pixels
#pragma unroll
    for (int j = 0; j < TPX; ++j) {
        const int p = (wpx * TPX + j) * 32 + l31;
        const int g = p / (RSEG * W), rem = p - g * (RSEG * W);
        pixb[j] = (g * XROWS + (rem >> LOG2W)) * XROW + (rem & (W - 1));
    }
    const int a_off = (wco * TCO * 32 + l31) * WROW + 8 * lh;

    f32x16 acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int wbuf, int xbuf, int ky) __attribute__((always_inline)) {
        const __bf16* Wb = Ws + wbuf * WBUF + a_off;
        const __bf16* Xb = Xs + xbuf * XBUF + (ky * XROW) * PXE + 8 * lh;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 ah[TCO], al[TCO], bh[TPX], bl[TPX];
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                ah[i] = *(const bf16x8*)(Wb + i * 32 * WROW + kx * 32);
                if (NP == 3) al[i] = *(const bf16x8*)(Wb + i * 32 * WROW + kx * 32 + 16);
            }
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                bh[j] = *(const bf16x8*)(Xb + (pixb[j] + kx) * PXE);
                if (NP == 3) bl[j] = *(const bf16x8*)(Xb + (pixb[j] + kx) * PXE + 16);
            }
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    if (NP == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    // prologue: sub-slab (0, ky 0) and input slab 0 staged.  Loads past the end are clamped to the last slab (valid memory,
    // written to a buffer nobody reads), so the loop body has no load-guarding branches.
    wload(0, 0);
    xload(0);
    __syncthreads();          // zero fill done before the first staged rows land
    wstore(0);
    xstore(0);
    __syncthreads();
    int wbuf = 0;
    for (int s = 0; s < S; ++s) {
        const int xbuf = s & 1, sn = min(s + 1, S - 1);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (ky < 2) wload(s, ky + 1); else wload(sn, 0);
            if (ky == 0) xload(sn);
            compute(wbuf, xbuf, ky);
            wstore(wbuf ^ 1);
            if (ky == 2) xstore(xbuf ^ 1);
            __syncthreads();
            wbuf ^= 1;
        }
    }

    // epilogue: accumulator register r of lane l = D[co = 32x32 row (r&3) + 8*(r>>2) + 4*(l>>5)][pixel l&31]
    const long ch_out = (long)W * W;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co >= a.Cout) continue;
            float sc = 1.f, sh;
            if (a.gamma) {
                sc = a.gamma[co] / sqrtf(a.var[co] + 1e-5f);
                sh = a.beta[co] - a.mean[co] * sc;
            } else {
                sh = a.bias[co];
            }
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int p = (wpx * TPX + j) * 32 + l31;
                const int g = p / (RSEG * W), rem = p - g * (RSEG * W);
                const int n = n0 + g;
                if (n >= a.Nimg) continue;
                const long pix = (long)y0 * W + rem;
                float v = acc[i][j][r] * sc + sh;
                if (a.res) v += a.res[(long)n * a.res_istride + co * ch_out + pix];
                if (a.relu) v = fmaxf(v, 0.f);
                a.out[(long)n * a.out_istride + co * ch_out + pix] = v;
            }
        }
    }
}

// wp: device scratch of at least conv_bf16_pack_bytes(Cout, Cin) bytes
static inline size_t conv_bf16_pack_bytes(int Cout, int Cin) { return (size_t)Cout * ((Cin + 15) / 16) * 9 * 64; }

template <class Cfg>
static hipError_t conv_bf16_launch(ConvArgs a, __bf16* wp, hipStream_t stream) {
    if (a.Nimg <= 0) return hipSuccess;
    const int S = (a.Cin + 15) / 16;
    const long items = (long)a.Cout * S * 9;
    hipLaunchKernelGGL(conv_pack_w_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a.w, wp, a.Cout, a.Cin, S);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = conv_bf16_kernel<Cfg>;
    static bool attr_done = false;
    if (!attr_done) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    a.tiles_co = (a.Cout + Cfg::CO_T - 1) / Cfg::CO_T;      // a ragged last tile re-reads the last channel's weights and skips its stores
    a.tiles_px = Cfg::G == 1 ? (int)((long)a.Nimg * Cfg::W * Cfg::W / Cfg::PX_T) : (a.Nimg + Cfg::G - 1) / Cfg::G;
    hipLaunchKernelGGL(kern, dim3(a.tiles_co * a.tiles_px), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, (const __bf16*)wp, S);
    return hipGetLastError();
}
