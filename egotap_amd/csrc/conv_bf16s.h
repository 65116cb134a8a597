// The U-Net decoder of HeatMap_UnrealEgo_Shared (model/net_architecture.py:139-173: layerK_1x1, upsample, concat, conv_up3/2/1,
// conv_heatmap; model/network_utils.py:144-148 convrelu) with bf16 ACTIVATIONS in HBM, channels-last -- the EGOTAP_PREC_BF16 mode of
// egotap_hm_forward (the reference's --use_amp autocast region, egotap_autoencoder_model.py:219).  82 % of an estimator's FLOPs are
// the three 3x3 decoder convolutions; on fp32 NCHW tensors (conv_bf16.h, one bf16 product) they were bound by the staging -- fp32
// reads, a VALU conversion and a scattered LDS write per element, 12 MFMAs per barrier -- not by the matrix pipe.
//
// Channels-last turns every convolution here into the bf16-storage GEMM of gemm_bf16s.h (256 x 256 tile, LDS-DMA ring, two wave
// groups one barrier apart) with nothing but a different X-operand loader:
//   rows    m = (image, y, x) of the output map, S x S pixels per image (S a power of two); columns n = output channel
//   3x3:    k = (ci / 32, tap, ci % 32), Cp = Cin padded to a multiple of 32: a 32-deep K-tile is one tap of one 32-channel slab and
//           row m of it is the 64 contiguous bytes of input pixel (y + dy, x + dx), channels 32 s..32 s + 31 -- one DMA per lane,
//           from a page of zeros where the tap leaves the image (the padding of F.conv2d).  Never im2col'ed.  The nine taps of a
//           slab are consecutive K-tiles, so eight of the nine reads of a pixel's 64 bytes hit L2 (tap-major order re-read the
//           whole input from the fabric per tap: 7.5 GB per launch of conv_up1 against 1.3 GB of input).
//   1x1:    the plain loader (rows are pixels).
// Weights are repacked per call into [Cout][tap][Cp] bf16 (the parameters stay the caller's live fp32 tensors).  Producers write
// straight into channel slices of the next concat buffer (row stride = the concat's channel count): no concat pass.
#pragma once
#include "gemm_bf16s.h"

// ---------------------------------------------------------------------------------------------------- loader
struct XConv3 {
    const __bf16* in;        // [Nimg * S * S, Cp] bf16, channels-last
    const __bf16* zero;      // 64 bytes of zeros
    int Cp, log2S;           // the slab / tap split of a K-tile index is a multiply by 7282 >> 16 (kt / 9 exactly for kt < 7000)
    struct Row { const __bf16* p; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int S = 1 << log2S, x = m & (S - 1), y = (m >> log2S) & (S - 1);
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (y + dy >= 0 && y + dy < S && x + dx >= 0 && x + dx < S) mask |= 1u << t;
        }
        return Row{in + (long)m * Cp, mask};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const {      // k0 wave-uniform: the tap arithmetic is scalar
        const int kt = k0 >> 5, slab = (kt * 7282) >> 16, tap = kt - 9 * slab;      // kt / 9 exactly for kt < 7000
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        const int off = ((dy << log2S) + dx) * Cp + 32 * slab;
        return ((r.mask >> tap) & 1u) ? r.p + off + ko : zero + ko;
    }
};

// ---------------------------------------------------------------------------------------------------- epilogues
template <bool GUARD>
struct SEpiConvBf16 {        // out bf16 = relu?(acc + bias); GUARD: the GEMM's N is padded, only columns < n_lim (a multiple of 8) exist
    static constexpr int W = 8, STORES = 1;
    const float* bias;       // N (padded) values
    __bf16* out;
    long ld;
    int n_lim, relu;
    typedef SBias8 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return load_bias8(bias, n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] += c.b0[i]; v[4 + i] += c.b1[i]; }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (!GUARD || n < n_lim) store_bf16x8(out + (long)m * ld + n, v);
    }
};
template <> struct s_epi_exact<SEpiConvBf16<true>> { static constexpr bool value = false; };    // its stores may be skipped: no store slack in the main loop

struct SEpiHeatNCHW {        // conv_heatmap: out f32 NCHW = acc + bias at out + image * img_stride + channel * HW + pixel
    static constexpr int W = 4, STORES = 4;
    const float* bias;       // N (padded) values
    float* out;
    long img_stride;
    int n_out, log2hw;
    typedef f32x4 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return *(const f32x4*)(bias + n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
        const int img = m >> log2hw, pix = m & ((1 << log2hw) - 1);
        float* o = out + (long)img * img_stride + ((long)n << log2hw) + pix;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < n_out) o[(long)i << log2hw] = v[i] + c[i];
    }
};

template <> struct s_epi_exact<SEpiHeatNCHW> { static constexpr bool value = false; };

// ---------------------------------------------------------------------------------------------------- helpers
// [Cout][Cin][3][3] f32 -> [Np][Cp / 32][tap][32] bf16 (channels past Cin and rows past Cout zero).  One thread = 8 consecutive ci of one (co, tap).
static __global__ __launch_bounds__(256) void pack_conv3x3_bf16s_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, int Cout, int Cin, int Cp,
                                                                        int Np) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8n = Cp / 8;
    if (i >= (long)Np * 9 * c8n) return;
    const int c8 = (int)(i % c8n), tap = (int)((i / c8n) % 9), co = (int)(i / (9L * c8n));
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = c8 * 8 + j;
        o[j] = (ci < Cin && co < Cout) ? (__bf16)w[((long)co * Cin + ci) * 9 + tap] : (__bf16)0.f;
    }
    *(bf16x8*)(wb + (long)co * 9 * Cp + ((long)(c8 >> 2) * 9 + tap) * 32 + (c8 & 3) * 8) = o;       // k = (ci / 32, tap, ci % 32)
}
// [Cout][Cin] f32 (+ bias) -> [Np][Cin] bf16, [Np] f32: rows past Cout zero
static __global__ __launch_bounds__(256) void pack_conv1x1_bf16s_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                                        __bf16* __restrict__ wb, float* __restrict__ bp, int Cout, int Cin, int Np) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8n = Cin / 8;
    if (i >= (long)Np * c8n) return;
    const int c8 = (int)(i % c8n), co = (int)(i / c8n);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = co < Cout ? (__bf16)w[(long)co * Cin + c8 * 8 + j] : (__bf16)0.f;
    *(bf16x8*)(wb + (long)co * Cin + c8 * 8) = o;
    if (c8 == 0 && b != nullptr) bp[co] = co < Cout ? b[co] : 0.f;
}
// pyramid level: f32 NCHW [Nimg, C, HW] -> bf16 channels-last [Nimg * HW, C].  Lanes along pixels (coalesced reads), 8 channels each.
static __global__ __launch_bounds__(256) void nchw_to_nhwc_bf16s_kernel(const float* __restrict__ in, __bf16* __restrict__ out, int C, int HW, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pix = (int)(i % HW);
    const long t = i / HW;
    const int c8 = (int)(t % (C / 8));
    const long img = t / (C / 8);
    const float* p = in + (img * C + c8 * 8) * (long)HW + pix;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)p[(long)j * HW];
    *(bf16x8*)(out + (img * HW + pix) * (long)C + c8 * 8) = o;
}
// F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) on channels-last bf16: in [Nimg, h, h, C] -> the first C channels
// of out rows [Nimg, 2h, 2h, ld].  One thread = 8 channels of one output pixel; fp32 arithmetic in the order of upsample2x_kernel.
static __global__ __launch_bounds__(256) void upsample2x_nhwc_bf16s_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int C, int h, long ld,
                                                                          long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c8n = C / 8, HO = 2 * h;
    const int c8 = (int)(i % c8n);
    const long pp = i / c8n;
    const int x = (int)(pp % HO), y = (int)((pp / HO) % HO);
    const long img = pp / ((long)HO * HO);
    const float scale = (float)(h - 1) / (float)(HO - 1);
    const float sy = scale * y, sx = scale * x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < h - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly, lx = sx - x0, hx = 1.f - lx;
    const __bf16* b = in + img * (long)h * h * C + c8 * 8;
    const bf16x8 v00 = *(const bf16x8*)(b + ((long)y0 * h + x0) * C), v01 = *(const bf16x8*)(b + ((long)y0 * h + x1) * C);
    const bf16x8 v10 = *(const bf16x8*)(b + ((long)y1 * h + x0) * C), v11 = *(const bf16x8*)(b + ((long)y1 * h + x1) * C);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (__bf16)(hy * (hx * (float)v00[j] + lx * (float)v01[j]) + ly * (hx * (float)v10[j] + lx * (float)v11[j]));
    *(bf16x8*)(out + pp * ld + c8 * 8) = o;
}

// ---------------------------------------------------------------------------------------------------- ResNet-18 backbone, same scheme
// The backbone runs on 2B images (n = 2 b + eye); the decoder wants the two eyes' pyramids concatenated along the channels
// (net_architecture.py:139-147).  Every backbone tensor is therefore stored as [B * S * S, 2 C] bf16 -- pixel (b, y, x) holds
// [left C | right C] -- so a pyramid level IS the decoder's operand (no concat, no conversion) and a conv of image n reads / writes
// the C-channel slice eye * C of rows b * S * S + pixel.  BasicBlock convs (3x3 stride 1 / 2, 1x1 stride-2 downsample) with
// BatchNorm (eval, folded to scale / shift per call), residual and ReLU in the epilogue.  Cout = 64 / 128 run as N = 256 (column guard).
struct XConvE {
    const __bf16* in;        // [B * Si * Si, 2 C] bf16, Si = So * stride
    const __bf16* zero;
    int C, log2So, stride, taps;
    struct Row { const __bf16* p; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int So = 1 << log2So, xo = m & (So - 1), yo = (m >> log2So) & (So - 1), n = m >> (2 * log2So);
        const int Si = So * stride, yi = yo * stride, xi = xo * stride;
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (yi + dy >= 0 && yi + dy < Si && xi + dx >= 0 && xi + dx < Si) mask |= 1u << t;
        }
        return Row{in + ((long)(n >> 1) * Si * Si + (long)yi * Si + xi) * (2 * C) + (n & 1) * C, mask};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const {
        if (taps == 1) return r.p + k0 + ko;                   // 1x1 (stride-2 downsample): the centre pixel, always inside
        const int kt = k0 >> 5, slab = (kt * 7282) >> 16, tap = kt - 9 * slab;
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        const int Si = stride << log2So;
        const int off = (dy * Si + dx) * (2 * C) + 32 * slab;
        return ((r.mask >> tap) & 1u) ? r.p + off + ko : zero + ko;
    }
};

struct SBn8 { f32x4 s0, s1, h0, h1; };
__device__ __forceinline__ void s_keep(const SBn8& b) { asm volatile("" ::"v"(b.s0), "v"(b.s1), "v"(b.h0), "v"(b.h1)); }
template <bool GUARD>
struct SEpiBnBf16 {          // out bf16 = relu?(acc * scale + shift (+ R)) in the eye-interleaved layout; GUARD: only columns < C exist
    static constexpr int W = 8, STORES = 1;
    const float* scale;      // N (padded) values each
    const float* shift;
    const __bf16* R;         // residual (same layout as out) or null
    __bf16* out;
    int C, log2So, relu;
    typedef SBn8 Col;
    typedef bf16x8 Aux;
    __device__ __forceinline__ long addr(int m) const {
        const int n = m >> (2 * log2So), pix = m & ((1 << (2 * log2So)) - 1);
        return (((long)(n >> 1) << (2 * log2So)) + pix) * (2 * C) + (n & 1) * C;
    }
    __device__ __forceinline__ Col col(int n) const {
        return Col{*(const f32x4*)(scale + n), *(const f32x4*)(scale + n + 4), *(const f32x4*)(shift + n), *(const f32x4*)(shift + n + 4)};
    }
    __device__ __forceinline__ Aux fetch(int m, int n) const {
        if (R == nullptr) return bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        return *(const bf16x8*)(R + addr(m) + (GUARD ? min(n, C - 8) : n));
    }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux& r, int m, int n) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = v[i] * c.s0[i] + c.h0[i] + (float)r[i];
            v[4 + i] = v[4 + i] * c.s1[i] + c.h1[i] + (float)r[4 + i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (!GUARD || n < C) store_bf16x8(out + addr(m) + n, v);
    }
};
template <> struct s_epi_exact<SEpiBnBf16<true>> { static constexpr bool value = false; };

// eval-mode BatchNorm folded to y = x * scale + shift, padded with zeros to Np channels (as conv_f32.h's epilogue computes it)
static __global__ __launch_bounds__(256) void bn_fold_bf16s_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ m,
                                                                   const float* __restrict__ v, float* __restrict__ scale, float* __restrict__ shift,
                                                                   int C, int Np) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Np) return;
    float sc = 0.f, sh = 0.f;
    if (c < C) {
        sc = g[c] / sqrtf(v[c] + 1e-5f);
        sh = b[c] - m[c] * sc;
    }
    scale[c] = sc;
    shift[c] = sh;
}

// ---------------------------------------------------------------------------------------------------- [r3] one launch for all of them
// An estimator forward repacked its 27 convolution weights and folded its 19 BatchNorms with 46 tiny launches (0.46 ms of a 15 ms
// forward at B = 256, 5 % at B = 32 and 512 x 512).  The parameters stay the caller's live fp32 tensors and nothing is cached across
// calls -- the whole set is simply converted by ONE launch at the start of the forward, into a region of the workspace that holds
// every layer's packed copy.  The segment table travels as a kernel argument (2.6 KB): no device-side table to allocate or upload.
struct PackSeg {
    const float *w, *b;              // [Cout][Cin][taps] weights, bias (1x1 with bias) or null
    unsigned dst_w, dst_b;           // byte offsets into the region: packed bf16 weights, padded fp32 bias
    unsigned short Cout, Cin, Cp, Np;
    short taps, slab;                // slab: input channels per K slab of a 3x3 weight: 32 (gemm_bf16s.h's XConv3 / XConvE) or 64 (gemm_bf16s64.h's X64Conv3)
    int first_block;
};
struct BnSeg {
    const float *g, *b, *m, *v;
    unsigned dst_sc, dst_sh;
    unsigned short C, Np;
    int first_block;
};
struct PackTable {
    static constexpr int MAXW = 44, MAXB = 36;        // resnet34: 32 + 3 + 8 convolutions, 35 BatchNorms
    PackSeg w[MAXW];
    BnSeg bn[MAXB];
    int nw, nb, blocks_w, blocks;
};
static_assert(sizeof(PackTable) <= 4000, "the table is a kernel argument");
static __global__ __launch_bounds__(256) void pack_all_bf16s_kernel(PackTable T, char* __restrict__ region) {
    const int blk = blockIdx.x;
    if (blk < T.blocks_w) {
        int si = 0;
        while (si + 1 < T.nw && T.w[si + 1].first_block <= blk) ++si;
        const PackSeg& sg = T.w[si];
        const long i = (long)(blk - sg.first_block) * 256 + threadIdx.x;
        __bf16* wb = (__bf16*)(region + sg.dst_w);
        if (sg.taps == 9) {
            // one thread = 8 input channels of one output channel, ALL nine taps: its 72 source floats are contiguous (w[co][ci][tap],
            // ci = 8 c8 ..), so a wave reads 64 x 288 consecutive bytes; per tap the 8 channels go out as one 16-byte piece.  (One
            // thread per (co, tap, c8) read the same lines nine times at a 36-byte stride: 262 us per estimator against 60.)
            const int c8n = sg.Cp / 8;
            if (i >= (long)sg.Np * c8n) return;
            const int c8 = (int)(i % c8n), co = (int)(i / c8n);
            float v[72];
            const bool row = co < sg.Cout;
            if (row && c8 * 8 + 8 <= sg.Cin && ((size_t)sg.w & 15) == 0) {                       // (a parameter bound from a 4-byte-aligned view takes the scalar branch)
                const f32x4* src = (const f32x4*)(sg.w + ((long)co * sg.Cin + c8 * 8) * 9);      // 288-byte pieces: 16-byte aligned
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    const f32x4 t = src[q];
                    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 72; ++q) {
                    const int ci = c8 * 8 + q / 9;
                    v[q] = (row && ci < sg.Cin) ? sg.w[((long)co * sg.Cin + ci) * 9 + q % 9] : 0.f;
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j * 9 + tap];
                // k = (ci / slab, tap, ci % slab)
                if (sg.slab == 64) *(bf16x8*)(wb + (long)co * 9 * sg.Cp + ((long)(c8 >> 3) * 9 + tap) * 64 + (c8 & 7) * 8) = o;
                else *(bf16x8*)(wb + (long)co * 9 * sg.Cp + ((long)(c8 >> 2) * 9 + tap) * 32 + (c8 & 3) * 8) = o;
            }
        } else {
            const int c8n = sg.Cin / 8;
            if (i >= (long)sg.Np * c8n) return;
            const int c8 = (int)(i % c8n), co = (int)(i / c8n);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = co < sg.Cout ? (__bf16)sg.w[(long)co * sg.Cin + c8 * 8 + j] : (__bf16)0.f;
            *(bf16x8*)(wb + (long)co * sg.Cin + c8 * 8) = o;
            if (c8 == 0 && sg.b != nullptr) ((float*)(region + sg.dst_b))[co] = co < sg.Cout ? sg.b[co] : 0.f;
        }
        return;
    }
    int si = 0;
    while (si + 1 < T.nb && T.bn[si + 1].first_block <= blk) ++si;
    const BnSeg& sg = T.bn[si];
    const int c = (blk - sg.first_block) * 256 + threadIdx.x;
    if (c >= sg.Np) return;
    float sc = 0.f, sh = 0.f;
    if (c < sg.C) {
        sc = sg.g[c] / sqrtf(sg.v[c] + 1e-5f);
        sh = sg.b[c] - sg.m[c] * sc;
    }
    ((float*)(region + sg.dst_sc))[c] = sc;
    ((float*)(region + sg.dst_sh))[c] = sh;
}
