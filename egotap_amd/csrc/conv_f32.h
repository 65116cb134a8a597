// NCHW fp32 convolution (3x3 pad 1 / 1x1, stride 1 or 2) as an implicit GEMM on the gfx950 matrix cores, with
// the BatchNorm(eval) / bias, residual add and ReLU of the reference fused into the epilogue:
//   HeatMap_UnrealEgo_Shared: torchvision BasicBlock convs + the U-Net decoder's convrelu blocks
//   (model/net_architecture.py:53-173, model/network_utils.py:144-148).
//
//   D[co][pixel] = sum_k W[co][k] * X[k][pixel],  k = (ci, ky, kx)
// * A operand = weights.  PyTorch's [Cout][Cin][3][3] layout IS a row-major [Cout][K] matrix, so a slab of CI_S
//   input channels is a contiguous run of CI_S*TAPS floats per output channel: staged by float4, read back as
//   ds_read_b128 fragments exactly like gemm_f32.h (lane half h takes the second half of the slab's channels).
// * B operand = the input, NEVER im2col'ed: the block stages the raw input rows of its pixel tile (+ halo) once
//   per slab into LDS in NCHW order and every (ci, ky, kx) tap is a ds_read_b32 at a compile-time offset from the
//   lane's pixel address.  Global->LDS traffic per MFMA is therefore weights-dominated (~0.043 vector-memory
//   instructions per MFMA for a 128 x 256 tile; each costs the matrix pipe ~56 cycles, tools/mfma_probe.hip).
// * Pixel tile = 256 consecutive pixels in (image, y, x) order = R full rows of width W (W = 128..8), spanning
//   G = R/H whole images when the map is smaller than the tile; x halos are always outside the image (zeros that
//   are written once), y halos are loaded or zero.
// * Output: accumulator row = co, lane = pixel -> every store instruction writes 128-byte row segments of NCHW.
//   The output (and residual) image stride is a parameter, so results land directly in channel slices of the
//   decoder's concat buffers and of the lifting head's input tensor (no torch.cat, no chunk).
#pragma once
#include "common.h"
#include <type_traits>

template <int TAPS_, int STRIDE_, int LOG2W_, int CO_T_, int WCO_, int WPX_, int CI_S_>
struct ConvCfg {
    static constexpr int TAPS = TAPS_, STRIDE = STRIDE_, LOG2W = LOG2W_, CO_T = CO_T_, WCO = WCO_, WPX = WPX_, CI_S = CI_S_;
    static constexpr int KS = TAPS == 9 ? 3 : 1, PAD = (KS - 1) / 2;
    static constexpr int W = 1 << LOG2W;                 // output width == height (square maps)
    static constexpr int WIN = W * STRIDE;               // input width == height
    static constexpr int PX_T = 256;
    static constexpr int R = PX_T / W;                   // output rows per tile
    static constexpr int G = R > W ? R / W : 1;          // whole images per tile when the map is small
    static constexpr int RSEG = R > W ? W : R;           // output rows per image segment
    static constexpr int RI = (RSEG - 1) * STRIDE + KS;  // staged input rows per segment
    static constexpr int COL0 = TAPS == 9 ? 4 : 0;       // column of x = 0 inside a staged row (16-byte aligned)
    static constexpr int ROWW = WIN + (TAPS == 9 ? 8 : 0);
    static constexpr int CHS = G * RI * ROWW;            // staged floats per input channel
    static constexpr int KH = CI_S / 2 * TAPS;           // k per lane half per slab
    static constexpr int NT = KH / 4;
    static constexpr int LDK = 2 * KH + 4;               // padded weight row (floats)
    static constexpr int A_FLOATS = CO_T * LDK, B_FLOATS = CI_S * CHS;
    static constexpr int STAGE = A_FLOATS + B_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static constexpr int THREADS = 64 * WCO * WPX;
    static constexpr int TCO = CO_T / WCO / 32, TPX = PX_T / WPX / 32;
    static constexpr int A_V4 = CO_T * (2 * KH / 4);     // float4 per weight slab
    static constexpr int B_V4 = CI_S * G * RI * (WIN / 4);
    static constexpr int A_IT = (A_V4 + THREADS - 1) / THREADS, B_IT = (B_V4 + THREADS - 1) / THREADS;
    static_assert(KH % 4 == 0 && CI_S % 2 == 0, "slab must split in two halves of a multiple of 4 k");
    static_assert(CO_T % (32 * WCO) == 0 && PX_T % (32 * WPX) == 0, "wave tile must be 32x32 MFMA tiles");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(WIN % 4 == 0, "rows are staged by float4");
};

struct ConvArgs {
    const float* in;        // [Nimg][Cin][HIN][WIN] with image stride in_istride (floats)
    const float* w;         // [Cout][Cin][KS][KS]
    float* out;             // image stride out_istride, channel stride H*W
    const float* res;       // optional residual, image stride res_istride, same channel layout as out
    // per-output-channel affine: BatchNorm (gamma, beta, mean, var; eps 1e-5) when gamma != nullptr, else bias
    const float *gamma, *beta, *mean, *var, *bias;
    long in_istride, out_istride, res_istride;
    int Nimg, Cin, Cout, relu;
    int tiles_co, tiles_px;
    int vec_ok;             // set by the launcher: every per-channel vector is 16-byte aligned (float4 loads of 4 channels)
    // [r4] serving batches: the input channels are split over blockIdx.y (splits ranges of slabs), raw partial sums go to part[split][Nimg][Cout][H*W]
    // and conv_f32_reduce_kernel applies BatchNorm / bias, residual and ReLU.  The caller sets part / part_floats (scratch it owns), the launcher the rest.
    float* part;
    size_t part_floats;
    int splits;
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv_f32_kernel(ConvArgs a) {
    constexpr int TAPS = Cfg::TAPS, STRIDE = Cfg::STRIDE, KS = Cfg::KS, PAD = Cfg::PAD, W = Cfg::W, WIN = Cfg::WIN;
    constexpr int G = Cfg::G, RSEG = Cfg::RSEG, RI = Cfg::RI, COL0 = Cfg::COL0, ROWW = Cfg::ROWW, CHS = Cfg::CHS;
    constexpr int CI_S = Cfg::CI_S, KH = Cfg::KH, NT = Cfg::NT, LDK = Cfg::LDK, CO_T = Cfg::CO_T;
    constexpr int TCO = Cfg::TCO, TPX = Cfg::TPX, THREADS = Cfg::THREADS, STAGE = Cfg::STAGE, A_FLOATS = Cfg::A_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    int tpx, tco;   // consecutive blocks of one XCD walk pixel tiles of one output-channel tile (weights stay in L2)
    xcd_tile(blockIdx.x, gridDim.x, a.tiles_px, a.tiles_co, 32, tpx, tco);
    const int co0 = tco * CO_T;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wco = wid / Cfg::WPX, wpx = wid % Cfg::WPX;
    const int l31 = lane & 31, lh = lane >> 5;

    // tile origin: image n0 (segments g -> image n0 + g), first output row y0
    int n0, y0;
    if (G == 1) {
        const int gr0 = tpx * Cfg::R;
        n0 = gr0 / W;
        y0 = gr0 - n0 * W;
    } else {
        n0 = tpx * G;
        y0 = 0;
    }
    const int yin0 = y0 * STRIDE - PAD;
    const long ch_in = (long)WIN * WIN;

    // zero both input stages once: x halos are never written again
    for (int i = tid; i < Cfg::B_FLOATS; i += THREADS) {
        smem[A_FLOATS + i] = 0.f;
        smem[STAGE + A_FLOATS + i] = 0.f;
    }

    f32x4 pa[Cfg::A_IT], pb[Cfg::B_IT];
    const long wrow = (long)a.Cin * TAPS;     // floats per output channel
    auto gload = [&](int slab) {
        const int c0 = slab * CI_S;
#pragma unroll
        for (int it = 0; it < Cfg::A_IT; ++it) {
            const int idx = tid + it * THREADS;
            const int row = idx / (2 * KH / 4), f4 = idx - row * (2 * KH / 4);
            const long koff = (long)c0 * TAPS + f4 * 4;
            const bool ok = idx < Cfg::A_V4 && co0 + row < a.Cout && koff < wrow;
            pa[it] = ok ? *(const f32x4*)(a.w + (long)(co0 + row) * wrow + koff) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < Cfg::B_IT; ++it) {
            const int idx = tid + it * THREADS;
            const int c4 = idx % (WIN / 4);
            int rest = idx / (WIN / 4);
            const int rr = rest % RI;
            rest /= RI;
            const int g = rest % G, cl = rest / G;
            const int y = yin0 + rr, n = n0 + g, ci = c0 + cl;
            const bool ok = idx < Cfg::B_V4 && y >= 0 && y < WIN && n < a.Nimg && ci < a.Cin;
            pb[it] = ok ? *(const f32x4*)(a.in + (long)n * a.in_istride + (long)ci * ch_in + (long)y * WIN + c4 * 4)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_FLOATS;
#pragma unroll
        for (int it = 0; it < Cfg::A_IT; ++it) {
            const int idx = tid + it * THREADS;
            const int row = idx / (2 * KH / 4), f4 = idx - row * (2 * KH / 4);
            if (idx < Cfg::A_V4) *(f32x4*)(As + row * LDK + f4 * 4) = pa[it];
        }
#pragma unroll
        for (int it = 0; it < Cfg::B_IT; ++it) {
            const int idx = tid + it * THREADS;
            const int c4 = idx % (WIN / 4);
            const int rest = idx / (WIN / 4);     // = (cl * G + g) * RI + rr
            if (idx < Cfg::B_V4) *(f32x4*)(Bs + rest * ROWW + COL0 + c4 * 4) = pb[it];
        }
    };

    // per px-tile lane address inside a staged channel (floats), lane half h reads the slab's second half
    int laneb[TPX];
#pragma unroll
    for (int j = 0; j < TPX; ++j) {
        const int p = (wpx * TPX + j) * 32 + l31;
        const int g = p / (RSEG * W), rem = p - g * (RSEG * W);
        const int yy = rem / W, x = rem - yy * W;
        laneb[j] = lh * (CI_S / 2) * CHS + (g * RI + yy * STRIDE) * ROWW + x * STRIDE + COL0 - PAD;
    }
    const int a_off = (wco * TCO * 32 + l31) * LDK + lh * KH;

    f32x16 acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab_all = (a.Cin + CI_S - 1) / CI_S;
    int s_lo = 0, s_hi = nslab_all;
    if (a.splits > 1) {
        const int per = (nslab_all + a.splits - 1) / a.splits;
        s_lo = min((int)blockIdx.y * per, nslab_all);
        s_hi = min(nslab_all, s_lo + per);
    }
    const int nslab = s_hi - s_lo;       // (0 for a trailing range: the workgroup stores zeros)
    if (nslab > 0) gload(s_lo);
    __syncthreads();          // zero fill done before the first staged rows land
    if (nslab > 0) lstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(s_lo + s + 1);
        const float* As = smem + buf * STAGE + a_off;
        const float* Bs = smem + buf * STAGE + A_FLOATS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 af[TCO];
#pragma unroll
            for (int i = 0; i < TCO; ++i) af[i] = *(const f32x4*)(As + i * 32 * LDK + 4 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                constexpr int dummy = 0;
                (void)dummy;
                const int jj = 4 * t + u;                       // k index inside the lane half (compile time)
                const int cl = jj / TAPS, tap = jj - cl * TAPS;
                const int off = cl * CHS + (tap / KS) * ROWW + (tap % KS);
                float bv[TPX];
#pragma unroll
                for (int j = 0; j < TPX; ++j) bv[j] = Bs[laneb[j] + off];
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int j = 0; j < TPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: accumulator register r of lane l = D[co = 32x32 row (r&3) + 8*(r>>2) + 4*(l>>5)][pixel l&31]
    // One group = the 4 consecutive output channels (r&3) of one r>>2, all TPX pixel tiles: 4*TPX stores.  The
    // per-channel constants of group g+1 and the residuals of group g+1 are loaded BEFORE the stores of group g
    // are issued: vmcnt retires in order, so a load issued after a store cannot be waited for without waiting for
    // the store's write acknowledge as well (the first version waited vmcnt(0) in front of every single store).
    // Tiles with every channel and image in range run an unpredicated copy (the waitcnt pass counts it exactly).
    const long ch_out = (long)W * W;
    auto epilogue = [&](auto full_tag, auto bn_tag, auto res_tag) {
        constexpr bool FULL = decltype(full_tag)::value, BN = decltype(bn_tag)::value, RES = decltype(res_tag)::value;
        constexpr int NG = TCO * 4;
        long ooff[TPX], roff[TPX];
        bool pok[TPX];
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int p = (wpx * TPX + j) * 32 + l31;
            const int g = p / (RSEG * W), rem = p - g * (RSEG * W);
            const int n = n0 + g;
            pok[j] = FULL || n < a.Nimg;
            const long pix = (long)y0 * W + rem;       // (y0 + yy) * W + x
            ooff[j] = (long)n * a.out_istride + pix;
            roff[j] = (long)n * a.res_istride + pix;
        }
        struct Raw {
            f32x4 g, v, b, m;      // bias rides in b when there is no BatchNorm
        };
        auto group_co = [&](int idx) { return co0 + (wco * TCO + (idx >> 2)) * 32 + 8 * (idx & 3) + 4 * lh; };
        auto ld4 = [&](const float* ptr, int co) -> f32x4 {
            if (FULL) return *(const f32x4*)(ptr + co);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = co + e < a.Cout ? ptr[co + e] : 0.f;
            return o;
        };
        auto load_raw = [&](int idx) -> Raw {
            const int co = group_co(idx);
            Raw o;
            if (BN) {
                o.g = ld4(a.gamma, co);
                o.v = ld4(a.var, co);
                o.b = ld4(a.beta, co);
                o.m = ld4(a.mean, co);
            } else {
                o.b = ld4(a.bias, co);
                o.g = o.v = o.m = o.b;
            }
            return o;
        };
        struct Res {
            float v[4][TPX];
        };
        auto load_res = [&](int idx) -> Res {
            const int co = group_co(idx);
            Res o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TPX; ++j)
                    o.v[e][j] = (RES && (FULL || (co + e < a.Cout && pok[j]))) ? a.res[roff[j] + (co + e) * ch_out] : 0.f;
            return o;
        };
        Raw cur = load_raw(0);
        Res rcur = load_res(0);
#pragma unroll
        for (int idx = 0; idx < NG; ++idx) {
            Raw nxt = cur;
            Res rnxt = rcur;
            if (idx + 1 < NG) {
                nxt = load_raw(idx + 1);
                rnxt = load_res(idx + 1);
            }
            // the scheduler must not pull group idx+1's arithmetic (which waits for the loads just issued, i.e. for
            // everything older: group idx-1's stores) in front of group idx's stores
            __builtin_amdgcn_sched_barrier(0);
            const int i = idx >> 2, q = idx & 3;
            const int co = group_co(idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sc = 1.f, sh;
                if (BN) {
                    sc = cur.g[e] / sqrtf(cur.v[e] + 1e-5f);
                    sh = cur.b[e] - cur.m[e] * sc;
                } else {
                    sh = cur.b[e];
                }
#pragma unroll
                for (int j = 0; j < TPX; ++j) {
                    float v = acc[i][j][4 * q + e] * sc + sh;
                    if (RES) v += rcur.v[e][j];
                    v = a.relu ? fmaxf(v, 0.f) : v;
                    if (FULL || (co + e < a.Cout && pok[j])) a.out[ooff[j] + (co + e) * ch_out] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            rcur = rnxt;
        }
    };
    if (a.splits > 1) {       // raw partial sums of this range of input channels
        float* P = a.part + (size_t)blockIdx.y * a.Nimg * a.Cout * ch_out;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                for (int j = 0; j < TPX; ++j) {
                    const int p = (wpx * TPX + j) * 32 + l31;
                    const int g = p / (RSEG * W), rem = p - g * (RSEG * W);
                    const int n = n0 + g;
                    if (co < a.Cout && n < a.Nimg) P[((size_t)n * a.Cout + co) * ch_out + (long)y0 * W + rem] = acc[i][j][r];
                }
            }
        return;
    }
    const bool full = a.vec_ok && co0 + CO_T <= a.Cout && n0 + G <= a.Nimg;
    using T = std::true_type;
    using F = std::false_type;
    if (__builtin_expect(full, 1)) {
        if (a.gamma) {
            if (a.res) epilogue(T{}, T{}, T{});
            else epilogue(T{}, T{}, F{});
        } else {
            if (a.res) epilogue(T{}, F{}, T{});
            else epilogue(T{}, F{}, F{});
        }
    } else {
        if (a.gamma) {
            if (a.res) epilogue(F{}, T{}, T{});
            else epilogue(F{}, T{}, F{});
        } else {
            if (a.res) epilogue(F{}, F{}, T{});
            else epilogue(F{}, F{}, F{});
        }
    }
}

// sum of the channel-range partials + the convolution's epilogue (same expressions as conv_f32_kernel's); one thread = 4 pixels of one (image, channel)
static __global__ __launch_bounds__(256) void conv_f32_reduce_kernel(ConvArgs a, int HW) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total4 = (long)a.Nimg * a.Cout * HW / 4;
    if (i >= total4) return;
    const int hw4 = HW / 4;
    const int px = (int)(i % hw4) * 4;
    const long nc = i / hw4;
    const int co = (int)(nc % a.Cout), n = (int)(nc / a.Cout);
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)a.Nimg * a.Cout * HW;
    const float* P = a.part + (size_t)nc * HW + px;
    for (int k = 0; k < a.splits; ++k) acc += *(const f32x4*)(P + (size_t)k * stride);
    float sc = 1.f, sh;
    if (a.gamma) {
        sc = a.gamma[co] / sqrtf(a.var[co] + 1e-5f);
        sh = a.beta[co] - a.mean[co] * sc;
    } else {
        sh = a.bias[co];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = acc[e] * sc + sh;
    if (a.res) o += *(const f32x4*)(a.res + (long)n * a.res_istride + (long)co * HW + px);
    if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    *(f32x4*)(a.out + (long)n * a.out_istride + (long)co * HW + px) = o;
}

template <class Cfg>
static hipError_t conv_f32_launch(ConvArgs a, hipStream_t stream, int num_cu = 256) {
    if (a.Nimg <= 0) return hipSuccess;
    if ((a.Cin * Cfg::TAPS) % 4 != 0) return hipErrorInvalidValue;
    auto kern = conv_f32_kernel<Cfg>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    a.tiles_co = (a.Cout + Cfg::CO_T - 1) / Cfg::CO_T;
    a.vec_ok = (((size_t)a.gamma | (size_t)a.beta | (size_t)a.mean | (size_t)a.var | (size_t)a.bias) & 15) == 0;
    const long px = (long)a.Nimg * Cfg::W * Cfg::W;
    a.tiles_px = Cfg::G == 1 ? (int)(px / Cfg::PX_T) : (a.Nimg + Cfg::G - 1) / Cfg::G;
    // [r4] a few frames: 8 ... 64 workgroups each walking the whole Cin (layer4 at B = 1: 8 workgroups x 64 slabs = 315 us; conv_up2: 32 x 160 slabs =
    // 600 us, on 256 CUs) -> input-channel ranges over blockIdx.y until ~2 workgroups per CU, at least two slabs per range, partials within the scratch
    a.splits = 1;
    const long wgs = (long)a.tiles_co * a.tiles_px;
    const int nslab = (a.Cin + Cfg::CI_S - 1) / Cfg::CI_S;
    const size_t out_floats = (size_t)a.Nimg * a.Cout * Cfg::W * Cfg::W;
    // (the reduce kernel moves f32x4: image strides AND base pointers must be 16-byte multiples -- a caller's channel-offset view of `out` may be
    // only 4-byte aligned, and then the scalar epilogue of the unsplit kernel is the one that works)
    if (a.part != nullptr && 2 * wgs <= num_cu && nslab >= 4 && ((size_t)a.out_istride % 4 == 0) && (a.res == nullptr || a.res_istride % 4 == 0) &&
        (((size_t)a.out | (size_t)a.res | (size_t)a.part) & 15) == 0) {
        long sp = (2L * num_cu) / wgs;
        if (sp > nslab / 2) sp = nslab / 2;
        if (sp > 32) sp = 32;
        while (sp > 1 && (size_t)sp * out_floats > a.part_floats) --sp;
        if (sp > 1) {
            const int per = (nslab + (int)sp - 1) / (int)sp;
            a.splits = (nslab + per - 1) / per;             // no empty trailing range
        }
    }
    hipLaunchKernelGGL(kern, dim3(a.tiles_co * a.tiles_px, a.splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.splits == 1) return e;
    const long total4 = (long)out_floats / 4;
    hipLaunchKernelGGL(conv_f32_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, a, Cfg::W * Cfg::W);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------- ResNet stem and the glue kernels
// conv 7x7 stride 2 pad 3 (3 -> 64) + BatchNorm(eval) + ReLU, torchvision resnet18.conv1/bn1/relu via
// net_architecture.py:69.  K = 147 is too ragged for the MFMA tiling and only 1 % of the FLOPs: direct VALU
// convolution, 16 x 16 output pixels per block (one per thread), all 64 channels per thread, weights through
// wave-uniform (scalar) loads from a [tap][co] transposed copy in LDS.
// Two input pointers: image n = 2*b + eye reads eye ? right : left (the stereo pair is never concatenated).
static __global__ __launch_bounds__(256) void stem_conv7_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                         const float* __restrict__ w, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float* __restrict__ out, int HIN) {
    constexpr int TO = 16, TI = TO * 2 + 5;            // 37 x 37 input patch per channel
    __shared__ float xs[3 * TI * TI];                   // 16.4 KB
    __shared__ float ws[147 * 64];                      // [tap][co], 37.6 KB
    const int HO = HIN / 2;
    const int tiles = HO / TO;
    const int n = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x;
    const float* src = ((n & 1) ? right : left) + (long)(n >> 1) * 3 * HIN * HIN;
    const int tid = threadIdx.x;
    for (int i = tid; i < 147 * 64; i += 256) {
        const int co = i / 147, tap = i - co * 147;
        ws[tap * 64 + co] = w[i];
    }
    const int iy0 = ty * TO * 2 - 3, ix0 = tx * TO * 2 - 3;
    for (int i = tid; i < 3 * TI * TI; i += 256) {
        const int c = i / (TI * TI), rem = i - c * TI * TI, yy = rem / TI, xx = rem - yy * TI;
        const int y = iy0 + yy, x = ix0 + xx;
        xs[i] = (y >= 0 && y < HIN && x >= 0 && x < HIN) ? src[((long)c * HIN + y) * HIN + x] : 0.f;
    }
    __syncthreads();
    const int oy = tid >> 4, ox = tid & 15;
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float xv = xs[(c * TI + oy * 2 + ky) * TI + ox * 2 + kx];
                const float* wp = ws + ((c * 7 + ky) * 7 + kx) * 64;
#pragma unroll
                for (int co = 0; co < 64; ++co) acc[co] = fmaf(xv, wp[co], acc[co]);
            }
    const int y = ty * TO + oy, x = tx * TO + ox;
    float* dst = out + (long)n * 64 * HO * HO + (long)y * HO + x;
#pragma unroll
    for (int co = 0; co < 64; ++co) {
        if (gamma == nullptr) {            // training: raw convolution output, BatchNorm (batch statistics) follows
            dst[(long)co * HO * HO] = acc[co];
            continue;
        }
        const float sc = gamma[co] / sqrtf(var[co] + 1e-5f);
        const float v = acc[co] * sc + (beta[co] - mean[co] * sc);
        dst[(long)co * HO * HO] = fmaxf(v, 0.f);
    }
    (void)tiles;
}

// The stem on the matrix cores: implicit GEMM  out[pixel][co] = sum_k patch[pixel][k] * w[co][k],  k = (c, ky, kx) < 147 (one zero
// column pads K to 148 = 74 MFMA steps of 2).  A workgroup of 8 waves owns 8 output rows x 128 output columns of one image: the
// 21 x 261 x 3 input patch (zero halo) and the [k][co] weights sit in LDS (104 KB), wave w takes output row w as four 32-pixel
// MFMA row tiles x two 32-channel column tiles.  An A value is one ds_read_b32 at patch[c][2 row + ky][2 x + kx] (lane = pixel,
// lane half = k parity), shared by both column tiles; a B value is Ws[k][co] (lane = channel), shared by the four row tiles: 6
// LDS reads per 8 MFMAs.  Workgroups are persistent over (image, row group, column half) items, the weights are staged once.
struct StemCfg {
    static constexpr int RG = 8, XT = 128, PR = 2 * RG + 5, PC = 2 * XT + 5, PLD = 264, KP = 148;
    static constexpr int XS_FLOATS = 3 * PR * PLD, WS_FLOATS = KP * 64;
    static constexpr int LDS_BYTES = (XS_FLOATS + WS_FLOATS) * 4;
    static constexpr int THREADS = 64 * RG;
};
// BF16OUT: the output goes out as bf16 channels-last with the two eyes interleaved -- [B * HO * HO, 2 x 64], image n = 2 b + eye in
// the 64-channel slice eye * 64 of pixel row b * HO * HO + y * HO + x (the layout of conv_bf16s.h's backbone) -- instead of fp32 NCHW.
template <bool BF16OUT>
static __global__ __launch_bounds__(StemCfg::THREADS, 2) void stem_conv7_mfma_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                                                    const float* __restrict__ w, const float* __restrict__ gamma,
                                                                                    const float* __restrict__ beta, const float* __restrict__ mean,
                                                                                    const float* __restrict__ var, float* __restrict__ out, int HIN,
                                                                                    int nimg) {
    using Cfg = StemCfg;
    constexpr int RG = Cfg::RG, XT = Cfg::XT, PR = Cfg::PR, PC = Cfg::PC, PLD = Cfg::PLD, KP = Cfg::KP, THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) float stem_sm[];
    float* xs = stem_sm;                          // [3][PR][PLD]
    float* ws = stem_sm + Cfg::XS_FLOATS;         // [KP][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int HO = HIN / 2, xsegs = HO / XT, ygroups = HO / RG;
    const long items = (long)nimg * ygroups * xsegs;
    for (int i = tid; i < KP * 64; i += THREADS) {
        const int k = i >> 6, co = i & 63;
        ws[i] = k < 147 ? w[co * 147 + k] : 0.f;
    }
    // per-lane BatchNorm constants of its two output channels (eval mode); gamma == nullptr: raw convolution output (training)
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (gamma != nullptr) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = nt * 32 + l31;
            sc[nt] = gamma[co] / sqrtf(var[co] + 1e-5f);
            sh[nt] = beta[co] - mean[co] * sc[nt];
        }
    }
    // the patch of the NEXT item is requested into registers before the MFMAs of the current one and written to LDS behind them
    constexpr int NPRE = (3 * PR * PC + THREADS - 1) / THREADS;
    float pre[NPRE];
    auto request = [&](long item) __attribute__((always_inline)) {
        const int xseg_ = (int)(item % xsegs), yg_ = (int)((item / xsegs) % ygroups);
        const int n_ = (int)(item / ((long)xsegs * ygroups));
        const float* src = ((n_ & 1) ? right : left) + (long)(n_ >> 1) * 3 * HIN * HIN;
        const int iy0 = yg_ * RG * 2 - 3, ix0 = xseg_ * XT * 2 - 3;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + j * THREADS;
            const int c = i / (PR * PC), rem = i - c * PR * PC, yy = rem / PC, xx = rem - yy * PC;
            const int y = iy0 + yy, x = ix0 + xx;
            pre[j] = (i < 3 * PR * PC && y >= 0 && y < HIN && x >= 0 && x < HIN) ? src[((long)c * HIN + y) * HIN + x] : 0.f;
        }
    };
    auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + j * THREADS;
            const int c = i / (PR * PC), rem = i - c * PR * PC, yy = rem / PC, xx = rem - yy * PC;
            if (i < 3 * PR * PC) xs[(c * PR + yy) * PLD + xx] = pre[j];
        }
    };
    if ((long)blockIdx.x < items) request(blockIdx.x);
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int xseg = (int)(it % xsegs), yg = (int)((it / xsegs) % ygroups);
        const int n = (int)(it / ((long)xsegs * ygroups));
        __syncthreads();                          // everyone is done with the previous patch (and the weights are staged)
        publish();
        __syncthreads();
        if (it + gridDim.x < items) request(it + gridDim.x);
        f32x16 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        const float* ap = xs + (2 * wid) * PLD + 2 * l31;        // output row wid, pixel l31 of row tile 0
        const float* bp = ws + lh * 64 + l31;
#pragma unroll
        for (int s2 = 0; s2 < KP / 2; ++s2) {
            constexpr int dummy = 0; (void)dummy;
            const int k0 = 2 * s2, k1 = 2 * s2 + 1;              // this lane half multiplies k = k0 + lh
            const int o0 = ((k0 / 49) * PR + (k0 % 49) / 7) * PLD + (k0 % 7);
            const int o1 = k1 < 147 ? ((k1 / 49) * PR + (k1 % 49) / 7) * PLD + (k1 % 7) : 0;
            const int ko = lh ? o1 : o0;
            float a[4], b[2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = ap[ko + 64 * mt];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b[nt] = bp[k0 * 64 + nt * 32];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
        const int y = yg * RG + wid;
        if constexpr (BF16OUT) {
            // lane = channel: 32 lanes write the 64 contiguous bytes of 32 channels of one pixel (the lane halves: two pixels)
            __bf16* ob = (__bf16*)out + ((((long)(n >> 1) * HO + y) * HO + xseg * XT) * 2 + (n & 1)) * 64 + l31;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int x = mt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                        const float t = acc[mt][nt][r] * sc[nt] + sh[nt];
                        ob[(long)x * 128 + nt * 32] = (__bf16)(gamma != nullptr ? fmaxf(t, 0.f) : t);
                    }
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float* dst = out + ((long)n * 64 + nt * 32 + l31) * HO * HO + (long)y * HO + xseg * XT;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float t = acc[mt][nt][4 * g + c] * sc[nt] + sh[nt];
                        v[c] = gamma != nullptr ? fmaxf(t, 0.f) : t;
                    }
                    *(f32x4*)(dst + mt * 32 + 8 * g + 4 * lh) = v;
                }
        }
    }
}
static inline hipError_t stem_conv7_launch(const float* left, const float* right, const float* w, const float* gamma, const float* beta,
                                           const float* mean, const float* var, float* out, int HIN, int nimg, int num_cu, hipStream_t s) {
    using Cfg = StemCfg;
    const int HO = HIN / 2;
    if (HO % Cfg::XT == 0 && HO % Cfg::RG == 0 && ((uintptr_t)out & 15) == 0) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)stem_conv7_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        const long items = (long)nimg * (HO / Cfg::RG) * (HO / Cfg::XT);
        const int grid = (int)(items < num_cu ? items : num_cu);
        hipLaunchKernelGGL(stem_conv7_mfma_kernel<false>, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, left, right, w, gamma, beta, mean, var, out, HIN, nimg);
    } else {
        hipLaunchKernelGGL(stem_conv7_kernel, dim3(HO / 16, HO / 16, nimg), dim3(256), 0, s, left, right, w, gamma, beta, mean, var, out, HIN);
    }
    return hipGetLastError();
}

// MaxPool2d(3, stride 2, pad 1) on [N*C] planes of HIN x HIN (torchvision resnet18.maxpool)
static __global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                         int HIN) {
    const int HO = HIN / 2;
    const long total = planes * HO * HO;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % HO), y = (int)((i / HO) % HO);
        const long pl = i / ((long)HO * HO);
        const float* p = in + pl * HIN * HIN;
        float m = -INFINITY;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = 2 * y + dy, xx = 2 * x + dx;
                if (yy >= 0 && yy < HIN && xx >= 0 && xx < HIN) m = fmaxf(m, p[(long)yy * HIN + xx]);
            }
        out[i] = m;
    }
}

// The same pool, four outputs of a row per thread: a row of the 3 x 9 input window is two float4 and one scalar instead of nine
// scalar loads (the scalar kernel is bound by load instructions: 2.4 TB/s of HBM traffic against ~5 achievable).  HIN % 8 == 0.
static __global__ __launch_bounds__(256) void maxpool3s2_v4_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int HIN) {
    const int HO = HIN / 2, Q = HO / 4;
    const long total = planes * HO * Q;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xq = (int)(i % Q), y = (int)((i / Q) % HO);
    const long pl = i / ((long)HO * Q);
    const float* p = in + pl * HIN * HIN + 8 * xq;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = 2 * y + dy;
        if (yy < 0 || yy >= HIN) continue;
        const float* r = p + (long)yy * HIN;
        const f32x4 a = *(const f32x4*)r, b = *(const f32x4*)(r + 4);
        const float l = xq > 0 ? r[-1] : -INFINITY;          // column 8 xq - 1
        m[0] = fmaxf(m[0], fmaxf(l, fmaxf(a[0], a[1])));
        m[1] = fmaxf(m[1], fmaxf(a[1], fmaxf(a[2], a[3])));
        m[2] = fmaxf(m[2], fmaxf(a[3], fmaxf(b[0], b[1])));
        m[3] = fmaxf(m[3], fmaxf(b[1], fmaxf(b[2], b[3])));
    }
    *(f32x4*)(out + (pl * HO + y) * HO + 4 * xq) = m;
}
static inline void maxpool3s2_launch(const float* in, float* out, long planes, int HIN, hipStream_t s) {
    if (HIN % 8 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const long total = planes * (HIN / 2) * (HIN / 8);
        hipLaunchKernelGGL(maxpool3s2_v4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, planes, HIN);
    } else {
        hipLaunchKernelGGL(maxpool3s2_kernel, dim3(2048), dim3(256), 0, s, in, out, planes, HIN);
    }
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (net_architecture.py:126), written into a
// channel slice of the next concat buffer: out[n][c] at out + n*out_istride + c*4*HIN*HIN.
// One thread = four consecutive x of one output row (float4 store); NO grid-stride loop: with a loop hipcc emits a
// peeled and an unrolled copy of the body that round differently, and a pixel's value then depends on the batch size.
static __global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C,
                                                         int HIN, long in_istride, long out_istride) {
    const int HO = 2 * HIN, Q = HO / 4;
    const float scale = (float)(HIN - 1) / (float)(HO - 1);
    const long total = (long)N * C * HO * Q;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xq = (int)(i % Q), y = (int)((i / Q) % HO);
    const long nc = i / ((long)HO * Q);
    const int c = (int)(nc % C);
    const long n = nc / C;
    const float sy = scale * y;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < HIN - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly;
    const float* p = in + n * in_istride + (long)c * HIN * HIN;
    const float* r0 = p + (long)y0 * HIN;
    const float* r1 = p + (long)y1 * HIN;
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = xq * 4 + k;
        const float sx = scale * x;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < HIN - 1 ? 1 : 0);
        const float lx = sx - x0, hx = 1.f - lx;
        v[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
    }
    *(f32x4*)(out + n * out_istride + (long)c * HO * HO + (long)y * HO + xq * 4) = v;
}
