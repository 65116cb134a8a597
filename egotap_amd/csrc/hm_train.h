// Training-step kernels of the heatmap estimator (stage 1 of the reference: model/heatmap_shared_model.py:98-172 drives
// HeatMap_UnrealEgo_Shared, model/net_architecture.py:25-173, in train mode): everything the forward convolution kernels of
// conv_f32.h do not already provide.
//   * conv_wgrad_kernel   dW[co][ci][tap] = sum_{n,y,x} dY[n][co][y][x] * X[n][ci][y*s + ky - pad][x*s + kx - pad]
//                         implicit GEMM on the fp32 matrix cores: the contraction index is the PIXEL; a slab is one output row
//                         of one image: dY rows and the raw input rows (+ halo) are staged as they lie in memory and every
//                         (ci, tap) column of the product is a fixed LDS offset per lane.  Tile 128 co x (32 ci x 9 taps): wave w
//                         owns 32 output channels and all nine 32-column tiles (one dY read feeds nine MFMAs); layers with at most 64 output
//                         channels: 2 x 2 waves over 64 co x 64 ci.  Split over output rows ([r3]: the count comes from wgrad_pick_splits,
//                         gemm_tn_f32.h), partial slabs summed in a fixed order (reduce_slabs_kernel): no float atomics, bitwise reproducible.
//   * input gradients need no kernel of their own: dX = conv(dY, W^T flipped) runs on the forward kernels after
//     conv_wt_kernel (flip + channel swap); stride-2 layers first spread dY over the even pixels (zero_upsample2_kernel).
//   * BatchNorm2d in train mode (batch statistics over N*H*W, running-stat update, fused residual add and ReLU) forward
//     and backward, per-channel sums (bias gradients), ReLU / max-pool / bilinear-upsample backward, MSE loss.
#pragma once
#include "common.h"
#include "gemm_tn_f32.h"

// ------------------------------------------------------------------------------------------------- weight gradient
// [r3] CIW input channels per wave: a block is (CO_T / 32) x (CI_T / CIW) waves, wave (wco, wci) owns 32 output channels x CIW input channels x
// all taps (CIW = CI_T: the round-2 layout, four waves side by side over 128 output channels; 64-channel layers use 2 x 2 waves over 64 co x 64 ci,
// so that no wave multiplies rows that do not exist)
template <int KS_, int STRIDE_, int LOG2W_, int CI_T_, int CO_T_ = 128, int CIW_ = CI_T_>
struct WgCfg {
    static constexpr int KS = KS_, TAPS = KS_ * KS_, PAD = (KS_ - 1) / 2, STRIDE = STRIDE_, W = 1 << LOG2W_, CI_T = CI_T_, CIW = CIW_;
    static constexpr int WIN = W * STRIDE, CO_T = CO_T_, WCO = CO_T_ / 32, WCI = CI_T_ / CIW_, WAVES = WCO * WCI, THREADS = 64 * WAVES;
    static constexpr int NCOL = CIW * TAPS;                          // product columns of a wave: (ci_local, tap)
    static constexpr int NT = (NCOL + 31) / 32, NT_W = NT;           // 32-column tiles: every wave takes all of its own
    static_assert(CI_T_ % CIW_ == 0 && CO_T_ % 32 == 0, "wave grid");
    static constexpr int ALD = W + 1;                                // dY row stride in LDS (floats): conflict-free b32
    static constexpr int ROWW = WIN + 2 * PAD + 1, CHS = KS * ROWW;  // staged input row / floats per channel
    static constexpr int A_FLOATS = CO_T * ALD, B_FLOATS = CI_T * CHS;
    static constexpr int LDS_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;  // double buffered
    static_assert(W % 4 == 0 && WIN % 4 == 0, "rows are staged by float4");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct WgArgs {
    const float* dy;      // [Nimg][Cout][W][W], image stride dy_istride
    const float* x;       // [Nimg][Cin][WIN][WIN], image stride x_istride
    float* slabs;         // [splits][Cout][Cin * TAPS]
    long dy_istride, x_istride;
    int Nimg, Cin, Cout, tiles_co, tiles_ci, splits;
    int per;              // [r3] slabs (output rows of an image) per split: the split unit is the row, not the image
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_kernel(WgArgs a) {
    constexpr int KS = Cfg::KS, TAPS = Cfg::TAPS, PAD = Cfg::PAD, STRIDE = Cfg::STRIDE, W = Cfg::W, WIN = Cfg::WIN;
    constexpr int CI_T = Cfg::CI_T, CO_T = Cfg::CO_T, THREADS = Cfg::THREADS, NCOL = Cfg::NCOL, NT_W = Cfg::NT_W;
    constexpr int ALD = Cfg::ALD, ROWW = Cfg::ROWW, CHS = Cfg::CHS, A_FLOATS = Cfg::A_FLOATS, STAGE = Cfg::A_FLOATS + Cfg::B_FLOATS;
    constexpr int CIW = Cfg::CIW, WCO = Cfg::WCO;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wco = wid % WCO, wci = wid / WCO;
    const int split = blockIdx.x % a.splits, tile = blockIdx.x / a.splits;
    const int tco = tile % a.tiles_co, tci = tile / a.tiles_co;
    const int co0 = tco * CO_T, ci0 = tci * CI_T;
    const int s_lo = split * a.per, s_hi = min(a.Nimg * W, s_lo + a.per);
    const int nslab = max(0, s_hi - s_lo);                // one slab = one output row of one image

    // zero both input stages once: the x halo columns are never written again
    for (int i = tid; i < Cfg::B_FLOATS; i += THREADS) {
        wsm[A_FLOATS + i] = 0.f;
        wsm[STAGE + A_FLOATS + i] = 0.f;
    }
    constexpr int A_V4 = CO_T * (W / 4), B_V4 = CI_T * KS * (WIN / 4);
    constexpr int A_IT = (A_V4 + THREADS - 1) / THREADS, B_IT = (B_V4 + THREADS - 1) / THREADS;
    f32x4 pa[A_IT], pb[B_IT];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int n = (s_lo + s) / W, y = (s_lo + s) - n * W;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * THREADS, row = idx / (W / 4), c4 = idx - row * (W / 4);
            const bool ok = idx < A_V4 && co0 + row < a.Cout;
            pa[it] = ok ? *(const f32x4*)(a.dy + (long)n * a.dy_istride + ((long)(co0 + row) * W + y) * W + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * THREADS, c4 = idx % (WIN / 4), rest = idx / (WIN / 4), kr = rest % KS, cl = rest / KS;
            const int yin = y * STRIDE - PAD + kr;
            const bool ok = idx < B_V4 && ci0 + cl < a.Cin && yin >= 0 && yin < WIN;
            pb[it] = ok ? *(const f32x4*)(a.x + (long)n * a.x_istride + ((long)(ci0 + cl) * WIN + yin) * WIN + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        float* As = wsm + buf * STAGE;
        float* Bs = As + A_FLOATS;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * THREADS, row = idx / (W / 4), c4 = idx - row * (W / 4);
            if (idx < A_V4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[row * ALD + c4 * 4 + e] = pa[it][e];       // ALD is odd: scalar stores
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * THREADS, c4 = idx % (WIN / 4), rest = idx / (WIN / 4);   // rest = cl * KS + kr
            if (idx < B_V4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[rest * ROWW + PAD + c4 * 4 + e] = pb[it][e];
            }
        }
    };

    // per-lane column offsets: column n = ci_local * TAPS + tap reads X_lds[ci_local][ky][x * STRIDE + kx]
    int boff[NT_W];
    bool bok[NT_W];
#pragma unroll
    for (int t = 0; t < NT_W; ++t) {
        const int ncol = t * 32 + l31;
        bok[t] = ncol < NCOL;
        const int cl = min(ncol, NCOL - 1) / TAPS, tap = min(ncol, NCOL - 1) - cl * TAPS;
        boff[t] = (wci * CIW + cl) * CHS + (tap / KS) * ROWW + (tap % KS) + lh * STRIDE;
    }
    f32x16 acc[NT_W];
#pragma unroll
    for (int t = 0; t < NT_W; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nslab > 0) gload(0);
    __syncthreads();
    if (nslab > 0) lstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(s + 1);
        const float* As = wsm + buf * STAGE + (wco * 32 + l31) * ALD + lh;
        const float* Bs = wsm + buf * STAGE + A_FLOATS;
#pragma unroll 4
        for (int x0 = 0; x0 < W; x0 += 2) {                 // MFMA k = 2 pixels: lane half h takes pixel x0 + h
            const float a0 = As[x0];
#pragma unroll
            for (int t = 0; t < NT_W; ++t) {
                const float bv = bok[t] ? Bs[boff[t] + x0 * STRIDE] : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[t], 0, 0, 0);
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
    // accumulator register r of lane l = dW[co = 32x32 row (r&3) + 8*(r>>2) + 4*(l>>5)][column l&31]
    const long ldw = (long)a.Cin * TAPS;
    float* out = a.slabs + (long)split * a.Cout * ldw;
#pragma unroll
    for (int t = 0; t < NT_W; ++t) {
        const int ncol = t * 32 + l31;
        const int cl = wci * CIW + ncol / TAPS;
        if (ncol >= NCOL || ci0 + cl >= a.Cin) continue;
        const long col = (long)(ci0 + wci * CIW) * TAPS + ncol;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < a.Cout) out[(long)co * ldw + col] = acc[t][r];
        }
    }
}

template <class Cfg>
static hipError_t conv_wgrad_launch(WgArgs a, float* dw, size_t slab_bytes, int num_cu, int accumulate, hipStream_t stream) {
    if (a.Nimg <= 0) return hipSuccess;
    a.tiles_co = (a.Cout + Cfg::CO_T - 1) / Cfg::CO_T;
    a.tiles_ci = (a.Cin + Cfg::CI_T - 1) / Cfg::CI_T;
    const int tiles = a.tiles_co * a.tiles_ci;
    const long n = (long)a.Cout * a.Cin * Cfg::TAPS;
    // a slab of a wave: W / 2 k-steps x NT MFMAs of 64 cycles
    const int splits = wgrad_pick_splits(tiles, (long)a.Nimg * Cfg::W, n, slab_bytes, num_cu, Cfg::LDS_BYTES, Cfg::W / 2 * Cfg::NT * 64 / 2400.0, 4.0, &a.per);
    if (splits < 1) return hipErrorOutOfMemory;
    a.splits = splits;
    auto kern = conv_wgrad_kernel<Cfg>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a.slabs, dw, n, splits, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------- input-gradient helpers
// wt[ci][co][KS*KS - 1 - tap] = w[co][ci][tap]: the weights of the convolution that maps dY to dX
static __global__ __launch_bounds__(256) void conv_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int taps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, total = (long)Cout * Cin * taps;
    if (i >= total) return;
    const int tap = (int)(i % taps);
    const long cc = i / taps;
    const int ci = (int)(cc % Cin), co = (int)(cc / Cin);
    wt[((long)ci * Cout + co) * taps + (taps - 1 - tap)] = w[i];
}

// out[plane][2y][2x] = in[plane][y][x], zero elsewhere (the adjoint of a stride-2 subsampling); planes = N * C
static __global__ __launch_bounds__(256) void zero_upsample2_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int H,
                                                              long in_istride, long out_istride, int C) {
    const int HO = 2 * H;
    const long total = planes * HO * HO, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % HO), y = (int)((i / HO) % HO);
    const long pl = i / ((long)HO * HO);
    const long n = pl / C, c = pl % C;
    float v = 0.f;
    if (!(x & 1) && !(y & 1)) v = in[n * in_istride + (c * H + (y >> 1)) * H + (x >> 1)];
    out[n * out_istride + (c * HO + y) * HO + x] = v;
}

// ------------------------------------------------------------------------------------------------- BatchNorm2d (train)
// per-channel partial sums over a range of images: part[split][c][k], k = 0: sum a, 1: sum a*b   (b = nullptr: sum a*a)
// MODE 0: a = z, b = z (statistics).  MODE 1: a = dy' = dy * mask(y), b = zhat = (z - mean) * rstd (backward sums).
template <int MODE>
__global__ __launch_bounds__(256) void chan_sums_kernel(const float* __restrict__ A, const float* __restrict__ Z, const float* __restrict__ Y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, double* __restrict__ part,
                                                        int N, int C, int HW, long a_istride, long z_istride, int relu, int per) {
    __shared__ double red[2][256];
    const int c = blockIdx.x, split = blockIdx.y, tid = threadIdx.x;
    const int n_lo = split * per, n_hi = min(N, n_lo + per);
    double s0 = 0.0, s1 = 0.0;
    const float mu = MODE == 1 ? mean[c] : 0.f, rs = MODE == 1 ? rstd[c] : 0.f;
    for (int n = n_lo; n < n_hi; ++n) {
        const float* ap = A + (long)n * a_istride + (long)c * HW;
        const float* zp = MODE == 1 ? Z + (long)n * z_istride + (long)c * HW : nullptr;
        const float* yp = (MODE == 1 && relu) ? Y + (long)n * z_istride + (long)c * HW : nullptr;
        for (int i = tid * 4; i < HW; i += 1024) {
            const f32x4 av = *(const f32x4*)(ap + i);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0 += av[e]; s1 += (double)av[e] * av[e]; }
            } else {
                const f32x4 zv = *(const f32x4*)(zp + i);
                f32x4 yv = {1.f, 1.f, 1.f, 1.f};
                if (relu) yv = *(const f32x4*)(yp + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = yv[e] > 0.f ? av[e] : 0.f;
                    s0 += d;
                    s1 += (double)d * ((zv[e] - mu) * rs);
                }
            }
        }
    }
    red[0][tid] = s0; red[1][tid] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        part[((long)split * C + c) * 2] = red[0][0];
        part[((long)split * C + c) * 2 + 1] = red[1][0];
    }
}

// statistics -> mean, rstd (biased variance), running stats (momentum, unbiased variance): nn.BatchNorm2d in train mode
static __global__ __launch_bounds__(256) void bn2d_finish_kernel(const double* __restrict__ part, int splits, int C, double count, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < splits; ++k) { s0 += part[((long)k * C + c) * 2]; s1 += part[((long)k * C + c) * 2 + 1]; }
    const double mu = s0 / count;
    double var = s1 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * count / (count - 1.0));
    }
}

// y = [relu]( (z - mean) * rstd * gamma + beta [+ res] ), float4 per thread; all tensors [N][C][HW] with their own image strides
static __global__ __launch_bounds__(256) void bn2d_apply_kernel(const float* __restrict__ Z, float* __restrict__ Y, const float* __restrict__ R,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd, int N, int C, int HW,
                                                         long z_istride, long y_istride, long r_istride, int relu) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x, per = HW / 4, total = (long)N * C * per;
    if (q >= total) return;
    const int i4 = (int)(q % per);
    const long nc = q / per;
    const int c = (int)(nc % C);
    const long n = nc / C;
    const float sc = rstd[c] * gamma[c], sh = beta[c] - mean[c] * sc;
    f32x4 v = *(const f32x4*)(Z + n * z_istride + (long)c * HW + i4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc + sh;
    if (R) v += *(const f32x4*)(R + n * r_istride + (long)c * HW + i4 * 4);
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *(f32x4*)(Y + n * y_istride + (long)c * HW + i4 * 4) = v;
}

// sums -> dgamma, dbeta (per channel)
static __global__ __launch_bounds__(256) void bn2d_bwd_finish_kernel(const double* __restrict__ part, int splits, int C, float* __restrict__ sums,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < splits; ++k) { s0 += part[((long)k * C + c) * 2]; s1 += part[((long)k * C + c) * 2 + 1]; }
    sums[2 * c] = (float)s0;
    sums[2 * c + 1] = (float)s1;
    if (dgamma) {
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s1;
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s0;
    }
}

// dz = gamma * rstd * (dy' - sum(dy')/M - zhat * sum(dy' zhat)/M),  dy' = dy * [y > 0];  optional dres = dy' (residual branch)
static __global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ Y, const float* __restrict__ dY,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ sums, float* __restrict__ dZ,
                                                             float* __restrict__ dR, int N, int C, int HW, long z_istride, long dy_istride,
                                                             float inv_count, int relu, int dres_accumulate) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x, per = HW / 4, total = (long)N * C * per;
    if (q >= total) return;
    const int i4 = (int)(q % per);
    const long nc = q / per;
    const int c = (int)(nc % C);
    const long n = nc / C;
    const long zo = n * z_istride + (long)c * HW + i4 * 4, go = n * dy_istride + (long)c * HW + i4 * 4;
    const float mu = mean[c], rs = rstd[c], g = gamma[c] * rs, m0 = sums[2 * c] * inv_count, m1 = sums[2 * c + 1] * inv_count;
    const f32x4 zv = *(const f32x4*)(Z + zo), dv = *(const f32x4*)(dY + go);
    f32x4 yv = {1.f, 1.f, 1.f, 1.f};
    if (relu) yv = *(const f32x4*)(Y + zo);
    f32x4 o, dp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        dp[e] = yv[e] > 0.f ? dv[e] : 0.f;
        o[e] = g * (dp[e] - m0 - (zv[e] - mu) * rs * m1);
    }
    *(f32x4*)(dZ + zo) = o;
    if (dR) {
        if (dres_accumulate) dp += *(const f32x4*)(dR + zo);
        *(f32x4*)(dR + zo) = dp;
    }
}

// ------------------------------------------------------------------------------------------------- pointwise backward
// dz = dy * [y > 0] on [N][C][HW] slices with image strides (decoder convrelu blocks write into concat slices)
static __global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ dY, float* __restrict__ dZ, int N, int C,
                                                       int HW, long y_istride, long dy_istride, long dz_istride) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x, per = HW / 4, total = (long)N * C * per;
    if (q >= total) return;
    const int i4 = (int)(q % per);
    const long nc = q / per;
    const int c = (int)(nc % C);
    const long n = nc / C;
    const f32x4 yv = *(const f32x4*)(Y + n * y_istride + (long)c * HW + i4 * 4), dv = *(const f32x4*)(dY + n * dy_istride + (long)c * HW + i4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? dv[e] : 0.f;
    *(f32x4*)(dZ + n * dz_istride + (long)c * HW + i4 * 4) = o;
}

// MaxPool2d(3, 2, 1) backward: every input pixel collects dy of the (up to 4) windows whose FIRST maximum it is
// (row-major window scan, as torch's max_pool2d_with_indices picks)
static __global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dX,
                                                             long planes, int HIN) {
    const int HO = HIN / 2;
    const long total = planes * HIN * HIN, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % HIN), y = (int)((i / HIN) % HIN);
    const long pl = i / ((long)HIN * HIN);
    const float* xp = X + pl * HIN * HIN;
    const float* gp = dY + pl * HO * HO;
    float acc = 0.f;
    for (int oy = max(0, y / 2); oy <= min(HO - 1, (y + 1) / 2); ++oy)
        for (int ox = max(0, x / 2); ox <= min(HO - 1, (x + 1) / 2); ++ox) {
            // window of (oy, ox): rows 2oy-1 .. 2oy+1, cols 2ox-1 .. 2ox+1
            float m = -INFINITY;
            int my = -1, mx = -1;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = 2 * oy + dy, xx = 2 * ox + dx;
                    if (yy < 0 || yy >= HIN || xx < 0 || xx >= HIN) continue;
                    const float v = xp[(long)yy * HIN + xx];
                    if (v > m) { m = v; my = yy; mx = xx; }
                }
            if (my == y && mx == x) acc += gp[(long)oy * HO + ox];
        }
    dX[i] = acc;
}

// [r3] strip form of the same: a workgroup owns RI input rows of one plane.  It stages rows y0 - 1 .. y0 + RI + 1 in LDS, finds the first
// maximum of each of the (RI / 2 + 1) x HO windows that touch its rows ONCE (position and dy kept in LDS), and every input pixel then looks
// at its (up to four) windows.  Same scan order and same order of the sum as the per-pixel kernel above (which recomputes four 3 x 3
// scans per pixel from global memory: 1.33 ms for the 64 x 64 x 128^2 planes of a 32-frame step, against 0.6 GB of traffic).
template <int RI>
static __global__ __launch_bounds__(256) void maxpool3s2_bwd_strip_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dX,
                                                                          long planes, int HIN) {
    extern __shared__ __attribute__((aligned(16))) float mp_sm[];
    const int HO = HIN / 2, strips = (HIN + RI - 1) / RI, tid = threadIdx.x;
    const long pl = blockIdx.x / strips;
    const int y0 = (int)(blockIdx.x % strips) * RI, oy0 = y0 / 2;
    float* xs = mp_sm;                                          // [RI + 3][HIN]: row r = input row y0 - 1 + r
    int* wpos = (int*)(xs + (RI + 3) * HIN);                    // [RI / 2 + 1][HO]: yy * HIN + xx of the window's first maximum
    float* wdy = (float*)(wpos + (RI / 2 + 1) * HO);
    const float* xp = X + pl * HIN * HIN;
    const float* gp = dY + pl * HO * HO;
    for (int i = tid * 4; i < (RI + 3) * HIN; i += 1024) {
        const int r = i / HIN, c = i - r * HIN, yy = y0 - 1 + r;
        if (yy >= 0 && yy < HIN) *(f32x4*)(xs + i) = *(const f32x4*)(xp + (long)yy * HIN + c);
    }
    __syncthreads();
    for (int w = tid; w < (RI / 2 + 1) * HO; w += 256) {
        const int oyl = w / HO, ox = w - oyl * HO, oy = oy0 + oyl;
        int pos = -1;
        float g = 0.f;
        if (oy < HO) {
            float m = -INFINITY;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = 2 * oy + dy, xx = 2 * ox + dx;
                    if (yy < 0 || yy >= HIN || xx < 0 || xx >= HIN) continue;
                    const float v = xs[(yy - y0 + 1) * HIN + xx];
                    if (v > m) { m = v; pos = yy * HIN + xx; }
                }
            g = gp[(long)oy * HO + ox];
        }
        wpos[w] = pos;
        wdy[w] = g;
    }
    __syncthreads();
    for (int i = tid; i < RI * HIN; i += 256) {
        const int yl = i / HIN, x = i - yl * HIN, y = y0 + yl;
        if (y >= HIN) break;
        float acc = 0.f;
        for (int oy = y / 2; oy <= min(HO - 1, (y + 1) / 2); ++oy)
            for (int ox = x / 2; ox <= min(HO - 1, (x + 1) / 2); ++ox) {
                const int w = (oy - oy0) * HO + ox;
                if (wpos[w] == y * HIN + x) acc += wdy[w];
            }
        dX[pl * HIN * HIN + (long)y * HIN + x] = acc;
    }
}

// adjoint of nn.Upsample(scale 2, bilinear, align_corners=True): gather form, one thread per INPUT pixel
static __global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dX, int N, int C, int HIN,
                                                             long dy_istride, long dx_istride) {
    const int HO = 2 * HIN;
    const long total = (long)N * C * HIN * HIN, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % HIN), y = (int)((i / HIN) % HIN);
    const long nc = i / ((long)HIN * HIN);
    const int c = (int)(nc % C);
    const long n = nc / C;
    const float scale = (float)(HIN - 1) / (float)(HO - 1);
    const float* g = dY + n * dy_istride + (long)c * HO * HO;
    // output rows whose source interval touches y: sy = scale * oy in (y-1, y+1)
    const int oy_lo = max(0, (int)floorf((y - 1) / scale)), oy_hi = min(HO - 1, (int)ceilf((y + 1) / scale));
    const int ox_lo = max(0, (int)floorf((x - 1) / scale)), ox_hi = min(HO - 1, (int)ceilf((x + 1) / scale));
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float sy = scale * oy;
        const int y0 = (int)sy, y1 = y0 + (y0 < HIN - 1 ? 1 : 0);
        const float ly = sy - y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const float sx = scale * ox;
            const int x0 = (int)sx, x1 = x0 + (x0 < HIN - 1 ? 1 : 0);
            const float lx = sx - x0;
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            if (wx != 0.f) acc += wy * wx * g[(long)oy * HO + ox];
        }
    }
    dX[n * dx_istride + ((long)c * HIN + y) * HIN + x] = acc;
}

// MSE losses of heatmap_shared_model.py:109-151 on one [B][Cn][HW] prediction: loss = lambda * (mean over the left half +
// mean over the right half) of (p - g)^2 / plen, plen = gt_plength (limb maps) or 1; dpred likewise.  part[blocks] partial sums.
static __global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ P, const float* __restrict__ G, const float* __restrict__ plen,
                                                       float* __restrict__ dP, double* __restrict__ part, int B, int Cn, int HW, float coef) {
    __shared__ double red[256];
    const long total = (long)B * Cn * (HW / 4);
    double s = 0.0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long bc = q / (HW / 4);
        const float w = plen ? 1.0f / plen[bc] : 1.0f;
        const f32x4 p = *(const f32x4*)(P + q * 4), g = *(const f32x4*)(G + q * 4);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float diff = p[e] - g[e];
            s += (double)diff * diff * w;
            d[e] = 2.0f * coef * w * diff;
        }
        *(f32x4*)(dP + q * 4) = d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
static __global__ void mse_finish_kernel(const double* __restrict__ part, int n, float coef, float* __restrict__ out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = (float)(s * coef);
}

// ------------------------------------------------------------------------------------------------- weight gradient, bf16 pipe
// 3x3 stride-1 weight gradient on v_mfma_f32_32x32x16_bf16 (opt-in modes; arithmetic as gemm_bf16.h: NP = 3 hi + lo split,
// NP = 1 rounding).  The contraction index is the pixel, so a fragment is 8 CONSECUTIVE PIXELS of one dY row (A operand) or of
// one input row shifted by the tap's kx (B operand).  A 16-byte LDS read must be 16-byte aligned, which a shift by one bf16
// pixel is not: the input rows are staged THREE times, pre-shifted by kx = 0, 1, 2 (the neighbouring pixels come from the
// adjacent lane by a shuffle; the row ends are the zero halo), so every fragment of every tap is one aligned ds_read_b128.
// Slab = one output row of one image; tile 128 co (wave w = 32 channels) x 16 ci x 9 taps = 144 columns (five 32-column
// tiles, the last half empty); LDS double buffered (2 x 76 KB at width 64); split over images + fixed-order slab reduction.
#include "gemm_bf16.h"

template <int LOG2W_, int NP_>
struct WgBfCfg {
    static constexpr int W = 1 << LOG2W_, NP = NP_, NIMG = NP_ == 3 ? 2 : 1;
    static constexpr int CO_T = 128, CI_T = 16, THREADS = 256, NCOL = CI_T * 9, NT = (NCOL + 31) / 32;
    static constexpr int AST = W + 8;                                  // image row stride in bf16 (odd multiple of 16 bytes)
    static constexpr int AIMG = CO_T * AST, BIMG = CI_T * 9 * AST;     // per hi / lo part
    static constexpr int STAGE = NIMG * (AIMG + BIMG);                 // bf16 per buffer
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
    static constexpr int CPR = W / 8;                                  // 8-pixel chunks per row
    static constexpr int A_CH = CO_T * CPR, B_CH = CI_T * 3 * CPR;
    static constexpr int A_IT = (A_CH + THREADS - 1) / THREADS, B_IT = (B_CH + THREADS - 1) / THREADS;
    static_assert(W >= 16 && (AST * 2 / 16) % 2 == 1, "row stride must be an odd number of 16-byte units");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_bf16_kernel(WgArgs a) {
    constexpr int W = Cfg::W, NP = Cfg::NP, NIMG = Cfg::NIMG, CO_T = Cfg::CO_T, CI_T = Cfg::CI_T, THREADS = Cfg::THREADS;
    constexpr int NCOL = Cfg::NCOL, NT = Cfg::NT, AST = Cfg::AST, AIMG = Cfg::AIMG, BIMG = Cfg::BIMG, STAGE = Cfg::STAGE;
    constexpr int CPR = Cfg::CPR, A_CH = Cfg::A_CH, B_CH = Cfg::B_CH, A_IT = Cfg::A_IT, B_IT = Cfg::B_IT;
    extern __shared__ __attribute__((aligned(16))) __bf16 gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x % a.splits, tile = blockIdx.x / a.splits;
    const int tco = tile % a.tiles_co, tci = tile / a.tiles_co;
    const int co0 = tco * CO_T, ci0 = tci * CI_T;
    const int s_lo = split * a.per, s_hi = min(a.Nimg * W, s_lo + a.per);
    const int nslab = max(0, s_hi - s_lo);

    f32x4 pa[A_IT][2], pb[B_IT][2];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int n = (s_lo + s) / W, y = (s_lo + s) - n * W;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * THREADS, row = idx / CPR, ch = idx - row * CPR;
            const bool ok = idx < A_CH && co0 + row < a.Cout;
            const float* p = a.dy + (long)n * a.dy_istride + ((long)(co0 + row) * W + y) * W + ch * 8;
            pa[it][0] = ok ? *(const f32x4*)p : f32x4{0.f, 0.f, 0.f, 0.f};
            pa[it][1] = ok ? *(const f32x4*)(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * THREADS, ch = idx % CPR, rowid = idx / CPR, kr = rowid % 3, cl = rowid / 3;
            const int yin = y - 1 + kr;
            const bool ok = idx < B_CH && ci0 + cl < a.Cin && yin >= 0 && yin < W;
            const float* p = a.x + (long)n * a.x_istride + ((long)(ci0 + cl) * W + yin) * W + ch * 8;
            pb[it][0] = ok ? *(const f32x4*)p : f32x4{0.f, 0.f, 0.f, 0.f};
            pb[it][1] = ok ? *(const f32x4*)(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        __bf16* Ah = gsm + buf * STAGE;                 // [NIMG][CO_T][AST]
        __bf16* Bh = Ah + NIMG * AIMG;                  // [NIMG][CI_T][3 ky][3 kx][AST]
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * THREADS, row = idx / CPR, ch = idx - row * CPR;
            bf16x8 h, l;
            bf16_split8(pa[it][0], pa[it][1], h, l);
            if (idx < A_CH) {
                *(bf16x8*)(Ah + row * AST + ch * 8) = h;
                if (NP == 3) *(bf16x8*)(Ah + AIMG + row * AST + ch * 8) = l;
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int idx = tid + it * THREADS, ch = idx % CPR, rowid = idx / CPR;      // rowid = cl * 3 + kr
            bf16x8 h, l;
            bf16_split8(pb[it][0], pb[it][1], h, l);
            // neighbours' edge pixels: left neighbour's pixel 7, right neighbour's pixel 0 (zero at the row ends = the x halo).
            // Every lane takes part in the shuffles (no divergence around them).
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 hv = __builtin_bit_cast(i32x4, h), lv = __builtin_bit_cast(i32x4, l);
            int hl = __shfl_up(hv[3], 1, 64), hr = __shfl_down(hv[0], 1, 64), ll = __shfl_up(lv[3], 1, 64), lr = __shfl_down(lv[0], 1, 64);
            if (ch == 0) { hl = 0; ll = 0; }
            if (ch == CPR - 1) { hr = 0; lr = 0; }
            auto shifted = [&](const i32x4& v, int left, int right, int kx) __attribute__((always_inline)) {
                // bf16 elements e0..e7 packed two per dword (low half first).  kx = 0: [L7, e0..e6]; 1: [e0..e7]; 2: [e1..e7, R0]
                i32x4 o = v;
                if (kx == 0) {
                    o[0] = (int)(((unsigned)left >> 16) | ((unsigned)v[0] << 16));
                    o[1] = (int)(((unsigned)v[0] >> 16) | ((unsigned)v[1] << 16));
                    o[2] = (int)(((unsigned)v[1] >> 16) | ((unsigned)v[2] << 16));
                    o[3] = (int)(((unsigned)v[2] >> 16) | ((unsigned)v[3] << 16));
                } else if (kx == 2) {
                    o[0] = (int)(((unsigned)v[0] >> 16) | ((unsigned)v[1] << 16));
                    o[1] = (int)(((unsigned)v[1] >> 16) | ((unsigned)v[2] << 16));
                    o[2] = (int)(((unsigned)v[2] >> 16) | ((unsigned)v[3] << 16));
                    o[3] = (int)(((unsigned)v[3] >> 16) | ((unsigned)right << 16));
                }
                return o;
            };
            if (idx < B_CH) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    __bf16* dst = Bh + (rowid * 3 + kx) * AST + ch * 8;
                    *(i32x4*)dst = shifted(hv, hl, hr, kx);
                    if (NP == 3) *(i32x4*)(dst + BIMG) = shifted(lv, ll, lr, kx);
                }
            }
        }
    };

    int boff[NT];
    bool bok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ncol = t * 32 + l31;
        bok[t] = ncol < NCOL;
        boff[t] = min(ncol, NCOL - 1) * AST + 8 * lh;          // column n = ci_local * 9 + ky * 3 + kx  ==  image row index
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nslab > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(s + 1);
        const __bf16* Ah = gsm + buf * STAGE + (wid * 32 + l31) * AST + 8 * lh;
        const __bf16* Bh = gsm + buf * STAGE + NIMG * AIMG;
#pragma unroll
        for (int k = 0; k < W / 16; ++k) {
            const bf16x8 ah = *(const bf16x8*)(Ah + 16 * k);
            bf16x8 al;
            if (NP == 3) al = *(const bf16x8*)(Ah + AIMG + 16 * k);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bf16x8 bh = *(const bf16x8*)(Bh + boff[t] + 16 * k);
                if (!bok[t]) bh = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                if (NP == 3) {
                    bf16x8 bl = *(const bf16x8*)(Bh + BIMG + boff[t] + 16 * k);
                    if (!bok[t]) bl = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                }
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
    const long ldw = (long)a.Cin * 9;
    float* out = a.slabs + (long)split * a.Cout * ldw;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ncol = t * 32 + l31;
        const int cl = ncol / 9;
        if (ncol >= NCOL || ci0 + cl >= a.Cin) continue;
        const long col = (long)ci0 * 9 + ncol;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < a.Cout) out[(long)co * ldw + col] = acc[t][r];
        }
    }
}

template <class Cfg>
static hipError_t conv_wgrad_bf16_launch(WgArgs a, float* dw, size_t slab_bytes, int num_cu, int accumulate, hipStream_t stream) {
    if (a.Nimg <= 0) return hipSuccess;
    a.tiles_co = (a.Cout + Cfg::CO_T - 1) / Cfg::CO_T;
    a.tiles_ci = (a.Cin + Cfg::CI_T - 1) / Cfg::CI_T;
    const int tiles = a.tiles_co * a.tiles_ci;
    const long n = (long)a.Cout * a.Cin * 9;
    // a slab: W / 16 k-steps x NT tiles x NP MFMAs of 32 cycles per wave, or the staging of (128 + 48) fp32 rows through L2 -- whichever is longer
    const double mfma_us = Cfg::W / 16 * Cfg::NT * Cfg::NP * 32 / 2400.0, stage_us = (Cfg::CO_T + 3 * Cfg::CI_T) * Cfg::W * 4 / 24.0e3;
    const int splits = wgrad_pick_splits(tiles, (long)a.Nimg * Cfg::W, n, slab_bytes, num_cu, Cfg::LDS_BYTES, mfma_us > stage_us ? mfma_us : stage_us, 4.0, &a.per);
    if (splits < 1) return hipErrorOutOfMemory;
    a.splits = splits;
    auto kern = conv_wgrad_bf16_kernel<Cfg>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a.slabs, dw, n, splits, accumulate);
    return hipGetLastError();
}
