// Row LayerNorm over D = 1024 floats (ViT hidden size), eps inside the sqrt, biased variance
// (modeling_vit.py:357-358, 537; nn.LayerNorm).  HBM-bound: one wave per row, the row lives in
// registers (4 x float4 per lane), two-pass mean / centred variance, wave reductions by DPP shuffles.
#pragma once
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int D>
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int rows, float eps) {
    static_assert(D % 256 == 0, "D must be a multiple of 256");
    constexpr int V = D / 256;   // float4 per lane
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    f32x4 g[V], b[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        g[i] = *(const f32x4*)(gamma + (i * 64 + lane) * 4);
        b[i] = *(const f32x4*)(beta + (i * 64 + lane) * 4);
    }
    for (int r = wave; r < rows; r += nwaves) {
        const float* x = X + (long)r * D;
        f32x4 v[V];
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = *(const f32x4*)(x + (i * 64 + lane) * 4);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = v[i][c] - mu;
                q += d * d;
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
        float* y = Y + (long)r * D;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mu) * rstd * g[i][c] + b[i][c];
            *(f32x4*)(y + (i * 64 + lane) * 4) = o;
        }
    }
}

// [r4] splitk_reduce_kernel<EpiBiasRes> for an N = D product FOLLOWED IN THE SAME PASS by the LayerNorm the ViT applies to its result
// (attention output projection -> LN2, MLP down -> the next layer's LN1 / the final LN): one wave per row sums the partials in split order,
// adds bias and residual, stores x and normalises it with layernorm_f32_kernel's arithmetic, operation for operation -- the same bits as the
// two launches it replaces (at B = 1: 6 of the forward's 66 launches, ~7 us each).
template <int D>
__global__ __launch_bounds__(256) void splitk_reduce_res_ln_kernel(const float* __restrict__ P, const float* __restrict__ bias, const float* R, float* X,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ Y,
                                                                   int rows, int splits, float eps) {
    static_assert(D % 256 == 0, "D must be a multiple of 256");
    constexpr int V = D / 256;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        f32x4 v[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int n = (i * 64 + lane) * 4;
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < splits; ++k) acc += *(const f32x4*)(P + ((long)k * rows + r) * D + n);
            v[i] = acc + *(const f32x4*)(bias + n) + *(const f32x4*)(R + (long)r * D + n);
            *(f32x4*)(X + (long)r * D + n) = v[i];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = v[i][c] - mu;
                q += d * d;
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int n = (i * 64 + lane) * 4;
            const f32x4 g = *(const f32x4*)(gamma + n), b = *(const f32x4*)(beta + n);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mu) * rstd * g[c] + b[c];
            *(f32x4*)(Y + (long)r * D + n) = o;
        }
    }
}
