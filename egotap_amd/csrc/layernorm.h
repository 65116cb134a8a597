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
