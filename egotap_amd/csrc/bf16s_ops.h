// Row / column kernels of the bf16-storage training mode (EGOTAP_PREC_BF16): LayerNorm forward and backward with bf16 activations,
// column sums of bf16 matrices (bias gradients), per-step weight preparation (fp32 master weights -> bf16 copy + bf16 transposed copy
// for the input-gradient GEMMs).  All HBM-bound: 16-byte accesses, one pass over each tensor.
// Reference semantics: nn.LayerNorm(1024, eps 1e-12) of model/modeling_vit.py:357-358, 367, 378, 609 and its autograd.
#pragma once
#include "gemm_bf16s.h"
#include "train_ops.h"

// ---------------------------------------------------------------------------------------------------- LayerNorm forward
// one wave per row: y bf16 = (x - mean) * rstd * gamma + beta; mean / rstd kept for the backward (fp32)
template <int D>
__global__ __launch_bounds__(256) void ln_fwd_bf16_kernel(const float* __restrict__ X, __bf16* __restrict__ Y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                                                          int rows, float eps) {
    constexpr int V = D / 512;                     // 8-element chunks per lane
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const float* x = X + (long)r * D;
    f32x4 v[V][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 8;
        v[i][0] = *(const f32x4*)(x + c);
        v[i][1] = *(const f32x4*)(x + c + 4);
        s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
    }
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float d = v[i][h][c] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
    if (lane == 0 && mean) { mean[r] = mu; rstd[r] = rs; }
    __bf16* y = Y + (long)r * D;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 8;
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 g = *(const f32x4*)(gamma + c + 4 * h), b = *(const f32x4*)(beta + c + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * h + e] = (v[i][h][e] - mu) * rs * g[e] + b[e];
        }
        store_bf16x8(y + c, o);
    }
}

// ---------------------------------------------------------------------------------------------------- LayerNorm backward
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres);  dx is written as fp32 (the residual stream's gradient) and,
// when dXb != nullptr, as bf16 too (the operand of the next weight-gradient / input-gradient GEMMs).  Partial column sums per
// workgroup: part[block][3][D] = (sum dy*xhat, sum dy, sum dx) -- the third is the bias gradient of the Linear layer that produced
// the residual branch this dx flows into (its dY IS this dx), so no separate column-sum pass reads dx again.
template <int D>
__global__ __launch_bounds__(256) void ln_bwd_bf16_kernel(const float* __restrict__ X, const __bf16* __restrict__ dY, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ dres, float* __restrict__ dX, __bf16* __restrict__ dXb,
                                                          float* __restrict__ part, int rows, int rows_per_wave) {
    constexpr int V = D / 512;
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + w;
    f32x4 g[V][2], dg[V][2], db[V][2], ds[V][2];
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            g[i][h] = *(const f32x4*)(gamma + (i * 64 + lane) * 8 + 4 * h);
            dg[i][h] = db[i][h] = ds[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int r_lo = wave * rows_per_wave, r_hi = min(rows, r_lo + rows_per_wave);
    for (int r = r_lo; r < r_hi; ++r) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[V][2], dy[V][2], rsd[V][2];
        float s1 = 0.f, s2 = 0.f;
        // [r3] the residual gradient of the row is requested together with x and dy, ahead of the two wave reductions (it used to be
        // loaded after them: one more exposed round trip per row and wave)
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                rsd[i][h] = dres ? *(const f32x4*)(dres + (long)r * D + (i * 64 + lane) * 8 + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 8;
            const bf16x8 d8 = *(const bf16x8*)(dY + (long)r * D + c);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 x = *(const f32x4*)(X + (long)r * D + c + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (float)d8[4 * h + e];
                    dy[i][h][e] = d;
                    xh[i][h][e] = (x[e] - mu) * rs;
                    const float gd = g[i][h][e] * d;
                    s1 += gd;
                    s2 += gd * xh[i][h][e];
                    dg[i][h][e] += d * xh[i][h][e];
                    db[i][h][e] += d;
                }
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 8;
            float o8[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][h][e] * dy[i][h][e] - m1 - xh[i][h][e] * m2);
                o += rsd[i][h];
                *(f32x4*)(dX + (long)r * D + c + 4 * h) = o;
                ds[i][h] += o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o8[4 * h + e] = o[e];
            }
            if (dXb) store_bf16x8(dXb + (long)r * D + c, o8);
        }
    }
    // workgroup reduction of the three partial rows, one quantity at a time through a 4 KB buffer
#pragma unroll
    for (int which = 0; which < 3; ++which)
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();
                red[w][lane] = which == 0 ? dg[i][h] : (which == 1 ? db[i][h] : ds[i][h]);
                __syncthreads();
                if (w == 0) {
                    const f32x4 a = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
                    *(f32x4*)(part + ((long)blockIdx.x * 3 + which) * D + (i * 64 + lane) * 8 + 4 * h) = a;
                }
            }
}

// ---------------------------------------------------------------------------------------------------- column sums of a bf16 matrix
// part[gridDim.y][N] = sum over the block's rows of Y[m][n] (bias gradients); finished by reduce_slabs_kernel
static __global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(const __bf16* __restrict__ Y, long ldy, float* __restrict__ part, int M, int N,
                                                                         int rows_per_block) {
    __shared__ f32x4 red[4][2][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c8 = blockIdx.x * 64 + l;                          // 8-element column chunk
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (c8 * 8 < N)
        for (int m = m_lo + q; m < m_hi; m += 4) {
            const bf16x8 v = *(const bf16x8*)(Y + (long)m * ldy + c8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0[e] += (float)v[e]; s1[e] += (float)v[4 + e]; }
        }
    red[q][0][l] = s0;
    red[q][1][l] = s1;
    __syncthreads();
    if (q == 0 && c8 * 8 < N) {
        *(f32x4*)(part + (long)blockIdx.y * N + c8 * 8) = (red[0][0][l] + red[1][0][l]) + (red[2][0][l] + red[3][0][l]);
        *(f32x4*)(part + (long)blockIdx.y * N + c8 * 8 + 4) = (red[0][1][l] + red[1][1][l]) + (red[2][1][l] + red[3][1][l]);
    }
}

static hipError_t colsum_bf16_launch(const __bf16* Y, long ldy, float* out, int M, int N, int accumulate, float* part, size_t part_bytes, hipStream_t s) {
    if (N % 8 != 0) return hipErrorInvalidValue;
    int rows_per_block = 256;
    int gy = (M + rows_per_block - 1) / rows_per_block;
    while ((size_t)gy * N * 4 > part_bytes && rows_per_block < (1 << 20)) {
        rows_per_block *= 2;
        gy = (M + rows_per_block - 1) / rows_per_block;
    }
    if ((size_t)gy * N * 4 > part_bytes) return hipErrorOutOfMemory;
    hipLaunchKernelGGL(colsum_bf16_partial_kernel, dim3((N / 8 + 63) / 64, gy), dim3(256), 0, s, Y, ldy, part, M, N, rows_per_block);
    // two fixed-order stages when there are many partial rows
    float* src = part;
    int n_rows = gy;
    if (gy > 64 && (size_t)(gy + (gy + 63) / 64) * N * 4 <= part_bytes) {
        float* part2 = part + (size_t)gy * N;
        const int gy2 = (gy + 63) / 64;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((N / 4 + 63) / 64, gy2), dim3(256), 0, s, (const float*)part, (long)N, part2, gy, N, 64);
        src = part2;
        n_rows = gy2;
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, s, (const float*)src, out, (long)N, n_rows, accumulate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- weight preparation
// W fp32 [N, K] (nn.Linear layout, the live master weights) -> Wb bf16 [N, K] and, when Wt != nullptr, Wt bf16 [K, N] with row stride ldt (the operand of
// the input-gradient GEMM dX = dY W, which wants W^T rows K-contiguous).  64 x 64 tiles through LDS, 16-byte stores on both sides.
static __global__ __launch_bounds__(256) void prep_weight_kernel(const float* __restrict__ W, __bf16* __restrict__ Wb, __bf16* __restrict__ Wt, int N, int K, long ldt) {
    __shared__ float t[64][65];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 x 16 threads, float4 along k
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 16 * i;
        const f32x4 v = (n0 + r < N && k0 + tx * 4 < K) ? *(const f32x4*)(W + (long)(n0 + r) * K + k0 + tx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) t[r][tx * 4 + e] = v[e];
    }
    __syncthreads();
    const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;      // 8 chunks of 8 x 32 rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = ry + 32 * i;
        float o[8];
        if (n0 + r < N && k0 + cx * 8 < K) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = t[r][cx * 8 + e];
            store_bf16x8(Wb + (long)(n0 + r) * K + k0 + cx * 8, o);
        }
        if (Wt != nullptr && k0 + r < K && n0 + cx * 8 < N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = t[cx * 8 + e][r];
            store_bf16x8(Wt + (long)(k0 + r) * ldt + n0 + cx * 8, o);
        }
    }
}
