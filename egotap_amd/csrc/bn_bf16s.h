// [r5] BatchNorm2d with BATCH statistics on the bf16 channels-last maps of the estimators' backbone -- what the reference's FROZEN
// estimators compute while the lifting head trains: train.py:91 model.train() leaves their BatchNorm2d in training mode
// (egotap_autoencoder_model.py:127-129 freezes parameters only, :177-216 runs them under autocast), the shared backbone runs once per
// eye (net_architecture.py:45-50), so every BatchNorm normalises each eye's batch with that eye's statistics and moves its running
// statistics twice per step (left, then right; momentum 0.1, unbiased variance, num_batches_tracked += 2).
//
// A backbone map is stored [B * S * S, 2 C] bf16, pixel (b, y, x) = [left C | right C] (conv_bf16s.h).  Per-eye, per-channel statistics over
// (batch, y, x) are therefore plain COLUMN statistics of that matrix, column = eye * C + channel, and the normalisation is a per-column
// scale / shift: no eye bookkeeping anywhere.
//   conv (raw bf16 z, the GEMM's unit-scale epilogue)  ->  bn_colstats_bf16s_kernel (sum, sum of squares per column: one read of z)
//   ->  bn_finish_bf16s_kernel (float64 fold of the partials, scale / shift per column, running statistics)  ->  bn_apply_bf16s_kernel
//   (y = relu?(z * scale + shift (+ residual)), in place).  All HBM-bound: z is written once, read twice, y written once.
// The statistics are those of the bf16 values the normalisation then reads (under autocast the reference's BatchNorm likewise sees the
// half-precision convolution output), accumulated in fp32 per thread (<= a few hundred values), folded in float64.
#pragma once
#include "gemm_bf16s.h"

// part[block][NC][2] = (sum, sum of squares) of rows [block * rows_per_block, ...) per column.  NC / 8 threads per row (16 bytes each).
static __global__ __launch_bounds__(256) void bn_colstats_bf16s_kernel(const __bf16* __restrict__ z, long R, int NC, long rows_per_block,
                                                                       float* __restrict__ part) {
    __shared__ float red[256 * 16];
    const int tid = threadIdx.x, tpr = NC >> 3;               // threads per row: 16 ... 128 (a power of two)
    const int c8 = tid & (tpr - 1), rsub = tid / tpr, rstep = 256 / tpr;
    const long r_lo = (long)blockIdx.x * rows_per_block, r_hi = min(R, r_lo + rows_per_block);
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    const __bf16* p = z + c8 * 8;
    long r = r_lo + rsub;
    for (; r + 3L * rstep < r_hi; r += 4L * rstep) {          // four rows in flight per thread
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(p + (r + (long)u * rstep) * NC);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float f = (float)v[u][i]; s[i] += f; q[i] += f * f; }
    }
    for (; r < r_hi; r += rstep) {
        const bf16x8 v = *(const bf16x8*)(p + r * NC);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; s[i] += f; q[i] += f * f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = s[i]; red[tid * 16 + 8 + i] = q[i]; }
    __syncthreads();
    // thread t < 2 NC: column t >> 1, quantity t & 1: fold the rstep row groups in a fixed order
    for (int t = tid; t < 2 * NC; t += 256) {
        const int col = t >> 1, qq = t & 1;
        float a = 0.f;
        for (int g = 0; g < rstep; ++g) a += red[(g * tpr + (col >> 3)) * 16 + qq * 8 + (col & 7)];
        part[((long)blockIdx.x * NC + col) * 2 + qq] = a;
    }
}

// one workgroup per channel (128 threads = 2 eyes x 64): both eyes' statistics from the partials -- float64, every thread a strided share of the
// partial rows, folded through LDS in a fixed order -- then scale / shift per column and the module's buffers updated as nn.BatchNorm2d does in
// training mode, left batch first, then right (net_architecture.py:45-50): running = (1 - m) running + m batch, unbiased variance, in fp32 with
// hm_train.h's bn2d_finish_kernel's expressions; num_batches_tracked += 2.  (One THREAD per channel walking up to 2048 partial rows took 0.25-1.1 ms
// per BatchNorm: 26 ms of a B = 1024 step.)
static __global__ __launch_bounds__(128) void bn_finish_bf16s_kernel(const float* __restrict__ part, int nparts, int C, double count, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                     long long* __restrict__ nbt, float* __restrict__ sc, float* __restrict__ sh) {
    __shared__ double red[2][128];
    const int c = blockIdx.x, tid = threadIdx.x, eye = tid >> 6, sub = tid & 63;
    const int col = eye * C + c;
    double s0 = 0.0, s1 = 0.0;
    for (int k = sub; k < nparts; k += 64) {
        const float2 v = *(const float2*)(part + ((long)k * 2 * C + col) * 2);
        s0 += (double)v.x; s1 += (double)v.y;
    }
    red[0][tid] = s0; red[1][tid] = s1;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (sub < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid != 0) return;
    const float momentum = 0.1f, eps = 1e-5f;
    for (int e = 0; e < 2; ++e) {
        const double mu = red[0][e * 64] / count;
        double var = red[1][e * 64] / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = rstd * gamma[c];
        sc[e * C + c] = scale;
        sh[e * C + c] = beta[c] - mean * scale;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * count / (count - 1.0));
        }
    }
    if (nbt && c == 0) *nbt += 2;
}

// y = relu?(z * scale[col] + shift[col] (+ res)) on [R, NC] bf16; out may be z (in place).  One thread = 8 columns of one row.
static __global__ __launch_bounds__(256) void bn_apply_bf16s_kernel(const __bf16* __restrict__ z, const __bf16* __restrict__ res, __bf16* __restrict__ out,
                                                                    const float* __restrict__ sc, const float* __restrict__ sh, long total8, int NC, int relu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int c8 = (int)(i & ((NC >> 3) - 1));
    const bf16x8 v = *(const bf16x8*)(z + i * 8);
    const f32x4 s0 = *(const f32x4*)(sc + c8 * 8), s1 = *(const f32x4*)(sc + c8 * 8 + 4);
    const f32x4 h0 = *(const f32x4*)(sh + c8 * 8), h1 = *(const f32x4*)(sh + c8 * 8 + 4);
    float y[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { y[j] = (float)v[j] * s0[j] + h0[j]; y[4 + j] = (float)v[4 + j] * s1[j] + h1[j]; }
    if (res) {
        const bf16x8 rr = *(const bf16x8*)(res + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] += (float)rr[j];
    }
    if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fmaxf(y[j], 0.f);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)y[j];
    *(bf16x8*)(out + i * 8) = o;
}

struct BnBatchScratch {       // caller-owned (workspace): reused by every BatchNorm of a forward, stream-ordered
    float* part;              // [MAX_PARTS][NC <= 1024][2]
    float *sc, *sh;           // [1024] each
    static constexpr int MAX_PARTS = 2048;
    static constexpr size_t part_floats() { return (size_t)MAX_PARTS * 1024 * 2; }
};
// statistics of z [R, 2 C] -> scale / shift in ws.sc / ws.sh (+ running statistics), then y = relu?(z * sc + sh (+ res)) in place
static inline hipError_t bn_batch_bf16s_launch(__bf16* z, const __bf16* res, long R, int C, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                               long long* nbt, int relu, const BnBatchScratch& ws, hipStream_t s) {
    const int NC = 2 * C;
    if (NC < 128 || NC > 1024 || (NC & (NC - 1)) != 0 || R < 2) return hipErrorInvalidValue;
    const int rstep = 256 / (NC >> 3);
    // workgroups: about 1024 of them (four per CU: the kernel is a stream), each at least 8 row steps, at most MAX_PARTS.  (64 row steps each left
    // a 32-frame batch 16-128 workgroups per launch -- the reference's own training batch: 18 us per BatchNorm where 6 suffice)
    long per = std::max<long>(8L * rstep, ((R + 1023) / 1024 + rstep - 1) / rstep * rstep);
    long nblk = (R + per - 1) / per;
    if (nblk > BnBatchScratch::MAX_PARTS) { per = ((R + BnBatchScratch::MAX_PARTS - 1) / BnBatchScratch::MAX_PARTS + rstep - 1) / rstep * rstep; nblk = (R + per - 1) / per; }
    hipLaunchKernelGGL(bn_colstats_bf16s_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const __bf16*)z, R, NC, per, ws.part);
    hipLaunchKernelGGL(bn_finish_bf16s_kernel, dim3(C), dim3(128), 0, s, (const float*)ws.part, (int)nblk, C, (double)R, gamma, beta, run_mean, run_var, nbt,
                       ws.sc, ws.sh);
    const long total8 = R * (NC >> 3);
    hipLaunchKernelGGL(bn_apply_bf16s_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, (const __bf16*)z, res, z, (const float*)ws.sc, (const float*)ws.sh,
                       total8, NC, relu);
    return hipGetLastError();
}
