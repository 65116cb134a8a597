// Propagation-Unit recurrence and the pose head (custom_cells.py:94-197,
// net_architecture.py:513-576, 732-751).
//
// The reference's PU writes its new state into the caller's tensors, so SkelNet's per-joint
// states alias one tensor and the "tree" is a chain over joint index: a 2-layer modified LSTM
// over J steps.  Everything that does not depend on the state (x2f, x2h,
// b2h of all J steps) is batched into ordinary GEMMs by the caller; what is left per step is
//     hp    = sigmoid(F_t[:, 0:H]) * h_{t-1}
//     gates = Gin_t + hp * Whh^T + bhh                       (gate order: forget, input, cell, output)
//     c_t   = c_{t-1} * sig(f) + sig(i) * tanh(g) ;  h_t = sig(o) * tanh(c_t)
// which pu_step_kernel does in one launch: a block owns 32 batch rows x 32 hidden units (all four
// gates of those units, so the LSTM pointwise runs on the MFMA accumulators), its 4 waves split
// K = H, partials meet in LDS.  Latency-bound by construction (30 dependent steps per forward).
#pragma once
#include "common.h"
#include "layernorm.h"   // wave_sum

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void pu_step_kernel(const float* __restrict__ F_t, int ldf,
                                                      const float* __restrict__ Gin_t,
                                                      const float* __restrict__ Whh,
                                                      const float* __restrict__ bhh,
                                                      const float* __restrict__ h_prev, float* __restrict__ c,
                                                      float* __restrict__ h_out, int B, int H) {
    __shared__ float red[3 * 4 * 16 * 64];   // partial accumulators of waves 1..3 (48 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int r0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const int arow = min(r0 + l31, B - 1);
    const int kq = H >> 2, k0 = wid * kq;

    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    const float* hp = h_prev + (long)arow * H + k0 + 4 * lh;
    const float* fp = F_t + (long)arow * ldf + k0 + 4 * lh;
    const float* wp = Whh + (long)(u0 + l31) * H + k0 + 4 * lh;
    for (int k = 0; k < kq; k += 8) {
        f32x4 a = *(const f32x4*)(hp + k);
        const f32x4 f = *(const f32x4*)(fp + k);
        f32x4 w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = *(const f32x4*)(wp + (long)g * H * H + k);
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] *= sigmoidf_(f[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w[g][u], acc[g], 0, 0, 0);
    }
    if (wid > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wid - 1) * 4 + g) * 16 + r) * 64 + lane] = acc[g][r];
    }
    __syncthreads();
    if (wid == 0) {
        const int unit = u0 + l31;
        float bg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bg[g] = bhh[g * H + unit];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = acc[g][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[((w * 4 + g) * 16 + r) * 64 + lane];
                pre[g] = v + bg[g];
            }
            if (row < B) {
                const float* gi = Gin_t + (long)row * 4 * H + unit;
                const float fg = sigmoidf_(pre[0] + gi[0]);
                const float ig = sigmoidf_(pre[1] + gi[H]);
                const float cg = tanhf(pre[2] + gi[2 * H]);
                const float og = sigmoidf_(pre[3] + gi[3 * H]);
                const float cn = c[(long)row * H + unit] * fg + ig * cg;
                c[(long)row * H + unit] = cn;
                h_out[(long)row * H + unit] = og * tanhf(cn);
            }
        }
    }
}

// Pose head: one block per sample.
//   pose_j = Wp . [left_j | right_j | skel_j] + bp  (+ global offset) ; UnrealEgo: head joint = global_mlp[3:6], output LAST.
// posz: [B*2J, hid] position embeddings (eye-major), hseq: [J, B, H] PU output (time-major).
__global__ __launch_bounds__(256) void pose_head_kernel(const float* __restrict__ posz, const float* __restrict__ hseq,
                                                        const float* __restrict__ Wp, const float* __restrict__ bp,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        float* __restrict__ pose, int B, int J, int hid, int H,
                                                        int estimate_head) {
    __shared__ float other[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = blockDim.x >> 6;
    if (tid < 8) other[tid] = 0.f;
    __syncthreads();
    if (estimate_head) {
        for (int o = wid; o < 6; o += nw) {
            float s = 0.f;
            for (int j = 0; j < J; ++j) {
                const float* hrow = hseq + ((long)j * B + b) * H;
                const float* wrow = Wg + (long)o * J * H + (long)j * H;
                for (int u = lane; u < H; u += 64) s += hrow[u] * wrow[u];
            }
            s = wave_sum(s);
            if (lane == 0) other[o] = s + bg[o];
        }
        __syncthreads();
    }
    const int kin = 2 * hid + H;
    const int nout = J * 3;
    for (int o = wid; o < nout; o += nw) {
        const int j = o / 3, cdim = o - j * 3;
        const float* wrow = Wp + (long)cdim * kin;
        const float* left = posz + ((long)b * 2 * J + j) * hid;
        const float* right = posz + ((long)b * 2 * J + J + j) * hid;
        const float* hrow = hseq + ((long)j * B + b) * H;
        float s = 0.f;
        for (int u = lane; u < hid; u += 64) s += left[u] * wrow[u] + right[u] * wrow[hid + u];
        for (int u = lane; u < H; u += 64) s += hrow[u] * wrow[2 * hid + u];
        s = wave_sum(s);
        if (lane == 0) pose[((long)b * (J + (estimate_head ? 1 : 0)) + j) * 3 + cdim] = s + bp[cdim] + other[cdim];
    }
    if (estimate_head && tid < 3) pose[((long)b * (J + 1) + J) * 3 + tid] = other[3 + tid];
}
