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

static __global__ __launch_bounds__(256) void pu_step_kernel(const float* __restrict__ F_t, int ldf,
                                                      const float* __restrict__ Gin_t,
                                                      const float* __restrict__ Whh,
                                                      const float* __restrict__ bhh,
                                                      const float* __restrict__ h_prev, const float* c_prev,
                                                      float* c_out, float* __restrict__ h_out,
                                                      float* __restrict__ gpre_out, int B, int H) {
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

    // operands of the LSTM pointwise (wave 0 finishes the step): requested before the k loop, so their latency is hidden by it
    // instead of being paid row by row behind the reduction (c_out may alias c_prev: every element is read here, once, by the
    // thread that later writes it)
    const int unit = u0 + l31;
    float gin[16][4], cpv[16], bg[4];
    if (wid == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bg[g] = bhh[g * H + unit];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(r0 + (r & 3) + 8 * (r >> 2) + 4 * lh, B - 1);
            const float* gi = Gin_t + (long)row * 4 * H + unit;
#pragma unroll
            for (int g = 0; g < 4; ++g) gin[r][g] = gi[(long)g * H];
            cpv[r] = c_prev[(long)row * H + unit];
        }
    }
    const float* hp = h_prev + (long)arow * H + k0 + 4 * lh;
    const float* fp = F_t + (long)arow * ldf + k0 + 4 * lh;
    const float* wp = Whh + (long)(u0 + l31) * H + k0 + 4 * lh;
    // The step is one of 30 dependent launches: a chunk of 32 k (4 k-steps x 6 float4 per lane) is requested ahead of the chunk
    // being multiplied so that no k-step waits for its own loads.  Measured: 32.7 -> 30.4 us per step at B = 256 (18.6 -> 19.9 at
    // B = 1): the loads were not the bulk of it -- 256 MFMAs per wave on one wave per SIMD (6.9 us), the LDS reduction and the
    // pointwise on a single wave are; the next step would be 16 waves per block (gate x k split).
    constexpr int CH = 4;                        // k-steps of 8 per chunk
    f32x4 xa[2][CH], xf[2][CH], xw[2][CH][4];
    auto request = [&](int buf, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = min(kc + 8 * c, kq - 8);   // kq % 32 != 0: the tail re-reads the last k-step (discarded by the guard below)
            xa[buf][c] = *(const f32x4*)(hp + k);
            xf[buf][c] = *(const f32x4*)(fp + k);
#pragma unroll
            for (int g = 0; g < 4; ++g) xw[buf][c][g] = *(const f32x4*)(wp + (long)g * H * H + k);
        }
    };
    auto multiply = [&](int buf, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (kc + 8 * c < kq) {
                f32x4 a = xa[buf][c];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] *= sigmoidf_(xf[buf][c][u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], xw[buf][c][g][u], acc[g], 0, 0, 0);
            }
        }
    };
    request(0, 0);
    for (int k = 0; k < kq; k += 16 * CH) {
        request(1, k + 8 * CH);
        multiply(0, k);
        request(0, k + 16 * CH);
        multiply(1, k + 8 * CH);
    }
    if (wid > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wid - 1) * 4 + g) * 16 + r) * 64 + lane] = acc[g][r];
    }
    __syncthreads();
    if (wid == 0) {
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
                const float pf = pre[0] + gin[r][0], pi = pre[1] + gin[r][1], pc = pre[2] + gin[r][2], po = pre[3] + gin[r][3];
                if (gpre_out) {      // training: keep the gate pre-activations for the backward pass
                    float* gp = gpre_out + (long)row * 4 * H + unit;
                    gp[0] = pf; gp[H] = pi; gp[2 * H] = pc; gp[3 * H] = po;
                }
                const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
                const float cn = cpv[r] * fg + ig * cg;
                c_out[(long)row * H + unit] = cn;
                h_out[(long)row * H + unit] = og * tanhf(cn);
            }
        }
    }
}

// The same step on 16 waves: wave = (gate, quarter of K) with one 32 x 32 accumulator, partial sums meet in LDS, and the LSTM
// pointwise runs one (row, unit) per thread on all 1024 threads.  pu_step_kernel spends its 20-30 us per step on 256 MFMAs per
// wave at one wave per SIMD (6.9 us), an LDS reduction into a single wave and that wave's 16-row pointwise; here a wave issues 64
// MFMAs, four waves share a SIMD, and nothing is serial behind one wave.  The k partition (quarters), the order inside a quarter
// and the order in which the four partial sums are added are those of pu_step_kernel, so the result is bit for bit the same.
static __global__ __launch_bounds__(1024) void pu_step16_kernel(const float* __restrict__ F_t, int ldf, const float* __restrict__ Gin_t,
                                                                const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                                const float* __restrict__ h_prev, const float* c_prev, float* c_out,
                                                                float* __restrict__ h_out, float* __restrict__ gpre_out, int B, int H) {
    __shared__ float red[16 * 16 * 64];          // [quarter * 4 + gate][register][lane] (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int g = wid & 3, kq4 = wid >> 2;
    const int r0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const int arow = min(r0 + l31, B - 1);
    const int kq = H >> 2, k0 = kq4 * kq;
    // operands of this thread's pointwise element, requested up front
    const int prow = r0 + (tid >> 5), punit = u0 + (tid & 31);
    const int prc = min(prow, B - 1);
    float gin[4], bg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gin[q] = Gin_t[(long)prc * 4 * H + (long)q * H + punit];
        bg[q] = bhh[q * H + punit];
    }
    const float cpv = c_prev[(long)prc * H + punit];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* hp = h_prev + (long)arow * H + k0 + 4 * lh;
    const float* fp = F_t + (long)arow * ldf + k0 + 4 * lh;
    const float* wp = Whh + (long)g * H * H + (long)(u0 + l31) * H + k0 + 4 * lh;
#pragma unroll 4
    for (int k = 0; k < kq; k += 8) {
        f32x4 a = *(const f32x4*)(hp + k);
        const f32x4 f = *(const f32x4*)(fp + k);
        const f32x4 w = *(const f32x4*)(wp + k);
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] *= sigmoidf_(f[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((kq4 * 4 + g) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (prow < B) {
        // element (row_local, unit) of a 32 x 32 accumulator: lane = unit + 32 * ((row_local >> 2) & 1), register (row_local & 3) + 4 * (row_local >> 3)
        const int rl = tid >> 5;
        const int idx = ((rl & 3) + 4 * (rl >> 3)) * 64 + (tid & 31) + 32 * ((rl >> 2) & 1);
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = red[(0 * 4 + q) * 1024 + idx];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[(w * 4 + q) * 1024 + idx];
            pre[q] = v + bg[q];
        }
        const float pf = pre[0] + gin[0], pi = pre[1] + gin[1], pc = pre[2] + gin[2], po = pre[3] + gin[3];
        if (gpre_out) {      // training: keep the gate pre-activations for the backward pass
            float* gp = gpre_out + (long)prow * 4 * H + punit;
            gp[0] = pf; gp[H] = pi; gp[2 * H] = pc; gp[3 * H] = po;
        }
        const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
        const float cn = cpv * fg + ig * cg;
        c_out[(long)prow * H + punit] = cn;
        h_out[(long)prow * H + punit] = og * tanhf(cn);
    }
}

// One PU step.  Both kernels give the same bits; measured per step: B = 1: 19.9 us (4 waves, operands requested ahead) against
// 22.7 (16 waves); B = 256: 30.4 against 28.0 -- neither is bound by its MFMAs (1.7 us on 16 waves), the floor is the chain of
// dependent memory round trips of a launch that starts cold 30 times per forward.
static inline void pu_step_launch(hipStream_t s, int B, int H, const float* F_t, int ldf, const float* Gin_t, const float* Whh, const float* bhh,
                                  const float* h_prev, const float* c_prev, float* c_out, float* h_out, float* gpre_out) {
    const dim3 grid((B + 31) / 32, H / 32);
    if (B >= 64) hipLaunchKernelGGL(pu_step16_kernel, grid, dim3(1024), 0, s, F_t, ldf, Gin_t, Whh, bhh, h_prev, c_prev, c_out, h_out, gpre_out, B, H);
    else hipLaunchKernelGGL(pu_step_kernel, grid, dim3(256), 0, s, F_t, ldf, Gin_t, Whh, bhh, h_prev, c_prev, c_out, h_out, gpre_out, B, H);
}

// Pose head: one block per sample.
//   pose_j = Wp . [left_j | right_j | skel_j] + bp  (+ global offset) ; UnrealEgo: head joint = global_mlp[3:6], output LAST.
// posz: [B*2J, hid] position embeddings (eye-major), hseq: [J, B, H] PU output (time-major).
static __global__ __launch_bounds__(256) void pose_head_kernel(const float* __restrict__ posz, const float* __restrict__ hseq,
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


// ----------------------------------------------------------------------------- backward of the propagation units
// One recurrent step, reverse time.  Pointwise part A (LSTM gates): from the total gradient on h_t and c_t to the
// gradient of the four gate pre-activations and of c_{t-1}.  The matrix part dhp = dG . Whh runs on the GEMM kernel
// (transposed weights); pointwise part B turns dhp into the gradient of h_{t-1} and of the x2f "forget" logits.
static __global__ __launch_bounds__(256) void pu_gates_bwd_kernel(const float* __restrict__ gpre, const float* __restrict__ c_prev,
                                                           const float* __restrict__ c_t, const float* __restrict__ dh_ext,
                                                           const float* __restrict__ dh_rec, const float* __restrict__ dc_next,
                                                           float* __restrict__ dG, float* __restrict__ dc_prev, int B, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H) return;
    const int row = (int)(i / H), u = (int)(i - (long)row * H);
    const float* gp = gpre + (long)row * 4 * H + u;
    const float f = sigmoidf_(gp[0]), ig = sigmoidf_(gp[H]), g = tanhf(gp[2 * H]), o = sigmoidf_(gp[3 * H]);
    const float tc = tanhf(c_t[i]);
    float dh = 0.f;
    if (dh_ext) dh += dh_ext[i];
    if (dh_rec) dh += dh_rec[i];
    float dc = dh * o * (1.0f - tc * tc);
    if (dc_next) dc += dc_next[i];
    float* dg = dG + (long)row * 4 * H + u;
    dg[0] = dc * c_prev[i] * f * (1.0f - f);
    dg[H] = dc * g * ig * (1.0f - ig);
    dg[2 * H] = dc * ig * (1.0f - g * g);
    dg[3 * H] = dh * tc * o * (1.0f - o);
    dc_prev[i] = dc * f;
}

// hp = sigmoid(f) * h_prev :  dh_prev = dhp * sigmoid(f) ;  df = dhp * h_prev * sigmoid'(f) ;  also emits hp (the
// operand of the h2h weight gradient)
static __global__ __launch_bounds__(256) void pu_hp_bwd_kernel(const float* __restrict__ dhp, const float* __restrict__ F_t, int ldf,
                                                        const float* __restrict__ h_prev, float* __restrict__ dh_rec_prev,
                                                        float* __restrict__ dF_t, int lddf, float* __restrict__ hp_out, int B, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H) return;
    const int row = (int)(i / H), u = (int)(i - (long)row * H);
    const float s = sigmoidf_(F_t[(long)row * ldf + u]);
    const float hprev = h_prev[i], d = dhp[i];
    dh_rec_prev[i] = d * s;
    dF_t[(long)row * lddf + u] = d * hprev * s * (1.0f - s);
    hp_out[i] = s * hprev;
}

// bridge gate of layer 0: b' = sigmoid(fb) * bridge ;  dfb = db' * bridge * sigmoid'(fb) (written into dF[:, H:]) ;
// dbridge = db' * sigmoid(fb) scattered back to the eye-major embedding layout [(b*2 + eye)*J + t, hid]
static __global__ __launch_bounds__(256) void pu_bridge_bwd_kernel(const float* __restrict__ dbp, const float* __restrict__ F, int ldf,
                                                            int fcol0, const float* __restrict__ rotz, float* __restrict__ dF,
                                                            float* __restrict__ drotz, int B, int J, int hid) {
    const int x = 2 * hid;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)J * B * x) return;
    const int m = (int)(i / x), k = (int)(i - (long)m * x);
    const int t = m / B, b = m - t * B, eye = k / hid, c = k - eye * hid;
    const long zi = ((long)(b * 2 + eye) * J + t) * hid + c;
    const float s = sigmoidf_(F[(long)m * ldf + fcol0 + k]);
    const float d = dbp[i], br = rotz[zi];
    dF[(long)m * ldf + fcol0 + k] = d * br * s * (1.0f - s);
    drotz[zi] = d * s;
}

// dposz[(b*2+eye)*J + t, c] = dxs[t*B + b, eye*hid + c] + dpose_feat contribution (already in dposz if accumulate)
static __global__ __launch_bounds__(256) void stereo_scatter_kernel(const float* __restrict__ dxs, float* __restrict__ dz, int B, int J,
                                                             int hid, int accumulate) {
    const int x = 2 * hid;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)J * B * x) return;
    const int m = (int)(i / x), k = (int)(i - (long)m * x);
    const int t = m / B, b = m - t * B, eye = k / hid, c = k - eye * hid;
    const long zi = ((long)(b * 2 + eye) * J + t) * hid + c;
    dz[zi] = (accumulate ? dz[zi] : 0.f) + dxs[i];
}

// Pose head backward.  Data gradients: one block per sample.
//   dpos[(b*2+eye)*J + j, c] = sum_k dpose'[b,j,k] Wp[k, eye*hid + c] ;  dskel[j,b,u] = sum_k dpose'[b,j,k] Wp[k, 2hid + u]
//                              + sum_o dother[b,o] Wg[o, j*H + u]   with dother[0:3] = sum_j dpose[b,j,:], dother[3:6] = dpose[b,J,:]
static __global__ __launch_bounds__(256) void pose_head_bwd_data_kernel(const float* __restrict__ dpose, const float* __restrict__ Wp,
                                                                 const float* __restrict__ Wg, float* __restrict__ dposz,
                                                                 float* __restrict__ dhseq, int B, int J, int hid, int H,
                                                                 int estimate_head) {
    __shared__ float dp[64 * 3], dother[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int JO = J + (estimate_head ? 1 : 0);
    for (int i = tid; i < JO * 3; i += 256) dp[i] = dpose[(long)b * JO * 3 + i];
    __syncthreads();
    if (tid < 6) {
        float s = 0.f;
        if (estimate_head) {
            if (tid < 3) for (int j = 0; j < J; ++j) s += dp[j * 3 + tid];
            else s = dp[J * 3 + tid - 3];
        }
        dother[tid] = s;
    }
    __syncthreads();
    const int kin = 2 * hid + H;
    for (int i = tid; i < J * kin; i += 256) {
        const int j = i / kin, k = i - j * kin;
        float s = dp[j * 3] * Wp[k] + dp[j * 3 + 1] * Wp[kin + k] + dp[j * 3 + 2] * Wp[2 * kin + k];
        if (k < 2 * hid) {
            const int eye = k / hid, c = k - eye * hid;
            dposz[((long)(b * 2 + eye) * J + j) * hid + c] = s;
        } else {
            const int u = k - 2 * hid;
            if (estimate_head)
#pragma unroll
                for (int o = 0; o < 6; ++o) s += dother[o] * Wg[(long)o * J * H + (long)j * H + u];
            dhseq[((long)j * B + b) * H + u] = s;
        }
    }
}

// Pose head weight gradients (tiny N): 32 lanes per weight column, lane l takes samples l, l + 32, ...; the lane partials meet in a
// fixed-order butterfly (deterministic).  One thread per column walked all B x J samples serially: 1.2 ms at B = 256, 5 ms at 1024.
static __global__ __launch_bounds__(256) void pose_head_bwd_weight_kernel(const float* __restrict__ dpose, const float* __restrict__ posz,
                                                                   const float* __restrict__ hseq, float* __restrict__ dWp,
                                                                   float* __restrict__ dbp, float* __restrict__ dWg,
                                                                   float* __restrict__ dbg, int B, int J, int hid, int H,
                                                                   int estimate_head, int accumulate) {
    const int kin = 2 * hid + H, JO = J + (estimate_head ? 1 : 0);
    const int col = blockIdx.x * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
    float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool is_w = col < kin, is_g = estimate_head && col >= kin && col < kin + J * H;
    const bool is_b = col == kin + (estimate_head ? J * H : 0);
    if (is_w) {                             // dWp[c][col] = sum_{b,j} dpose[b,j,c] * feat[b,j,col]
        for (int b = bl; b < B; b += 32)
            for (int j = 0; j < J; ++j) {
                float f;
                if (col < 2 * hid) {
                    const int eye = col / hid, c = col - eye * hid;
                    f = posz[((long)(b * 2 + eye) * J + j) * hid + c];
                } else {
                    f = hseq[((long)j * B + b) * H + col - 2 * hid];
                }
                const float* d = dpose + ((long)b * JO + j) * 3;
                s[0] += d[0] * f; s[1] += d[1] * f; s[2] += d[2] * f;
            }
    } else if (is_g) {                      // dWg[o][q] = sum_b dother[b,o] * skel_flat[b,q]
        const int q = col - kin, j = q / H, u = q - j * H;
        for (int b = bl; b < B; b += 32) {
            const float f = hseq[((long)j * B + b) * H + u];
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            for (int jj = 0; jj < J; ++jj) { const float* d = dpose + ((long)b * JO + jj) * 3; d0 += d[0]; d1 += d[1]; d2 += d[2]; }
            const float* dh = dpose + ((long)b * JO + J) * 3;
            s[0] += d0 * f; s[1] += d1 * f; s[2] += d2 * f; s[3] += dh[0] * f; s[4] += dh[1] * f; s[5] += dh[2] * f;
        }
    } else if (is_b) {                      // biases: s[0..2] pose bias, s[3..8] global_mlp bias
        for (int b = bl; b < B; b += 32) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            for (int j = 0; j < J; ++j) { const float* d = dpose + ((long)b * JO + j) * 3; d0 += d[0]; d1 += d[1]; d2 += d[2]; }
            s[0] += d0; s[1] += d1; s[2] += d2;
            if (estimate_head) {
                const float* dh = dpose + ((long)b * JO + J) * 3;
                s[3] += d0; s[4] += d1; s[5] += d2; s[6] += dh[0]; s[7] += dh[1]; s[8] += dh[2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s[i] += __shfl_xor(s[i], off, 64);
    if (bl != 0) return;
    if (is_w) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dWp[c * kin + col] = (accumulate ? dWp[c * kin + col] : 0.f) + s[c];
    } else if (is_g) {
        const int q = col - kin;
#pragma unroll
        for (int o = 0; o < 6; ++o) dWg[(long)o * J * H + q] = (accumulate ? dWg[(long)o * J * H + q] : 0.f) + s[o];
    } else if (is_b) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dbp[c] = (accumulate ? dbp[c] : 0.f) + s[c];
        if (estimate_head)
#pragma unroll
            for (int o = 0; o < 6; ++o) dbg[o] = (accumulate ? dbg[o] : 0.f) + s[3 + o];
    }
}
