// Small kernels of the training step (egotap_autoencoder_model.py:284-323, utils/loss.py:54-85, network.py:72-78):
// loss forward/backward, pose-head backward, LayerNorm with saved statistics + backward, BatchNorm1d(train) +
// LeakyReLU forward/backward, GELU backward, fused AdamW.  All fp32; every reduction has a fixed order (no float
// atomics), so a training step is bitwise reproducible run to run.
#pragma once
#include "common.h"
#include "layernorm.h"

// ----------------------------------------------------------------------------- loss
// L = lam_pose * mean_{b,j} |gt - pred|_2  +  lam_cos * mean_b sum_bones cos(bone_pred, bone_gt)
// (LossFuncMPJPE utils/loss.py:83-85; LossFuncCosSim :54-77 with nn.CosineSimilarity's eps 1e-8 clamp on each norm).
// UnrealEgo: 16 joints, bones (j, parent[j]) for j = 1..15 of ue_kinematic_parents (utils/util.py:51).
// EgoCap: a zero root joint is prepended (18 joints, egocap_kinematic_parents :52), the first bone is dropped.
// One block per sample writes its two partial sums and d loss / d pred; a second tiny kernel adds the partials in order.
struct LossArgs {
    const float *pred, *gt;     // [B, J, 3]
    float* dpred;               // [B, J, 3]
    float* partial;             // [B, 2]  (sum_j dist, sum_bones cos)
    int B, J, estimate_head;
    float lam_pose, lam_cos;    // lambda_mpjpe, lambda_cos_sim * lambda_mpjpe
    int parents[20];
};

static __global__ __launch_bounds__(64) void pose_loss_kernel(LossArgs a) {
    __shared__ float P[20][3], G[20][3], Gb[20][3];     // Gb[t]: d loss / d bone_t (child side), zero where no bone
    const int b = blockIdx.x, t = threadIdx.x;
    const int J = a.J, off = a.estimate_head ? 0 : 1, JJ = J + off;     // JJ joints incl. the virtual EgoCap root
    if (t < JJ) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool real = t >= off;
            P[t][c] = real ? a.pred[((long)b * J + t - off) * 3 + c] : 0.f;
            G[t][c] = real ? a.gt[((long)b * J + t - off) * 3 + c] : 0.f;
            Gb[t][c] = 0.f;
        }
    }
    __syncthreads();
    float dist = 0.f, cosv = 0.f, dm[3] = {0.f, 0.f, 0.f};
    const float inv_bj = 1.0f / ((float)a.B * J), inv_b = 1.0f / (float)a.B;
    if (t >= off && t < JJ) {            // MPJPE term of joint t
        const float dx = P[t][0] - G[t][0], dy = P[t][1] - G[t][1], dz = P[t][2] - G[t][2];
        dist = sqrtf(dx * dx + dy * dy + dz * dz);
        const float s = dist > 0.f ? a.lam_pose * inv_bj / dist : 0.f;
        dm[0] = s * dx; dm[1] = s * dy; dm[2] = s * dz;
    }
    // bones: child t, parent parents[t], t = 1..JJ-1; EgoCap drops bone t = 1
    if (t >= 1 + off && t < JJ) {
        const int p = a.parents[t];
        float bp[3], bg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { bp[c] = P[t][c] - P[p][c]; bg[c] = G[t][c] - G[p][c]; }
        const float np = sqrtf(bp[0] * bp[0] + bp[1] * bp[1] + bp[2] * bp[2]);
        const float ng = sqrtf(bg[0] * bg[0] + bg[1] * bg[1] + bg[2] * bg[2]);
        const float cp = fmaxf(np, 1e-8f), cg = fmaxf(ng, 1e-8f);
        const float dot = bp[0] * bg[0] + bp[1] * bg[1] + bp[2] * bg[2];
        cosv = dot / (cp * cg);
        // d cos / d bp = bg / (cp cg) - dot * bp / (cp^3 cg)   (a clamped norm has zero derivative)
        const float k = a.lam_cos * inv_b;
#pragma unroll
        for (int c = 0; c < 3; ++c) Gb[t][c] = k * (bg[c] / (cp * cg) - (np > 1e-8f ? dot * bp[c] / (cp * cp * cp * cg) : 0.f));
    }
    __syncthreads();
    // gather, fixed order: own distance term + own bone (child side) - bones of the children in index order
    if (t >= off && t < JJ) {
        float d[3] = {Gb[t][0], Gb[t][1], Gb[t][2]};
        for (int ch = 1 + off; ch < JJ; ++ch)
            if (a.parents[ch] == t) { d[0] -= Gb[ch][0]; d[1] -= Gb[ch][1]; d[2] -= Gb[ch][2]; }
        // plane 0: d loss_pose / d pred, plane 1: d loss_cos_sim / d pred (the caller weighs them with the two upstream gradients)
        const long plane = (long)a.B * J * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.dpred[((long)b * J + t - off) * 3 + c] = dm[c];
            a.dpred[plane + ((long)b * J + t - off) * 3 + c] = d[c];
        }
    }
    dist = wave_sum(dist);
    cosv = wave_sum(cosv);
    if (t == 0) { a.partial[2 * b] = dist; a.partial[2 * b + 1] = cosv; }
}

static __global__ void pose_loss_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int J, float lam_pose,
                                        float lam_cos) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sd = 0.f, sc = 0.f;
    for (int b = 0; b < B; ++b) { sd += partial[2 * b]; sc += partial[2 * b + 1]; }
    out[0] = lam_pose * sd / ((float)B * J);     // loss_pose
    out[1] = lam_cos * sc / (float)B;            // loss_cos_sim
}

// ----------------------------------------------------------------------------- AdamW (torch.optim.AdamW semantics)
// p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
static __global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
}

// the same update for MANY tensors in one launch: gradients, exp_avg and exp_avg_sq live in flat arenas with one layout (segment s
// covers arena elements [off_s, off_s + n_s)), the parameters stay where the framework allocated them.  table[s] = {off_s, n_s, param
// pointer}, sorted by offset.
// [r5] One thread = FOUR consecutive arena elements (segment offsets are multiples of 4 in both wrappers' arenas: 16-byte loads and stores
// wherever the four lie inside one segment and the parameter pointer is 16-byte aligned, element by element otherwise), and the segment is
// found once per WAVE on the scalar unit (binary search on the wave's first element) and then walked forward per lane.  The round-1 form -- one
// element per thread, a 7-step binary search of global loads in front of every 28 bytes of traffic -- ran at 2.2 TB/s: 1.17 ms for the head's 97 M
// parameters, 6 % of a training step at the reference's batch of 32.  Same expressions per element: same bits.
static __global__ __launch_bounds__(256) void adamw_multi_kernel(const long* __restrict__ table, int nseg, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long span, float lr, float b1,
                                                          float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    // wave-uniform start: the first lane's element (lanes are consecutive)
    const long w0 = ((long)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)i0);
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[3 * mid] <= w0) lo = mid; else hi = mid - 1;
    }
    if (i0 >= span) return;
    int sg = lo;
    while (sg + 1 < nseg && table[3 * (sg + 1)] <= i0) ++sg;   // (a wave spans 1024 elements: a few steps where tiny tensors follow each other)
    auto upd = [&](float* p, long i) __attribute__((always_inline)) {
        const float gi = g[i];
        float pi = *p * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        *p = pi;
    };
    const long off = table[3 * sg], n = table[3 * sg + 1];
    float* pb = (float*)table[3 * sg + 2];
    if (i0 >= off && i0 + 4 <= off + n && (((size_t)(pb + (i0 - off))) & 15) == 0 && i0 + 4 <= span) {
        float* p = pb + (i0 - off);
        const f32x4 g4 = *(const f32x4*)(g + i0), m4 = *(const f32x4*)(m + i0), v4 = *(const f32x4*)(v + i0);
        f32x4 p4 = *(const f32x4*)p, mo, vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pi = p4[e] * (1.0f - lr * wd);
            const float mi = b1 * m4[e] + (1.0f - b1) * g4[e];
            const float vi = b2 * v4[e] + (1.0f - b2) * g4[e] * g4[e];
            mo[e] = mi;
            vo[e] = vi;
            pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
            p4[e] = pi;
        }
        *(f32x4*)(m + i0) = mo;
        *(f32x4*)(v + i0) = vo;
        *(f32x4*)p = p4;
        return;
    }
    for (int e = 0; e < 4; ++e) {                               // a segment's ragged end, alignment padding, or the start of the next segment
        const long i = i0 + e;
        if (i >= span) return;
        while (sg + 1 < nseg && table[3 * (sg + 1)] <= i) ++sg;
        const long o2 = table[3 * sg], n2 = table[3 * sg + 1];
        if (i >= o2 && i < o2 + n2) upd((float*)table[3 * sg + 2] + (i - o2), i);
    }
}

// ----------------------------------------------------------------------------- LayerNorm (training)
// forward that also stores mean / rstd per row
template <int D>
__global__ __launch_bounds__(256) void layernorm_fwd_stats_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                                  float eps) {
    constexpr int V = D / 256;
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const float* x = X + (long)r * D;
    f32x4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        v[i] = *(const f32x4*)(x + (i * 64 + lane) * 4);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float d = v[i][c] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    float* y = Y + (long)r * D;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const f32x4 g = *(const f32x4*)(gamma + (i * 64 + lane) * 4), b = *(const f32x4*)(beta + (i * 64 + lane) * 4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mu) * rs * g[c] + b[c];
        *(f32x4*)(y + (i * 64 + lane) * 4) = o;
    }
}

// backward: dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres); partial column sums of dy*xhat, dy per block.
// One wave per row, ROWS_PER_BLOCK rows per wave-slot; part[block][2][D].
template <int D>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dX, float* __restrict__ part, int rows,
                                                            int rows_per_wave) {
    constexpr int V = D / 256;
    __shared__ f32x4 red[2][4][V][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + w;
    f32x4 g[V], dg[V], db[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        g[i] = *(const f32x4*)(gamma + (i * 64 + lane) * 4);
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int r_lo = wave * rows_per_wave, r_hi = min(rows, r_lo + rows_per_wave);
    for (int r = r_lo; r < r_hi; ++r) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[V], dy[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const f32x4 x = *(const f32x4*)(X + (long)r * D + (i * 64 + lane) * 4);
            dy[i] = *(const f32x4*)(dY + (long)r * D + (i * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xh[i][c] = (x[c] - mu) * rs;
                const float gd = g[i][c] * dy[i][c];
                s1 += gd;
                s2 += gd * xh[i][c];
                dg[i][c] += dy[i][c] * xh[i][c];
                db[i][c] += dy[i][c];
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = rs * (g[i][c] * dy[i][c] - m1 - xh[i][c] * m2);
            if (dres) o += *(const f32x4*)(dres + (long)r * D + (i * 64 + lane) * 4);
            *(f32x4*)(dX + (long)r * D + (i * 64 + lane) * 4) = o;
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { red[0][w][i][lane] = dg[i]; red[1][w][i][lane] = db[i]; }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const f32x4 a = (red[0][0][i][lane] + red[0][1][i][lane]) + (red[0][2][i][lane] + red[0][3][i][lane]);
            const f32x4 b = (red[1][0][i][lane] + red[1][1][i][lane]) + (red[1][2][i][lane] + red[1][3][i][lane]);
            *(f32x4*)(part + ((long)blockIdx.x * 2 + 0) * D + (i * 64 + lane) * 4) = a;
            *(f32x4*)(part + ((long)blockIdx.x * 2 + 1) * D + (i * 64 + lane) * 4) = b;
        }
    }
}

// ----------------------------------------------------------------------------- column statistics (BatchNorm1d train)
// part[blockIdx.y][2][C]: sum over the block's rows of f1, f2 with
//   MODE 0: f1 = z,            f2 = (z - mean)^2                 (mean == nullptr on the first pass: f2 = 0)
//   MODE 1: f1 = dyb,          f2 = dyb * xhat,   dyb = dy * (y > 0 ? 1 : slope), xhat = (z - mean) * rstd
template <int MODE>
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ Z, const float* __restrict__ Yv,
                                                       const float* __restrict__ dY, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ part, int R, int C,
                                                       int rows_per_block, float slope) {
    __shared__ f32x4 red[2][4][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + l;
    const int r_lo = blockIdx.y * rows_per_block, r_hi = min(R, r_lo + rows_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < C) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f};
        if (mean) mu = *(const f32x4*)(mean + c4 * 4);
        if (MODE == 1) rs = *(const f32x4*)(rstd + c4 * 4);
        for (int r = r_lo + q; r < r_hi; r += 4) {
            const f32x4 z = *(const f32x4*)(Z + (long)r * C + c4 * 4);
            if (MODE == 0) {
                s1 += z;
                if (mean) { const f32x4 d = z - mu; s2 += d * d; }
            } else {
                const f32x4 y = *(const f32x4*)(Yv + (long)r * C + c4 * 4);
                f32x4 d = *(const f32x4*)(dY + (long)r * C + c4 * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = y[c] > 0.f ? d[c] : slope * d[c];
                s1 += d;
                s2 += d * ((z - mu) * rs);
            }
        }
    }
    red[0][q][l] = s1;
    red[1][q][l] = s2;
    __syncthreads();
    if (q == 0 && c4 * 4 < C) {
        *(f32x4*)(part + ((long)blockIdx.y * 2 + 0) * C + c4 * 4) = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        *(f32x4*)(part + ((long)blockIdx.y * 2 + 1) * C + c4 * 4) = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}

// y = LeakyReLU((z - mean) * rstd * gamma + beta)
static __global__ __launch_bounds__(256) void bn_apply_lrelu_kernel(const float* __restrict__ Z, float* __restrict__ Y,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             long n4, int C, float slope) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 z = *(const f32x4*)(Z + i * 4), mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c);
    const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
    f32x4 y = (z - mu) * rs * g + b;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = y[k] > 0.f ? y[k] : slope * y[k];
    *(f32x4*)(Y + i * 4) = y;
}

// dz = gamma * rstd * (dyb - m1 - xhat * m2),  m1 = mean(dyb), m2 = mean(dyb * xhat)  (sums s1, s2 over R rows)
static __global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ Yv,
                                                           const float* __restrict__ dY, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           float* __restrict__ dZ, long n4, int C, int R, float slope) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 z = *(const f32x4*)(Z + i * 4), y = *(const f32x4*)(Yv + i * 4);
    f32x4 d = *(const f32x4*)(dY + i * 4);
    const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c), g = *(const f32x4*)(gamma + c);
    const f32x4 a1 = *(const f32x4*)(s1 + c), a2 = *(const f32x4*)(s2 + c);
    const float inv = 1.0f / (float)R;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dyb = y[k] > 0.f ? d[k] : slope * d[k];
        const float xh = (z[k] - mu[k]) * rs[k];
        o[k] = g[k] * rs[k] * (dyb - a1[k] * inv - xh * a2[k] * inv);
    }
    *(f32x4*)(dZ + i * 4) = o;
}

// finish of the statistics pass: mean = s1/R ; (second pass) var = s2/R, rstd, running stats (momentum, unbiased var)
static __global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ s, float* __restrict__ mean_or_rstd, int C, int R,
                                                        int stage, float eps, float momentum, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var, const float* __restrict__ mean_in) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (stage == 0) {
        mean_or_rstd[c] = s[c] / (float)R;
    } else {
        const float var = s[c] / (float)R;
        mean_or_rstd[c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean_in[c];
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * ((float)R / (float)(R - 1));
        }
    }
}

// ----------------------------------------------------------------------------- GELU backward (exact erf form)
// dz = dh * (0.5 (1 + erf(z/sqrt2)) + z * exp(-z^2/2) / sqrt(2 pi))
static __global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ Zp, const float* __restrict__ dH, float* __restrict__ dZ,
                                                       long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 z = *(const f32x4*)(Zp + i * 4), d = *(const f32x4*)(dH + i * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = d[k] * dgelu_erf(z[k]);
    *(f32x4*)(dZ + i * 4) = o;
}

// out[i] += in[i] (gradient accumulation of a residual branch), n4 float4
static __global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ out, const float* __restrict__ in, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    *(f32x4*)(out + i * 4) = *(const f32x4*)(out + i * 4) + *(const f32x4*)(in + i * 4);
}
