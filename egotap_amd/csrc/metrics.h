// Evaluation metrics right behind the lifting head: per-sample MPJPE and Procrustes-aligned MPJPE
// (EgoTAPAutoEncoderModel.evaluate, model/egotap_autoencoder_model.py:329-350: a Python loop over the batch around
// utils/util.py:328-379 batch_compute_similarity_transform_torch, which runs a batched 3x3 torch.svd and a per-sample
// trace loop).  Here: one thread per sample, everything in registers, float64 inside.
//   mu1, mu2 = joint means;  Y = X - mu;  var1 = |Y1|^2;  K = Y1 Y2^T (3x3)
//   K = U S V^T,  R = V diag(1, 1, sign det(U V^T)) U^T,  scale = trace(R K) / var1,  t = mu2 - scale R mu1
//   pa_mpjpe = mean_j | X2_j - (scale R X1_j + t) |,   mpjpe = mean_j | X2_j - X1_j |
// The SVD comes from the symmetric eigenproblem K^T K = V S^2 V^T (cyclic Jacobi, 3x3), singular values sorted
// descending so the reflection fix lands on the smallest one, exactly where torch.svd's ordering puts it.
#pragma once
#include "common.h"

#define EGOTAP_MAX_JOINTS 32

__device__ __forceinline__ void jacobi_eig3(double A[3][3], double V[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {      // A <- A J
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {      // A <- J^T A
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {      // V <- V J
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
}

static __global__ __launch_bounds__(64) void pose_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B, int J,
                                                          float* __restrict__ mpjpe, float* __restrict__ pa_mpjpe,
                                                          float* __restrict__ aligned) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* x1 = pred + (long)b * J * 3;
    const float* x2 = gt + (long)b * J * 3;
    double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0}, e = 0.0;
    for (int j = 0; j < J; ++j) {
        double d2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a = x1[3 * j + c], g = x2[3 * j + c];
            mu1[c] += a; mu2[c] += g;
            d2 += (g - a) * (g - a);
        }
        e += sqrt(d2);
    }
    mpjpe[b] = (float)(e / J);
#pragma unroll
    for (int c = 0; c < 3; ++c) { mu1[c] /= J; mu2[c] /= J; }
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
    for (int j = 0; j < J; ++j) {
        double y1[3], y2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { y1[c] = x1[3 * j + c] - mu1[c]; y2[c] = x2[3 * j + c] - mu2[c]; var1 += y1[c] * y1[c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) K[r][c] += y1[r] * y2[c];
    }
    double A[3][3], V[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];   // K^T K
    jacobi_eig3(A, V);
    // sort eigenpairs descending
    double ev[3] = {A[0][0], A[1][1], A[2][2]};
    int ord[3] = {0, 1, 2};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 - i; ++j)
            if (ev[ord[j]] < ev[ord[j + 1]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double Vs[3][3], U[3][3], sv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sv[k] = sqrt(fmax(ev[ord[k]], 0.0));
#pragma unroll
        for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][ord[k]];
    }
    // U columns: u_k = K v_k / s_k, Gram-Schmidt against the earlier ones; a vanishing singular value takes the cross product
    const double tiny = 1e-12 * fmax(sv[0], 1e-300);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) u[r] = K[r][0] * Vs[0][k] + K[r][1] * Vs[1][k] + K[r][2] * Vs[2][k];
        for (int p = 0; p < k; ++p) {
            const double dot = u[0] * U[0][p] + u[1] * U[1][p] + u[2] * U[2][p];
#pragma unroll
            for (int r = 0; r < 3; ++r) u[r] -= dot * U[r][p];
        }
        double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!(sv[k] > tiny) || !(n > tiny * 1e-3)) {
            if (k == 2) {
                u[0] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
                u[1] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
                u[2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
            } else {        // rank <= 1 input (all joints collinear or coincident): any orthonormal completion
                const int ax = fabs(k ? U[0][0] : 0.0) < 0.9 ? 0 : 1;
                u[0] = ax == 0; u[1] = ax == 1; u[2] = 0.0;
                for (int p = 0; p < k; ++p) {
                    const double dot = u[0] * U[0][p] + u[1] * U[1][p] + u[2] * U[2][p];
#pragma unroll
                    for (int r = 0; r < 3; ++r) u[r] -= dot * U[r][p];
                }
            }
            n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) U[r][k] = u[r] / n;
    }
    auto det3 = [](const double M[3][3]) {
        return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
               M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    };
    const double dd = det3(U) * det3(Vs);
    const double z = dd > 0.0 ? 1.0 : (dd < 0.0 ? -1.0 : 0.0);          // torch.sign
    double R[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + z * Vs[r][2] * U[c][2];
    double tr = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) tr += R[r][c] * K[c][r];
    const double scale = tr / var1;
    double t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = mu2[r] - scale * (R[r][0] * mu1[0] + R[r][1] * mu1[1] + R[r][2] * mu1[2]);
    double pa = 0.0;
    for (int j = 0; j < J; ++j) {
        double d2 = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double h = scale * (R[r][0] * x1[3 * j] + R[r][1] * x1[3 * j + 1] + R[r][2] * x1[3 * j + 2]) + t[r];
            if (aligned) aligned[((long)b * J + j) * 3 + r] = (float)h;
            const double d = x2[3 * j + r] - h;
            d2 += d * d;
        }
        pa += sqrt(d2);
    }
    pa_mpjpe[b] = (float)(pa / J);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// What the reference computes for a BATCH of 2 or 3 frames.  utils/util.py:337 tests S1.shape[0] against 3 and 2 (meant for
// unbatched 3 x N / 2 x N point sets) and then skips its transpose: the [B, J, 3] poses are read as J "coordinates" x 3 "points" --
// means over the three columns, K = X1 X2^T a J x J matrix of rank <= 2, R = V Z U^T from its SVD, no transpose back.  test.py and
// train_evaluate print these numbers for a ragged last batch of 2 or 3 frames, so the drop-in reproduces them (default; see
// egotap_pose_metrics_batch_axes in egotap.h).  The J x J SVD reduces exactly to 3 x 3 and 2 x 2 problems: with thin SVDs
// X1 = Ua Sa Va^T, X2 = Ub Sb Vb^T (J x 2, 2 x 2, 3 x 2: the columns are centred, rank <= 2) and M = Sa Va^T Vb Sb = Um Sm Vm^T,
//   K = (Ua Um) Sm (Ub Vm)^T,   scale = (sm_0 + sm_1) / |X1|^2     (the sign fix Z lands on a zero singular value),
//   S1_hat = scale (Ub Vm Um^T Ua^T) X1 + mu2 1^T = scale X2 W + mu2 1^T,   W = Vb Sb^-1 (Vm Um^T) Sa Va^T   (3 x 3)
// (the null-space terms of R S1 and of t = mu2 - scale R mu1 cancel).  Checked against the reference's own outputs
// (tests/golden/procrustes_batch_axes.npz) and against the J x J restatement in oracle/lift_ref.py.
__device__ __forceinline__ void top2_eig3(double G[3][3], double V2[3][2], double s2[2]) {
    double V[3][3];
    jacobi_eig3(G, V);
    const double ev[3] = {G[0][0], G[1][1], G[2][2]};
    int o0 = 0, o1 = 1, o2 = 2;
    if (ev[o0] < ev[o1]) { const int t = o0; o0 = o1; o1 = t; }
    if (ev[o1] < ev[o2]) { const int t = o1; o1 = o2; o2 = t; }
    if (ev[o0] < ev[o1]) { const int t = o0; o0 = o1; o1 = t; }
    s2[0] = sqrt(fmax(ev[o0], 0.0));
    s2[1] = sqrt(fmax(ev[o1], 0.0));
#pragma unroll
    for (int r = 0; r < 3; ++r) { V2[r][0] = V[r][o0]; V2[r][1] = V[r][o1]; }
}

static __global__ __launch_bounds__(64) void pose_metrics_batch_axes_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B, int J,
                                                                     float* __restrict__ mpjpe, float* __restrict__ pa_mpjpe,
                                                                     float* __restrict__ aligned) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* x1 = pred + (long)b * J * 3;
    const float* x2 = gt + (long)b * J * 3;
    double G1[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, G2[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0, e = 0.0;
    for (int j = 0; j < J; ++j) {
        double a[3], g[3], d2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] = x1[3 * j + c]; g[c] = x2[3 * j + c]; d2 += (g[c] - a[c]) * (g[c] - a[c]); }
        e += sqrt(d2);
        const double m1 = (a[0] + a[1] + a[2]) / 3.0, m2 = (g[0] + g[1] + g[2]) / 3.0;      // the "mean point" of row j: over the 3 columns
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] -= m1; g[c] -= m2; var1 += a[c] * a[c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { G1[r][c] += a[r] * a[c]; G2[r][c] += g[r] * g[c]; }
    }
    mpjpe[b] = (float)(e / J);
    double Va[3][2], Vb[3][2], sa[2], sb[2];
    top2_eig3(G1, Va, sa);
    top2_eig3(G2, Vb, sb);
    double M[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) M[i][k] = sa[i] * (Va[0][i] * Vb[0][k] + Va[1][i] * Vb[1][k] + Va[2][i] * Vb[2][k]) * sb[k];
    // 2 x 2 SVD through the eigenvectors of M^T M: v_i, s_i = |M v_i|, u_i = M v_i / s_i;  T = sum_i v_i u_i^T = Vm Um^T
    const double p = M[0][0] * M[0][0] + M[1][0] * M[1][0], q = M[0][1] * M[0][1] + M[1][1] * M[1][1], r_ = M[0][0] * M[0][1] + M[1][0] * M[1][1];
    const double th = 0.5 * atan2(2.0 * r_, p - q), cs = cos(th), sn = sin(th);
    const double vm[2][2] = {{cs, sn}, {-sn, cs}};               // vm[i] = i-th right singular vector
    double T[2][2] = {{0, 0}, {0, 0}}, ssum = 0.0;
    const double tiny = 1e-14 * fmax(sa[0] * sb[0], 1e-300);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double u0 = M[0][0] * vm[i][0] + M[0][1] * vm[i][1], u1 = M[1][0] * vm[i][0] + M[1][1] * vm[i][1];
        const double s = sqrt(u0 * u0 + u1 * u1);
        if (s > tiny) {
            ssum += s;
            T[0][0] += vm[i][0] * u0 / s; T[0][1] += vm[i][0] * u1 / s;
            T[1][0] += vm[i][1] * u0 / s; T[1][1] += vm[i][1] * u1 / s;
        }
    }
    const double scale = ssum / var1;
    // W = Vb Sb^-1 T Sa Va^T
    double W[3][3];
    const double isb[2] = {sb[0] > tiny ? 1.0 / sb[0] : 0.0, sb[1] > tiny ? 1.0 / sb[1] : 0.0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double w = 0.0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 2; ++k) w += Vb[r][i] * isb[i] * T[i][k] * sa[k] * Va[c][k];
            W[r][c] = w;
        }
    double pa = 0.0;
    for (int j = 0; j < J; ++j) {
        const double g0 = x2[3 * j], g1 = x2[3 * j + 1], g2 = x2[3 * j + 2];
        const double m2 = (g0 + g1 + g2) / 3.0;
        const double y[3] = {g0 - m2, g1 - m2, g2 - m2}, gg[3] = {g0, g1, g2};
        double d2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double h = scale * (y[0] * W[0][c] + y[1] * W[1][c] + y[2] * W[2][c]) + m2;
            if (aligned) aligned[((long)b * J + j) * 3 + c] = (float)h;
            const double d = gg[c] - h;
            d2 += d * d;
        }
        pa += sqrt(d2);
    }
    pa_mpjpe[b] = (float)(pa / J);
}
