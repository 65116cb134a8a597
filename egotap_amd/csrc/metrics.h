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
