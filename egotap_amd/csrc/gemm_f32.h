// fp32 GEMM on the gfx950 matrix cores:  C[M,N] = epi( A[M,K] * W[N,K]^T ).
//
// * v_mfma_f32_32x32x2_f32 (exact f32, one rounding per product, 64 FLOP/clk/SIMD;
//   157.3 TFLOP/s chip peak) -- the north star pins fp32 at 1e-4, so no bf16.
// * A is produced by a LOADER functor, so the reference's reshuffles (tiling the
//   heatmaps into the ViT image, regrouping tokens per heatmap, the stereo
//   cos/sin interleave) are index math inside the load, never a copy.
// * W is nn.Linear's [N,K] layout, up to three row segments (fused Q|K|V).
// * The epilogue functor fuses bias / residual / exact GELU / folded
//   BatchNorm1d + LeakyReLU / position-embedding + mask-token.
//
// Tiling: BM x BN block, BK-deep slabs double buffered in LDS (rows padded by 4
// floats so the ds_read_b128 fragment reads are bank-conflict free), register
// staged prefetch of the next slab under the MFMAs of the current one, one
// barrier per slab.  Each wave owns a (BM/WM) x (BN/WN) sub-tile = TM x TN MFMA
// tiles of 32x32.  The MFMA k index is permuted (lane half h, step 4t+u reads
// k = 8t+4h+u for both operands) so every fragment read is one 16-byte LDS read
// feeding four MFMAs.
#pragma once
#include "common.h"
#include <type_traits>

// ----------------------------------------------------------------------------- A loaders
struct ALoadPlain {
    const float* A;
    long lda;
    struct Row { const float* p; };
    __device__ __forceinline__ Row row(int m) const { return Row{A + (long)m * lda}; }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const { return *(const f32x4*)(r.p + k); }
    // address of the 4 floats load() returns: loaders that are pure address math can feed the global -> LDS DMA (gemm_f32_dma.h)
    static constexpr bool HAS_PTR = true;
    __device__ __forceinline__ const float* ptr(const Row& r, int k) const { return r.p + k; }
    __host__ bool dma_ok() const { return lda % 4 == 0 && lda < (1L << 21) && ((uintptr_t)A & 15) == 0; }
};

// ViT patch embedding input: token (b, pr, pc) of the tiled (grid*hm)^2 image, K = 16*16
// pixels of one patch.  Reference: net_architecture.py:375-383 + modeling_vit.py:195.
struct ALoadPatch {
    const float* hm;   // [B, C, S, S] heatmaps, position channels first
    int C, S, seq, side, ppd, grid, T;
    // [r3] zeros != nullptr: 256 zero floats (16-byte aligned) that stand in for the dummy cells, so that the loader is pure address math and can
    // feed the global -> LDS DMA (gemm_f32_dma.h); nullptr (the default): the register-staged kernels, as before
    const float* zeros = nullptr;
    struct Row { const float* p; };   // nullptr: dummy cell (zeros)
    static constexpr bool HAS_PTR = true;
    __device__ __forceinline__ const float* ptr(const Row& r, int k) const { return r.p ? r.p + (k >> 4) * S + (k & 15) : zeros + k; }
    __host__ bool dma_ok() const { return zeros != nullptr && S % 4 == 0 && ((uintptr_t)hm & 15) == 0 && ((uintptr_t)zeros & 15) == 0; }
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / seq, tok = m - b * seq;
        const int pr = tok / side, pc = tok - pr * side;
        const int cell = (pr / ppd) * grid + pc / ppd;
        if (cell >= T) return Row{nullptr};
        return Row{hm + ((long)(b * C + cell) * S + (pr % ppd) * 16) * S + (pc % ppd) * 16};
    }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const {
        if (r.p == nullptr) return f32x4{0.f, 0.f, 0.f, 0.f};
        return *(const f32x4*)(r.p + (k >> 4) * S + (k & 15));
    }
};

// fc1 of the position encoder: row (b, i) = the ppd x ppd patch tokens of heatmap i,
// flattened (patch-row, patch-col, channel).  Reference: net_architecture.py:388-406.
struct ALoadTokens {
    const float* Y;    // [B*seq, D] final-LayerNorm tokens
    int T, D, seq, side, ppd, grid;
    struct Row { const float* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / T, i = m - b * T;
        return Row{Y + ((long)b * seq + (long)(ppd * (i / grid)) * side + ppd * (i % grid)) * D};
    }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const {
        const int s = k / D, c = k - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        return *(const f32x4*)(r.p + (long)(prl * side + pcl) * D + c);
    }
    static constexpr bool HAS_PTR = true;
    __device__ __forceinline__ const float* ptr(const Row& r, int k) const {
        const int s = k / D, c = k - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        return r.p + (long)(prl * side + pcl) * D + c;
    }
    __host__ bool dma_ok() const { return D % 4 == 0 && ((uintptr_t)Y & 15) == 0; }
};

// fc1 of the rotation encoder: row (b, eye*J + j) = [cos map | sin map] of limb j of that eye.
// Reference: net_architecture.py:690-694 (channel order L_cos, L_sin, R_cos, R_sin after 2J position maps).
struct ALoadRot {
    const float* hm;   // [B, C, S, S]
    int C, J, HW;      // HW = S*S
    struct Row { const float* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int T = 2 * J;
        const int b = m / T, t = m - b * T;
        const int eye = t / J, j = t - eye * J;
        return Row{hm + (long)(b * C + 2 * J + eye * 2 * J + j) * HW};
    }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const {
        const int cs = k / HW;
        return *(const f32x4*)(r.p + (long)cs * J * HW + (k - cs * HW));
    }
    static constexpr bool HAS_PTR = true;
    __device__ __forceinline__ const float* ptr(const Row& r, int k) const {
        const int cs = k / HW;
        return r.p + (long)cs * J * HW + (k - cs * HW);
    }
    __host__ bool dma_ok() const { return HW % 4 == 0 && ((uintptr_t)hm & 15) == 0; }
};

// Propagation-unit inputs, time-major rows m = t*B + b: [left_t | right_t] features of
// joint t from Z[(b*2 + eye)*J + t, hid].  Reference: net_architecture.py:699-705, 722-723.
struct ALoadStereo {
    static constexpr bool HAS_PTR = false;      // small PU GEMMs only
    const float* Z;
    int B, J, hid;
    struct Row { const float* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int t = m / B, b = m - t * B;
        return Row{Z + ((long)b * 2 * J + t) * hid};
    }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const {
        const int eye = k / hid;
        return *(const f32x4*)(r.p + (long)eye * J * hid + (k - eye * hid));
    }
};

// Bridge operand of PU layer 0: b' = sigmoid(F[m, fcol0 + k]) * bridge(m, k)   (custom_cells.py:102).
struct ALoadStereoGated {
    static constexpr bool HAS_PTR = false;      // the gate is arithmetic on the operand
    ALoadStereo z;
    const float* F;    // [J*B, ldf] x2f output
    int ldf, fcol0;
    struct Row { ALoadStereo::Row r; const float* f; };
    __device__ __forceinline__ Row row(int m) const { return Row{z.row(m), F + (long)m * ldf + fcol0}; }
    __device__ __forceinline__ f32x4 load(const Row& r, int k) const {
        f32x4 v = z.load(r.r, k);
        const f32x4 g = *(const f32x4*)(r.f + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= 1.0f / (1.0f + expf(-g[i]));
        return v;
    }
};

// ----------------------------------------------------------------------------- epilogues
struct EpiBias {          // C = acc + bias
    SegVec bias;
    struct Col { float b; };
    __device__ __forceinline__ Col col(int n) const { return Col{bias.at(n)}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const { return acc + c.b; }
    struct Col4 { f32x4 b; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{bias.at4(n0)}; }
    static constexpr bool HAS_RES = false;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const { return acc + c.b; }
};
struct EpiBiasRes {       // C = acc + bias + R[m, n]   (R may alias C)
    SegVec bias;
    const float* R;
    long ldr;
    struct Col { float b; };
    __device__ __forceinline__ Col col(int n) const { return Col{bias.at(n)}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const {
        return acc + c.b + R[(long)m * ldr + n];
    }
    struct Col4 { f32x4 b; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{bias.at4(n0)}; }
    static constexpr bool HAS_RES = true;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return *(const f32x4*)(R + (long)m * ldr + n0); }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const { return acc + c.b + res; }
};
struct EpiBiasGelu {      // exact erf GELU (modeling_vit.py:320-327, hidden_act='gelu')
    SegVec bias;
    struct Col { float b; };
    __device__ __forceinline__ Col col(int n) const { return Col{bias.at(n)}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const {
        const float x = acc + c.b;
        return gelu_erf(x);
    }
    struct Col4 { f32x4 b; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{bias.at4(n0)}; }
    static constexpr bool HAS_RES = false;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = acc[i] + c.b[i];
            o[i] = gelu_erf(x);
        }
        return o;
    }
};
struct EpiBnLrelu {       // LeakyReLU_0.2(BatchNorm1d_eval(acc + bias))  (network_utils.py:123-142)
    const float *bias, *gamma, *beta, *mean, *var;
    float eps, slope;
    struct Col { float b, mu, sc, sh; };
    __device__ __forceinline__ Col col(int n) const {
        return Col{bias[n], mean[n], gamma[n] / sqrtf(var[n] + eps), beta[n]};
    }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const {
        const float y = (acc + c.b - c.mu) * c.sc + c.sh;
        return y > 0.f ? y : slope * y;
    }
    struct Col4 { Col c[4]; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{{col(n0), col(n0 + 1), col(n0 + 2), col(n0 + 3)}}; }
    static constexpr bool HAS_RES = false;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = apply(acc[i], c.c[i], m, n0 + i);
        return o;
    }
};
struct EpiPatch {         // (dummy ? mask_token : acc + bias) + position_embeddings   (modeling_vit.py:137-153)
    const float *bias, *mask_tok, *pos;
    int D, seq, side, ppd, grid, T;
    struct Col { float b, mt; };
    __device__ __forceinline__ Col col(int n) const { return Col{bias[n], mask_tok[n]}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const {
        const int tok = m % seq;
        const int pr = tok / side, pc = tok - pr * side;
        const bool dummy = (pr / ppd) * grid + pc / ppd >= T;
        return (dummy ? c.mt : acc + c.b) + pos[(long)tok * D + n];
    }
    struct Col4 { f32x4 b, mt; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{*(const f32x4*)(bias + n0), *(const f32x4*)(mask_tok + n0)}; }
    static constexpr bool HAS_RES = true;     // the position embedding row plays the residual's role
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return *(const f32x4*)(pos + (long)(m % seq) * D + n0); }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const {
        const int tok = m % seq;
        const int pr = tok / side, pc = tok - pr * side;
        const bool dummy = (pr / ppd) * grid + pc / ppd >= T;
        return (dummy ? c.mt : acc + c.b) + res;
    }
};

// ---- epilogues used by the training step -------------------------------------------------------------------
struct EpiNone {          // C = acc (input-gradient GEMMs)
    struct Col {};
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const { return acc; }
    struct Col4 {};
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{}; }
    static constexpr bool HAS_RES = false;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const { return acc; }
};
struct EpiAccum {         // C = acc + R[m, n]  (R may alias C): gradient accumulation of a second branch
    const float* R;
    long ldr;
    struct Col {};
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const { return acc + R[(long)m * ldr + n]; }
    struct Col4 {};
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{}; }
    static constexpr bool HAS_RES = true;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return *(const f32x4*)(R + (long)m * ldr + n0); }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const { return acc + res; }
};
struct EpiBiasGeluSave {  // z = acc + bias is stored to Z (kept for the backward), C = GELU(z)
    SegVec bias;
    float* Z;
    long ldz;
    struct Col { float b; };
    __device__ __forceinline__ Col col(int n) const { return Col{bias.at(n)}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const {
        const float x = acc + c.b;
        Z[(long)m * ldz + n] = x;
        return gelu_erf(x);
    }
    struct Col4 { f32x4 b; };
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{bias.at4(n0)}; }
    static constexpr bool HAS_RES = false;
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const {
        const f32x4 x = acc + c.b;
        *(f32x4*)(Z + (long)m * ldz + n0) = x;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = gelu_erf(x[i]);
        return o;
    }
};
struct EpiGeluGrad {      // C = acc * gelu'(Z[m, n])   (input gradient of the MLP-down layer, fused with GELU backward)
    const float* Z;
    long ldz;
    __device__ __forceinline__ static float dgelu(float z) {
        return dgelu_erf(z);
    }
    struct Col {};
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ float apply(float acc, const Col& c, int m, int n) const { return acc * dgelu(Z[(long)m * ldz + n]); }
    struct Col4 {};
    __device__ __forceinline__ Col4 col4(int n0) const { return Col4{}; }
    static constexpr bool HAS_RES = true;      // Z rows are fetched like a residual
    __device__ __forceinline__ f32x4 res4(int m, int n0) const { return *(const f32x4*)(Z + (long)m * ldz + n0); }
    __device__ __forceinline__ f32x4 apply4(f32x4 acc, const Col4& c, f32x4 res, int m, int n0) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = acc[i] * dgelu(res[i]);
        return o;
    }
};

// ----------------------------------------------------------------------------- kernel
template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, MINW = MINW_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int LDK = BK + 4;                 // padded LDS row (floats)
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int A_V4 = BM * BK / 4 / THREADS; // float4 per thread per slab
    static constexpr int B_V4 = BN * BK / 4 / THREADS;
    static constexpr int ROWS_PER_PASS = THREADS / (BK / 4);
    static constexpr int LDS_BYTES = 2 * (BM + BN) * LDK * 4;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tile must be a multiple of 32x32");
    static_assert(BK % 8 == 0, "BK multiple of 8");
    static_assert((BM * BK / 4) % THREADS == 0 && (BN * BK / 4) % THREADS == 0, "staging must divide evenly");
};

template <class Cfg, class ALoad, class Epi>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void gemm_f32_kernel(
    ALoad al, SegMat W, Epi epi, float* C, long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4, RPP = Cfg::ROWS_PER_PASS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                        // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;         // [2][BN][LDK]

    int tm, tn;
    xcd_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, 8, tm, tn);
    const int bm = tm * BM, bn = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // staging assignment: thread -> (row r0 + i*RPP, float4 column c4)
    const int c4 = tid % (BK / 4), r0 = tid / (BK / 4);
    typename ALoad::Row arow[A_V4];
    const float* brow[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) arow[i] = al.row(min(bm + r0 + i * RPP, M - 1));
#pragma unroll
    for (int i = 0; i < B_V4; ++i) brow[i] = W.row(bn + r0 + i * RPP);

    // [r4] global loads run TWO slabs ahead in two register sets (slab j: set j & 1 from its request to its LDS write): with one set a slab's
    // loads had the length of one slab's MFMAs to arrive, which a workgroup alone on its CU (serving batches: one wave per SIMD, 1.7 us per
    // 128 x 128 slab, 0.85 for 64 x 128) does not cover -- a 64-row tile ran at 1.5 us per slab
    f32x4 pa0[A_V4], pb0[B_V4], pa1[A_V4], pb1[B_V4];
    auto gload = [&](f32x4 (&pa)[A_V4], f32x4 (&pb)[B_V4], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) pa[i] = al.load(arow[i], k0 + c4 * 4);
#pragma unroll
        for (int i = 0; i < B_V4; ++i) pb[i] = *(const f32x4*)(brow[i] + k0 + c4 * 4);
    };
    auto lstore = [&](const f32x4 (&pa)[A_V4], const f32x4 (&pb)[B_V4], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) *(f32x4*)(As + (buf * BM + r0 + i * RPP) * LDK + c4 * 4) = pa[i];
#pragma unroll
        for (int i = 0; i < B_V4; ++i) *(f32x4*)(Bs + (buf * BN + r0 + i * RPP) * LDK + c4 * 4) = pb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* Ab = As + (buf * BM + wm * (TM * 32) + l31) * LDK + 4 * lh;
        const float* Bb = Bs + (buf * BN + wn * (TN * 32) + l31) * LDK + 4 * lh;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(Ab + i * 32 * LDK + 8 * t);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bb + j * 32 * LDK + 8 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
        }
    };
    const int KT = K / BK;
    constexpr bool DEEP = TM * TN <= 4 && A_V4 + B_V4 <= 8;     // (the wide tiles of the sweep tool have no registers for a second set: one slab ahead)
    if constexpr (DEEP) {
        gload(pa0, pb0, 0);
        lstore(pa0, pb0, 0);
        if (KT > 1) gload(pa1, pb1, BK);
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 2 < KT) gload(pa0, pb0, (kt + 2) * BK);
            compute(0);
            if (kt + 1 < KT) lstore(pa1, pb1, 1);
            __syncthreads();
            if (kt + 1 >= KT) break;
            if (kt + 3 < KT) gload(pa1, pb1, (kt + 3) * BK);
            compute(1);
            if (kt + 2 < KT) lstore(pa0, pb0, 0);
            __syncthreads();
        }
    } else {
        gload(pa0, pb0, 0);
        lstore(pa0, pb0, 0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < KT) gload(pa0, pb0, (kt + 1) * BK);
            compute(buf);
            if (kt + 1 < KT) lstore(pa0, pb0, buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: acc reg r of lane l is C[32x32 tile row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn + wn * (TN * 32) + j * 32 + l31;
        const typename Epi::Col cc = epi.col(n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mbase = bm + wm * (TM * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < M) C[(long)m * ldc + n] = epi.apply(acc[i][j][r], cc, m, n);
            }
        }
    }
}

// ----------------------------------------------------------------------------- pipelined kernel
// Same tiling and numerics as gemm_f32_kernel (identical k order, bit-identical results) but with
// the per-slab bubbles removed, which the first version lost ~22 % of the MFMA pipe to
// (SQ_VALU_MFMA_BUSY_CYCLES 0.78; its two co-resident waves per SIMD run the same program in
// lockstep, so their ds_read / ds_write / barrier gaps coincide):
//   * THREE LDS slabs: slab kt+1 is already published while slab kt is computed, so the first
//     fragments of the next slab are read BEFORE the barrier, which drops off the critical path;
//   * fragments are double buffered in registers: the ds_reads of group t+1 are issued ahead of
//     the 4*TM*TN MFMAs of group t (sched_barrier keeps the compiler from sinking them);
//   * global loads run two slabs ahead in two register sets (a slab is only BK=16 deep, so a
//     set is 4 float4), the LDS write of slab kt+2 sits in the middle of slab kt's MFMAs.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_>
struct PipeCfg : GemmCfg<BM_, BN_, BK_, WM_, WN_, MINW_> {
    using Base = GemmCfg<BM_, BN_, BK_, WM_, WN_, MINW_>;
    static constexpr int NS = 3;
    static constexpr int LDS_BYTES = NS * (BM_ + BN_) * Base::LDK * 4;
    static_assert((BK_ / 8) % 2 == 0, "need an even number of 8-deep fragment groups per slab");
};

template <class Cfg, class ALoad, class Epi>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void gemm_f32_pipe_kernel(
    ALoad al, SegMat W, Epi epi, float* C, long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK, NS = Cfg::NS;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4, RPP = Cfg::ROWS_PER_PASS;
    constexpr int G = BK / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [NS][BM][LDK]
    float* Bs = smem + NS * BM * LDK;         // [NS][BN][LDK]

    int tm, tn;
    xcd_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, 8, tm, tn);
    const int bm = tm * BM, bn = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int c4 = tid % (BK / 4), r0 = tid / (BK / 4);
    typename ALoad::Row arow[A_V4];
    const float* brow[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) arow[i] = al.row(min(bm + r0 + i * RPP, M - 1));
#pragma unroll
    for (int i = 0; i < B_V4; ++i) brow[i] = W.row(bn + r0 + i * RPP);

    f32x4 ga[2][A_V4], gb[2][B_V4];          // two global-load register sets
    f32x4 fa[2][TM], fb[2][TN];              // two fragment register sets
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = K / BK;
    const int a_off = (wm * (TM * 32) + l31) * LDK + 4 * lh;
    const int b_off = (wn * (TN * 32) + l31) * LDK + 4 * lh;

#define GLOAD(SET, SLAB)                                                                             \
    {                                                                                                \
        const int k0_ = (SLAB) * BK + c4 * 4;                                                        \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i) ga[SET][i] = al.load(arow[i], k0_);         \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i) gb[SET][i] = *(const f32x4*)(brow[i] + k0_); \
    }
#define LSTORE(SET, BUF)                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i)                                                         \
            *(f32x4*)(As + ((BUF) * BM + r0 + i * RPP) * LDK + c4 * 4) = ga[SET][i];                             \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i)                                                         \
            *(f32x4*)(Bs + ((BUF) * BN + r0 + i * RPP) * LDK + c4 * 4) = gb[SET][i];                             \
    }
#define FRAGS(FSET, BUF, T)                                                                                      \
    {                                                                                                            \
        const float* ap_ = As + (BUF) * BM * LDK + a_off + 8 * (T);                                              \
        const float* bp_ = Bs + (BUF) * BN * LDK + b_off + 8 * (T);                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[FSET][i] = *(const f32x4*)(ap_ + i * 32 * LDK);        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[FSET][j] = *(const f32x4*)(bp_ + j * 32 * LDK);        \
    }
#define MFMAS(FSET)                                                                                              \
    {                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[FSET][i][u], fb[FSET][j][u], acc[i][j], 0, 0, 0); \
    }

    // prologue: slabs 0 and 1 into LDS, slab 2 in flight, first fragments of slab 0 in registers
    GLOAD(0, 0)
    if (KT > 1) GLOAD(1, 1)
    LSTORE(0, 0)
    if (KT > 2) GLOAD(0, 2)
    if (KT > 1) LSTORE(1, 1)
    __syncthreads();
    FRAGS(0, 0, 0)

    int buf = 0;   // LDS slab of iteration kt
    auto body = [&](auto PT, int kt) {
        constexpr int P = decltype(PT)::value;      // parity of kt: register set of slab kt+2 (same parity) is P
        const int b1 = buf + 1 >= NS ? buf + 1 - NS : buf + 1;
        const int b2 = b1 + 1 >= NS ? b1 + 1 - NS : b1 + 1;
        if (kt + 3 < KT) GLOAD(P ^ 1, kt + 3)
#pragma unroll
        for (int t = 0; t < G; ++t) {
            // prefetch the fragments of the next group (next slab after the last group)
            if (t + 1 < G) {
                FRAGS((t + 1) & 1, buf, t + 1)
            } else if (kt + 1 < KT) {
                FRAGS(0, b1, 0)
            }
            __builtin_amdgcn_sched_barrier(0);
            MFMAS(t & 1)
            __builtin_amdgcn_sched_barrier(0);
            if (t == G / 2 - 1 && kt + 2 < KT) {
                LSTORE(P, b2)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf = b1;
    };
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
        body(std::integral_constant<int, 0>{}, kt);
        body(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < KT) body(std::integral_constant<int, 0>{}, kt);
#undef GLOAD
#undef LSTORE
#undef FRAGS
#undef MFMAS

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn + wn * (TN * 32) + j * 32 + l31;
        const typename Epi::Col cc = epi.col(n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mbase = bm + wm * (TM * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < M) C[(long)m * ldc + n] = epi.apply(acc[i][j][r], cc, m, n);
            }
        }
    }
}

template <class Cfg, class ALoad, class Epi>
static hipError_t gemm_f32_pipe_launch(const ALoad& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N,
                                       int K, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0) return hipErrorInvalidValue;
    auto kern = gemm_f32_pipe_kernel<Cfg, ALoad, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, al, W, epi, C, ldc, M,
                       N, K, tiles_m, tiles_n);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------- persistent kernel
// gemm_f32_pipe_kernel's slab pipeline run as ONE block per CU that walks its share of the tiles: the slab stream
// simply continues into the next tile (its first slabs are loaded / staged under the tail of the current one), so
// the per-tile prologue (exposed first-slab latency, ~3 slab times) disappears, and the C tile is stored as whole
// 128-byte row segments with global_store_dwordx4 after a per-wave transpose through a private LDS patch
// (4x fewer store instructions than the accumulator-shaped dword stores; a vector-memory instruction costs the
// matrix pipe ~56 cycles).  Tile order: blocks with equal blockIdx % 8 share an L2; they take consecutive tiles of a
// contiguous chunk of the grouped order in lockstep, so co-resident blocks share A / W panels.  Same k order as the
// other kernels: bit-identical results.
template <class Cfg, class ALoad, class Epi>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void gemm_f32_persist_kernel(
    ALoad al, SegMat W, Epi epi, float* C, long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK, NS = Cfg::NS;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4, RPP = Cfg::ROWS_PER_PASS;
    constexpr int G = BK / 8, ELD = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + NS * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float* Es = smem + NS * (BM + BN) * LDK + wid * 32 * ELD;      // this wave's 32 x 32 transpose patch
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;
    // staging assignment: the 64 lanes of a wave cover (64 / CH) rows x CH float4 chunks with the ROW on the low lane
    // bits, so the 8-lane groups of a ds_write_b128 hit 8 different rows (stride LDK = 20 floats: 8 distinct bank
    // quads) -- chunk-on-low-bits was a 2-way conflict on every staging write (SQ_LDS_BANK_CONFLICT): +2-3 % TFLOP/s.
    constexpr int CH = BK / 4, RW = 64 / CH;
    const int c4 = lane / RW, r0 = wid * RW + lane % RW;

    // tiles of this block
    const int ntiles = tiles_m * tiles_n, nb = gridDim.x, x8 = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (x8 < (nb & 7) ? 1 : 0);
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int lo = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8;
    const int cnt = q8 + (x8 < r8 ? 1 : 0);
    const int my_n = cnt > jb ? (cnt - jb + nbx - 1) / nbx : 0;
    const int KT = K / BK;
    const int total = my_n * KT;
    if (total == 0) return;
    auto tile_of = [&](int i, int& tm, int& tn) __attribute__((always_inline)) {
        const int lin = lo + jb + i * nbx;
        const int per_group = 8 * tiles_n;
        const int g = lin / per_group, first = g * 8;
        const int gsz = min(tiles_m - first, 8);
        const int in = lin - g * per_group;
        tm = first + in % gsz;
        tn = in / gsz;
    };

    typename ALoad::Row arow[A_V4];
    const float* brow[B_V4];
    int l_tile = 0, l_kt = 0;           // load position
    auto set_rows = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
#pragma unroll
        for (int v = 0; v < A_V4; ++v) arow[v] = al.row(min(tm * BM + r0 + v * RPP, M - 1));
#pragma unroll
        for (int v = 0; v < B_V4; ++v) brow[v] = W.row(tn * BN + r0 + v * RPP);
    };
    set_rows(0);

    f32x4 ga[A_V4], gb[B_V4];            // ONE global-load staging set: loads are issued right after the LDS write that
                                         // frees it and have a whole slab time (~3 us) to land
    f32x4 fa[2][TM], fb[2][TN];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = (wm * (TM * 32) + l31) * LDK + 4 * lh;
    const int b_off = (wn * (TN * 32) + l31) * LDK + 4 * lh;

#define GLOAD()                                                                                      \
    {                                                                                                \
        const int k0_ = l_kt * BK + c4 * 4;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i) ga[i] = al.load(arow[i], k0_);              \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i) gb[i] = *(const f32x4*)(brow[i] + k0_);     \
        if (++l_kt == KT) {                                                                          \
            l_kt = 0;                                                                                \
            if (++l_tile < my_n) set_rows(l_tile);                                                   \
        }                                                                                            \
    }
#define LSTORE(BUF)                                                                                              \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < A_V4; ++i)                                                         \
            *(f32x4*)(As + ((BUF) * BM + r0 + i * RPP) * LDK + c4 * 4) = ga[i];                                  \
        _Pragma("unroll") for (int i = 0; i < B_V4; ++i)                                                         \
            *(f32x4*)(Bs + ((BUF) * BN + r0 + i * RPP) * LDK + c4 * 4) = gb[i];                                  \
    }
#define FRAGS(FSET, BUF, T)                                                                                      \
    {                                                                                                            \
        const float* ap_ = As + (BUF) * BM * LDK + a_off + 8 * (T);                                              \
        const float* bp_ = Bs + (BUF) * BN * LDK + b_off + 8 * (T);                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[FSET][i] = *(const f32x4*)(ap_ + i * 32 * LDK);        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[FSET][j] = *(const f32x4*)(bp_ + j * 32 * LDK);        \
    }
#define MFMAS(FSET)                                                                                              \
    {                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[FSET][i][u], fb[FSET][j][u], acc[i][j], 0, 0, 0); \
    }

    // prologue: slabs 0 and 1 staged, slab 2 in flight, first fragments of slab 0 in registers
    GLOAD()
    LSTORE(0)
    if (total > 1) {
        GLOAD()
        LSTORE(1)
    }
    if (total > 2) GLOAD()
    __syncthreads();
    FRAGS(0, 0, 0)

    int buf = 0, c_tile = 0, c_kt = 0;
    auto epilogue = [&]() __attribute__((always_inline)) {
        // per 32x32 MFMA tile: accumulators -> this wave's LDS patch -> row-major float4 -> epilogue -> dwordx4 stores.
        // The residual rows of tile q+1 are requested BEFORE the stores of tile q are issued, so the wait for them is
        // a counted vmcnt(4) and never drains the stores (vmcnt is in-order and counts stores on gfx950).
        int tm, tn;
        tile_of(c_tile, tm, tn);
        const int er = lane >> 3, ec = (lane & 7) * 4;            // float4 #lane of an 8-row stripe
        const int nb0 = tn * BN + wn * (TN * 32) + ec, mb0 = tm * BM + wm * (TM * 32) + er;
        f32x4 rs[4];
        if (Epi::HAS_RES) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) rs[s4] = epi.res4(min(mb0 + s4 * 8, M - 1), nb0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = nb0 + j * 32;
            const typename Epi::Col4 cc = epi.col4(n0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = mb0 + i * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * lh) * ELD + l31] = acc[i][j][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                const bool more = Epi::HAS_RES && (j * TM + i + 1 < TM * TN);
                const int qn = j * TM + i + 1, jn = qn / TM, in = qn % TM;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    // one row group at a time (few live registers): result, then the NEXT tile's residual request,
                    // then the store -- load/store alternate, so the next tile waits with a counted vmcnt(7)
                    const bool ok = mb + s4 * 8 < M;      // an epilogue may have side effects (it stores the GELU
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};       // pre-activation): never run it for rows past M
                    if (ok) o = epi.apply4(*(const f32x4*)(Es + (s4 * 8 + er) * ELD + ec), cc, rs[s4], mb + s4 * 8, n0);
                    if (more) rs[s4] = epi.res4(min(mb0 + in * 32 + s4 * 8, M - 1), nb0 + jn * 32);
                    if (ok) *(f32x4*)(C + (long)(mb + s4 * 8) * ldc + n0) = o;
                }
            }
        }
    };
    for (int gs = 0; gs < total; ++gs) {
        const int b1 = buf + 1 >= NS ? buf + 1 - NS : buf + 1;
        const int b2 = b1 + 1 >= NS ? b1 + 1 - NS : b1 + 1;
#pragma unroll
        for (int t = 0; t < G; ++t) {
            if (t + 1 < G) {
                FRAGS((t + 1) & 1, buf, t + 1)
            } else if (gs + 1 < total) {
                FRAGS(0, b1, 0)
            }
#ifndef EGOTAP_SB_VARIANT
#define EGOTAP_SB_VARIANT 0
#endif
            if (EGOTAP_SB_VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
            MFMAS(t & 1)
            if (EGOTAP_SB_VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
            if (t == G / 2 - 1) {
                if (gs + 2 < total) LSTORE(b2)          // slab gs+2 (requested one slab ago) -> the free LDS slab
                if (gs + 3 < total) GLOAD()             // request slab gs+3 into the registers just freed
                if (EGOTAP_SB_VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf = b1;
        if (++c_kt == KT) {
            epilogue();
            c_kt = 0;
            ++c_tile;
        }
    }
#undef GLOAD
#undef LSTORE
#undef FRAGS
#undef MFMAS
}

template <class Cfg, class ALoad, class Epi>
static hipError_t gemm_f32_persist_launch(const ALoad& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N,
                                          int K, int num_cu, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0) return hipErrorInvalidValue;
    auto kern = gemm_f32_persist_kernel<Cfg, ALoad, Epi>;
    constexpr int LDS = Cfg::LDS_BYTES + (Cfg::THREADS / 64) * 32 * 36 * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget (3 slabs + per-wave transpose patches)");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), LDS, stream, al, W, epi, C, ldc, M, N, K, tiles_m, tiles_n);
    return hipGetLastError();
}

// host launcher; returns hipError_t (no allocation, no sync: capture safe)
template <class Cfg, class ALoad, class Epi>
static hipError_t gemm_f32_launch(const ALoad& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N,
                                  int K, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0) return hipErrorInvalidValue;
    auto kern = gemm_f32_kernel<Cfg, ALoad, Epi>;
    static bool attr_done = false;     // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, al, W, epi, C, ldc, M,
                       N, K, tiles_m, tiles_n);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------- split-K for skinny problems
// Small batches make fc1 of the two encoders a skinny GEMM (M = 30 rows per frame, N = 2048, K = 16384 / 8192): one row of
// output tiles would walk the whole K on 16 CUs (3.5 ms at B = 1, a quarter of the forward).  gemm_f32_splitk_kernel is
// gemm_f32_kernel over a K range per blockIdx.y writing raw partial sums P[split][M][N]; splitk_reduce_kernel adds the
// partials in a fixed order and applies the epilogue.  Used only below a row threshold, so large-batch results (and their
// bit-identity across batch sizes) are untouched.
template <class Cfg, class ALoad>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void gemm_f32_splitk_kernel(ALoad al, SegMat W, float* P, int M, int N, int K, int tiles_m,
                                                                                 int tiles_n, int kper) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4, RPP = Cfg::ROWS_PER_PASS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m, split = blockIdx.y;
    const int bm = tm * BM, bn = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int c4 = tid % (BK / 4), r0 = tid / (BK / 4);
    typename ALoad::Row arow[A_V4];
    const float* brow[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) arow[i] = al.row(min(bm + r0 + i * RPP, M - 1));
#pragma unroll
    for (int i = 0; i < B_V4; ++i) brow[i] = W.row(bn + r0 + i * RPP);
    f32x4 pa0[A_V4], pb0[B_V4], pa1[A_V4], pb1[B_V4];      // two register sets: loads run two slabs ahead (gemm_f32_kernel)
    auto gload = [&](f32x4 (&pa)[A_V4], f32x4 (&pb)[B_V4], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) pa[i] = al.load(arow[i], k0 + c4 * 4);
#pragma unroll
        for (int i = 0; i < B_V4; ++i) pb[i] = *(const f32x4*)(brow[i] + k0 + c4 * 4);
    };
    auto lstore = [&](const f32x4 (&pa)[A_V4], const f32x4 (&pb)[B_V4], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) *(f32x4*)(As + (buf * BM + r0 + i * RPP) * LDK + c4 * 4) = pa[i];
#pragma unroll
        for (int i = 0; i < B_V4; ++i) *(f32x4*)(Bs + (buf * BN + r0 + i * RPP) * LDK + c4 * 4) = pb[i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* Ab = As + (buf * BM + wm * (TM * 32) + l31) * LDK + 4 * lh;
        const float* Bb = Bs + (buf * BN + wn * (TN * 32) + l31) * LDK + 4 * lh;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(Ab + i * 32 * LDK + 8 * t);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bb + j * 32 * LDK + 8 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[j][u], acc[i][j], 0, 0, 0);
        }
    };
    const int k_lo = split * kper, k_hi = min(K, k_lo + kper);
    const int KT = (k_hi - k_lo) / BK;
    if (KT > 0) {
        gload(pa0, pb0, k_lo);
        lstore(pa0, pb0, 0);
        if (KT > 1) gload(pa1, pb1, k_lo + BK);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        if (kt + 2 < KT) gload(pa0, pb0, k_lo + (kt + 2) * BK);
        compute(0);
        if (kt + 1 < KT) lstore(pa1, pb1, 1);
        __syncthreads();
        if (kt + 1 >= KT) break;
        if (kt + 3 < KT) gload(pa1, pb1, k_lo + (kt + 3) * BK);
        compute(1);
        if (kt + 2 < KT) lstore(pa0, pb0, 0);
        __syncthreads();
    }
    float* out = P + (long)split * M * N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mbase = bm + wm * (TM * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < M) out[(long)m * N + n] = acc[i][j][r];
            }
        }
    }
}

template <class Epi>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, Epi epi, float* __restrict__ C, long ldc, int M, int N,
                                                            int splits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += P[(long)k * M * N + i];
    C[(long)m * ldc + n] = epi.apply(s, epi.col(n), m, n);
}

// [r4] The split count with the least estimated time: a CU runs its ceil(tiles * splits / CUs) workgroups' K-slabs one after the other
// on its four matrix pipes (1.7 us per 128 x 128 x 32 fp32 slab at the MFMA rate, ~1.9 measured; ~2.1 when the workgroup is alone on its CU:
// one wave per SIMD and nothing to run under its loads) + ~2.5 us per workgroup (first slab's round trip, partial store), and the partials
// cost their bytes twice.  The doubling rule it replaces overshot (qkv at B = 1: 120 tiles x 8 = 960 workgroups of 4 slabs each, 56 MB of
// partials, 44.8 us; 2 splits: one workgroup per CU) or left half-empty rounds (N = 4096: 160 x 4 = 2.5 per CU).
struct SplitPlan { int splits; double us; };
template <class Cfg>
static inline SplitPlan gemm_f32_splitk_plan(int M, int N, int K, size_t p_floats, int num_cu) {
    const int KT = K / Cfg::BK, tiles = ((M + Cfg::BM - 1) / Cfg::BM) * (N / Cfg::BN);
    SplitPlan best{1, 1e30};
    for (int sp = 1; sp <= 32; ++sp) {
        const int kp = (KT + sp - 1) / sp;
        if ((sp > 1 && kp < 4) || (long)kp * (sp - 1) >= KT) continue;      // at least 4 slabs per range; no empty last range
        if ((size_t)sp * M * N > p_floats) break;
        const int per_cu = (tiles * sp + num_cu - 1) / num_cu;
        constexpr double AREA = (double)Cfg::BM * Cfg::BN / (128.0 * 128.0);      // slab time scales with the tile's MFMA count (the constants are the 128 x 128 tile's)
        const double t = per_cu * (kp * (per_cu == 1 ? 2.1 : 1.9) * AREA + 2.5) + (sp > 1 ? 2.0 * sp * (double)M * N * 4 / 4.0e6 : 0.0);
        if (t < best.us - 1e-9) best = SplitPlan{sp, t};
    }
    return best;
}
// Estimated time (us) of the same 128 x 128 kernel WITHOUT a split, its epilogue inside (gemm_f32_launch), for shapes with many slabs per
// tile -- fitted to tools/gemm_small_vs_big_probe.py (48 shapes, M = 3456 ... 55296, K = 1024 / 4096): p = tiles / CUs workgroups per CU.
// p <= 1: every workgroup is alone on its CU: 2.3 us per slab + 15.  p > 1: two workgroups share a CU (1.95 us per slab each + 10) and the
// dispatcher hands a CU whose pair has finished the next PAIR, so rounds come in twos: ceil(p / 2) x 2 x that -- p = 2.25 costs what p = 3.4
// costs (1016 / 1028 us at K = 4096).  Within 3 % of every K = 4096 point, 10 % of every K = 1024 one.
template <class Cfg>
static inline double gemm_f32_direct_estimate_us(int M, int N, int K, int num_cu) {
    const int KT = K / Cfg::BK, tiles = ((M + Cfg::BM - 1) / Cfg::BM) * (N / Cfg::BN);
    const double p = (double)tiles / num_cu;
    return p <= 1.0 ? 2.3 * KT + 15.0 : ceil(p / 2.0) * 2.0 * (1.95 * KT + 10.0);
}

// P: scratch of at least splits * M * N floats; returns hipErrorInvalidValue for shapes the tile does not cover
// the partial-sum launch alone: *splits_out ranges of K into P[split][M][N] (the caller reduces)
// `splits` comes from the caller's plan (gemm_f32_splitk_plan: ONE decision, shared by whoever reduces the partials afterwards)
template <class Cfg, class ALoad>
static hipError_t gemm_f32_splitk_partials(const ALoad& al, const SegMat& W, float* P, size_t p_floats, int M, int N, int K, hipStream_t stream, int splits) {
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0) return hipErrorInvalidValue;
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    const int KT = K / Cfg::BK;
    if (splits < 1 || (splits > 1 && (long)((KT + splits - 1) / splits) * (splits - 1) >= KT)) return hipErrorInvalidValue;      // no empty last range
    if ((size_t)splits * M * N > p_floats) return hipErrorInvalidValue;
    const int kper = ((KT + splits - 1) / splits) * Cfg::BK;
    auto kern = gemm_f32_splitk_kernel<Cfg, ALoad>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, al, W, P, M, N, K, tiles_m, tiles_n, kper);
    return hipGetLastError();
}
// splits = 0: the plan's count for this shape
template <class Cfg, class ALoad, class Epi>
static hipError_t gemm_f32_splitk_launch(const ALoad& al, const SegMat& W, const Epi& epi, float* C, long ldc, float* P, size_t p_floats, int M,
                                         int N, int K, hipStream_t stream, int num_cu = 256, int splits = 0) {
    if (M <= 0) return hipSuccess;
    if (splits == 0) splits = gemm_f32_splitk_plan<Cfg>(M, N, K, p_floats, num_cu).splits;
    hipError_t e = gemm_f32_splitk_partials<Cfg>(al, W, P, p_floats, M, N, K, stream, splits);
    if (e != hipSuccess) return e;
    const long total = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel<Epi>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float*)P, epi, C, ldc, M, N, splits);
    return hipGetLastError();
}
