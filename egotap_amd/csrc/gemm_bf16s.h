// bf16-STORAGE GEMM for the reduced-precision training / inference mode (EGOTAP_PREC_BF16):
//     OUT = epi( X[M,K] * W[N,K]^T ),   X, W bf16 in HBM (activations are written as bf16 by their producers, weights are
// rounded once per step by prep_weights_kernel), fp32 accumulate, OUT bf16 or fp32 -- nn.Linear forward and input gradient
// (modeling_vit.py:226-230, 271, 319-344; net_architecture.py:366-368, 259-261) without a conversion pass per operand.
//
// Structure (256 x 256 tile, 8 waves = 2 groups x 4, one wave of each group per SIMD):
//   * v_mfma_f32_16x16x32_bf16, MFMA A operand = W rows, B operand = X rows, so a lane's 4 accumulator registers are 4
//     consecutive n of one row m.  Wave (g, wc) owns rows g*128..+127, columns wc*64..+63: 8 x 4 MFMA tiles, 128 accumulators.
//   * K is walked in 32-deep K-tiles through a RING of four 32 KB LDS stages ([256 X rows | 256 W rows] x 64 bytes), filled
//     by global_load_lds_dwordx4 (no staging registers, no ds_write).  A row's four 16-byte chunks are stored at chunk
//     position c ^ swz(row), swz = (4 - ((row >> 2) & 3)) & 3: the ds_read_b128 of a 16x16x32 fragment (16 rows, chunk
//     lane >> 4) then touches 16 distinct 16-byte bank groups in each of the instruction's four lane groups.  The swizzle
//     is applied on the global side (which 16 bytes a lane fetches); the DMA itself writes lane i at base + 16 i.
//   * Two PHASES per K-tile: phase a multiplies the wave's row half a (4 m-tiles x 4 n-tiles = 16 MFMAs); phase 0 reads the
//     4 W fragments (kept for phase 1) and 4 X fragments, phase 1 reads 4 X fragments: 12 ds_read_b128 per 32 MFMAs.
//   * The two wave groups run one barrier apart (group 1 takes one extra barrier up front): while a group's 16 MFMAs own the
//     SIMD's matrix pipe, the other group's wave on that SIMD issues its DMA, its fragment reads and its waits.  Every
//     phase is  [DMA of one half stage; fragment reads; vmcnt(8)] barrier [16 MFMAs] barrier.
//   * Ring schedule (T = K-tile index of this workgroup's slab stream, which runs on across output tiles): phase 2T issues
//     W(T+3), phase 2T+1 issues X(T+3).  Write-after-read: W(T) is last read in phase 2T, X(T) in 2T+1; with the groups
//     one barrier apart a region may be refilled two phases after its last read -- exactly W(T+4) at 2T+2, X(T+4) at 2T+3.
//     Read-after-write: the vmcnt(8) at the end of phase 2T-1 (four newer half stages may stay in flight) retires X(T) and W(T);
//     the barrier that follows publishes them to the phase-2T readers of both groups.
//   * Epilogue per wave through a private 4 KB LDS patch (xor-swizzled 16 x 64 fp32): accumulators -> rows -> epilogue
//     functor on 4 (fp32 out) or 8 (bf16 out) consecutive n -> full-row-segment global stores.  No barrier inside, so the
//     other group's MFMAs run under it.
#pragma once
#include <type_traits>

#include "gemm_bf16.h"

typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));

struct SCfg {
    static constexpr int BM = 256, BN = 256, BK = 32, NS = 4, THREADS = 512;
    static constexpr int ROWB = 64;                       // bytes per LDS row (32 bf16)
    static constexpr int PART = 256 * ROWB;               // X part or W part of a stage: 16 KiB
    static constexpr int STAGE = 2 * PART;                // 32 KiB
    static constexpr int EPATCH = 16 * 64 * 4;            // per-wave epilogue patch: 16 rows x 64 fp32
    static constexpr int LDS_BYTES = NS * STAGE + (THREADS / 64) * EPATCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------------------------------------------- X-operand loaders (bf16)
// ptr(row, k0, ko): address of the 8 consecutive k (16 bytes) k0 + ko .. of logical row `row`; k0 = first k of a 32-deep K-tile
// (wave-uniform: the index arithmetic on it stays on the scalar unit), ko = 0, 8, 16 or 24
struct XPlain {
    const __bf16* A;
    long lda;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const { return Row{A + (long)m * lda}; }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const { return r.p + k0 + ko; }
};
// fc1 of the position encoder: row (b, i) = the ppd x ppd patch tokens of heatmap i (net_architecture.py:388-406), tokens bf16 [B*seq, D]
struct XTokens {
    const __bf16* Y;
    int T, D, seq, side, ppd, grid;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / T, i = m - b * T;
        return Row{Y + ((long)b * seq + (long)(ppd * (i / grid)) * side + ppd * (i % grid)) * D};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const {      // D % 32 == 0: a K-tile stays inside one patch token
        const int s = k0 / D, c = k0 - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        return r.p + (long)(prl * side + pcl) * D + c + ko;
    }
};
// fc1 of the rotation encoder: row (b, eye*J + j) = [cos map | sin map] of limb j (net_architecture.py:690-694), hm bf16 [B, C, HW]
struct XRot {
    const __bf16* hm;
    int C, J, HW;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int T = 2 * J;
        const int b = m / T, t = m - b * T;
        const int eye = t / J, j = t - eye * J;
        return Row{hm + (long)(b * C + 2 * J + eye * 2 * J + j) * HW};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const {      // HW % 32 == 0
        const int cs = k0 / HW;
        return r.p + (long)cs * J * HW + (k0 - cs * HW) + ko;
    }
};

// [r3] patch embedding (modeling_vit.py:137-153 over the tiled heatmap image, net_architecture.py:326-336): row (b, token) = the 16 x 16
// patch of heatmap `cell` the token covers, k = (py, px) -- 8 consecutive k are 8 consecutive pixels of one patch row: one 16-byte piece of
// the bf16 heatmaps [B, C, S, S].  Dummy cells (cell >= T) read a page of zeros; SEpiPatchF32 puts the mask token there.
struct XPatch {
    const __bf16* hm;
    const __bf16* zero;
    int C, S, seq, side, ppd, grid, T;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / seq, tok = m - b * seq;
        const int pr = tok / side, pc = tok - pr * side;
        const int cell = (pr / ppd) * grid + pc / ppd;
        if (cell >= T) return Row{nullptr};
        return Row{hm + ((long)(b * C + cell) * S + (pr % ppd) * 16) * S + (pc % ppd) * 16};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int k0, int ko) const {
        const int k = k0 + ko;
        return r.p ? r.p + (k >> 4) * S + (k & 15) : zero + ko;
    }
};

// ---------------------------------------------------------------------------------------------------- epilogues
// W = elements per lane and call: 4 for fp32 outputs (16 lanes cover a 256-byte row segment), 8 for bf16 outputs (8 lanes cover
// 128 bytes).  col(n): per-column constants of the lane's W columns (loaded once per tile); fetch(m, n): per-element operands from
// memory (residual, saved pre-activation), requested one 16-row block ahead of their use; emit(v, col, aux, m, n): v[0..W) =
// accumulators of row m, columns n..n+W-1 (m < M guaranteed) -> exactly STORES global store instructions (the main loop counts
// them: vmcnt is one in-order counter for DMA, loads and stores).
struct SNoAux {};
// false for an epilogue whose emit() may skip its stores for whole waves (column guards): the main loop then gives the previous
// tile's stores no allowance in its vmcnt waits (the allowance must never exceed the stores really in flight)
template <class E> struct s_epi_exact { static constexpr bool value = true; };
// "use" of a loaded value outside any lane predicate: the compiler then waits for the load HERE.  Without it a value whose only
// uses sit under `if (m < M)` counts as possibly pending at the loop's back edge, and the waitcnt pass protects the registers it
// lands in with vmcnt(0) waits in the middle of the main loop (they are reused as fragment registers there).
__device__ __forceinline__ void s_keep(const SNoAux&) {}
__device__ __forceinline__ void s_keep(const f32x4& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void s_keep(const __bf16 __attribute__((ext_vector_type(8))) & v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void store_bf16x8(__bf16* p, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    // [r3] streaming store: a bf16 activation is 0.3-2.4 GB, written once and read by the NEXT kernel -- keeping its lines out of
    // the L2 leaves the operand panels this kernel re-reads there (+2-3 % on every bf16-output GEMM, profiles/r03_gemm_ablation.md)
#if defined(EGOTAP_ABL) && (EGOTAP_ABL & 4)      // timing-only: the functor's arithmetic without the store
    asm volatile("" ::"v"(o));
#elif defined(EGOTAP_ABL) && (EGOTAP_ABL & 8)    // A/B: write-back instead of streaming stores
    *(bf16x8*)p = o;
#else
    __builtin_nontemporal_store(o, (bf16x8*)p);
#endif
}
struct SBias8 { f32x4 b0, b1; };
__device__ __forceinline__ void s_keep(const SBias8& b) { asm volatile("" ::"v"(b.b0), "v"(b.b1)); }
__device__ __forceinline__ SBias8 load_bias8(const float* bias, int n) {
    if (bias == nullptr) return SBias8{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    return SBias8{*(const f32x4*)(bias + n), *(const f32x4*)(bias + n + 4)};
}
struct SEpiBf16 {            // out bf16 = acc (+ bias)
    static constexpr int W = 8, STORES = 1;
    const float* bias;       // may be null
    __bf16* out;
    long ld;
    typedef SBias8 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return load_bias8(bias, n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] += c.b0[i]; v[4 + i] += c.b1[i]; }
        store_bf16x8(out + (long)m * ld + n, v);
    }
};
struct SEpiResF32 {          // out f32 = acc + bias + R   (R may alias out)
    static constexpr int W = 4, STORES = 1;
    const float* bias;
    const float* R;
    float* out;
    long ld;
    typedef f32x4 Col;
    typedef f32x4 Aux;
    __device__ __forceinline__ Col col(int n) const { return *(const f32x4*)(bias + n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return *(const f32x4*)(R + (long)m * ld + n); }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux& r, int m, int n) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[i] + c[i] + r[i];
        *(f32x4*)(out + (long)m * ld + n) = o;
    }
};
struct SEpiF32 {             // out f32 = acc + bias  (fc1: the BatchNorm statistics need the fp32 pre-activation)
    static constexpr int W = 4, STORES = 1;
    const float* bias;
    float* out;
    long ld;
    typedef f32x4 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return *(const f32x4*)(bias + n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[i] + c[i];
        *(f32x4*)(out + (long)m * ld + n) = o;
    }
};
struct SPatch4 { f32x4 b, mt; };
__device__ __forceinline__ void s_keep(const SPatch4& c) { asm volatile("" ::"v"(c.b), "v"(c.mt)); }
struct SEpiPatchF32 {        // out f32 = (dummy cell ? mask_token : acc + bias) + position_embeddings[token]   (EpiPatch of gemm_f32.h)
    static constexpr int W = 4, STORES = 1;
    const float *bias, *mask_tok, *pos;
    float* out;
    int D, seq, side, ppd, grid, T;
    typedef SPatch4 Col;
    typedef f32x4 Aux;
    __device__ __forceinline__ Col col(int n) const { return Col{*(const f32x4*)(bias + n), *(const f32x4*)(mask_tok + n)}; }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return *(const f32x4*)(pos + (long)(m % seq) * D + n); }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux& pe, int m, int n) const {
        const int tok = m % seq;
        const int pr = tok / side, pc = tok - pr * side;
        const bool dummy = (pr / ppd) * grid + pc / ppd >= T;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (dummy ? c.mt[i] : v[i] + c.b[i]) + pe[i];
        *(f32x4*)(out + (long)m * D + n) = o;
    }
};
struct SEpiRawF32 {          // out f32 = acc   (split-K partial sums: slab [M][ksplit * N])
    static constexpr int W = 4, STORES = 1;
    float* out;
    long ld;
    typedef SNoAux Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col&, const Aux&, int m, int n) const {
        *(f32x4*)(out + (long)m * ld + n) = f32x4{v[0], v[1], v[2], v[3]};
    }
};
struct SEpiGeluSave {        // z = acc + bias -> Z (bf16, kept for the backward); H = GELU(z) (bf16)
    static constexpr int W = 8, STORES = 2;
    const float* bias;
    __bf16 *Z, *H;
    long ld;
    typedef SBias8 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return load_bias8(bias, n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
        float g[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] += c.b0[i]; v[4 + i] += c.b1[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = gelu_erf(v[i]);
        store_bf16x8(Z + (long)m * ld + n, v);
        store_bf16x8(H + (long)m * ld + n, g);
    }
};
struct SEpiGelu {            // out bf16 = GELU(acc + bias)   (inference: the pre-activation is not kept)
    static constexpr int W = 8, STORES = 1;
    const float* bias;
    __bf16* H;
    long ld;
    typedef SBias8 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return load_bias8(bias, n); }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
        float g[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[i] = gelu_erf(v[i] + c.b0[i]); g[4 + i] = gelu_erf(v[4 + i] + c.b1[i]); }
        store_bf16x8(H + (long)m * ld + n, g);
    }
};
struct SBn4 { f32x4 b, mu, sc, sh; };
__device__ __forceinline__ void s_keep(const SBn4& c) { asm volatile("" ::"v"(c.b), "v"(c.mu), "v"(c.sc), "v"(c.sh)); }
struct SEpiBnLreluF32 {      // out f32 = LeakyReLU_0.2(BatchNorm1d_eval(acc + bias))   (network_utils.py:123-142, eval mode)
    static constexpr int W = 4, STORES = 1;
    const float *bias, *gamma, *beta, *mean, *var;
    float eps, slope;
    float* out;
    long ld;
    typedef SBn4 Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const {
        const f32x4 g = *(const f32x4*)(gamma + n), vv = *(const f32x4*)(var + n);
        f32x4 sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = g[i] / sqrtf(vv[i] + eps);
        return Col{*(const f32x4*)(bias + n), *(const f32x4*)(mean + n), sc, *(const f32x4*)(beta + n)};
    }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col& c, const Aux&, int m, int n) const {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = (v[i] + c.b[i] - c.mu[i]) * c.sc[i] + c.sh[i];
            o[i] = y > 0.f ? y : slope * y;
        }
        *(f32x4*)(out + (long)m * ld + n) = o;
    }
};
struct SEpiGeluGrad {        // out bf16 = acc * GELU'(Z)   (Z bf16: the saved pre-activation)
    static constexpr int W = 8, STORES = 1;
    const __bf16* Z;
    __bf16* out;
    long ld;
    typedef SNoAux Col;
    typedef bf16x8 Aux;
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return *(const bf16x8*)(Z + (long)m * ld + n); }
    __device__ __forceinline__ void emit(float* v, const Col&, const Aux& z, int m, int n) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= dgelu_erf((float)z[i]);
        store_bf16x8(out + (long)m * ld + n, v);
    }
};
// [r3] the same, also emitting column sums of the bf16 values it stores: the bias gradient of the layer whose output gradient this
// is (intermediate.dense: d bias = column sums of dz) leaves the kernel as per-wave partial sums -- row 2 tile_m + group of
// colpart[2 tiles_m][N] holds the sums over that wave's 128 rows -- instead of costing a pass over the 4.8 GB tensor
// (colsum_bf16_partial_kernel: 2.5 ms of the B = 1024 step).  A small fp32 column sum over the 2 tiles_m rows finishes it.
struct SEpiGeluGradCS : SEpiGeluGrad {
    float* colpart;
};
template <class E> struct s_epi_colsum { static constexpr bool value = false; };
template <class E> struct s_epi_stagger { static constexpr bool value = false; };
template <> struct s_epi_stagger<SEpiResF32> { static constexpr bool value = true; };
// (the same staggered start on the two GELU epilogues of the training step: 199.3 / 199.8 ms against 199.5 / 199.3 ms per config-3 step, same box: not adopted)
template <> struct s_epi_colsum<SEpiGeluGradCS> { static constexpr bool value = true; };
// input gradient of fc1 of the position encoder, scattered back to token order: row (b, i), column (s, c) -> token
// (b, patch s of heatmap i), channel c.  The gather of XTokens is a bijection onto the non-dummy tokens (dummy rows stay zero:
// the caller clears the buffer once).  out bf16 [B*seq, D]
struct SEpiScatterTokens {
    static constexpr int W = 8, STORES = 1;
    __bf16* out;
    int T, D, seq, side, ppd, grid;
    typedef SNoAux Col;
    typedef SNoAux Aux;
    __device__ __forceinline__ Col col(int n) const { return Col{}; }
    __device__ __forceinline__ Aux fetch(int m, int n) const { return Aux{}; }
    __device__ __forceinline__ void emit(float* v, const Col&, const Aux&, int m, int n) const {
        const int b = m / T, i = m - b * T;
        const int s = n / D, c = n - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        const long tok = (long)b * seq + (long)(ppd * (i / grid) + prl) * side + ppd * (i % grid) + pcl;
        store_bf16x8(out + tok * D + c, v);
    }
};

// ---------------------------------------------------------------------------------------------------- kernel
// NI = 16-column MFMA tiles per wave: 4 -> a 256 x 256 output tile (the default); [r3] 2 -> 256 x 128, 1 -> 256 x 64, for GEMMs whose N
// is 128 / 64 (the ResNet-18 stages with 128 / 64 output channels ran as N = 256 with a column guard: 2-4 x the executed FLOPs).  The W
// part of a stage shrinks to 64 NI rows, a phase to 4 NI MFMAs per wave; ring, barriers and X traffic are unchanged.
template <class XL, class Epi, int NI = 4>
__global__ __launch_bounds__(SCfg::THREADS, 2) void gemm_bf16s_kernel(XL xl, const __bf16* __restrict__ Wb, long ldw, Epi epi, int M, int N, int K,
                                                                       int tiles_m, int tiles_n_real, int ksplit) {
    // [r3] ksplit > 1: split-K for few-row GEMMs (fc1 of the encoders below a full chip of tiles).  The launch is the GEMM
    // [M, ksplit * N] with K = the per-split depth: column tile tnx = (split, tn) reads W rows tn at k offset split * K and X at the
    // same k offset, and its output columns are tnx * BN .. of a [M][ksplit * N] fp32 slab that splitk_rows_reduce_kernel folds.
    using Cfg = SCfg;
    const int tiles_n = tiles_n_real * ksplit;
    static_assert(NI == 4 || NI == 2 || NI == 1, "n-tiles per wave");
    constexpr int BM = Cfg::BM, BN = 64 * NI, BK = Cfg::BK, NS = Cfg::NS, ROWB = Cfg::ROWB, PART = Cfg::PART, STAGE = Cfg::STAGE;
    constexpr int NWI = NI == 4 ? 2 : 1;             // W-part DMA instructions per wave and K-tile (NI = 1: waves w and w + 4 fetch the same block)
    constexpr int VMC = 4 + 2 * NWI;                 // vmcnt allowance: the four newest half stages (2 X + 2 W) may stay in flight
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wc = wid & 3;
    float* Es = (float*)(smem_s + NS * STAGE + wid * Cfg::EPATCH);

    // tiles of this workgroup: XCD-aware chunk of the grouped tile order (as gemm_f32_persist_kernel)
    const int ntiles = tiles_m * tiles_n, nb = gridDim.x, x8 = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (x8 < (nb & 7) ? 1 : 0);
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int lo_t = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8;
    const int cnt = q8 + (x8 < r8 ? 1 : 0);
    const int my_n = cnt > jb ? (cnt - jb + nbx - 1) / nbx : 0;
    const int KT = K / BK;
    const int total = my_n * KT;                     // K-tiles of this workgroup's stream
    if (total == 0) return;
    auto tile_of = [&](int i, int& tm, int& tn) __attribute__((always_inline)) {
        const int lin = lo_t + jb + i * nbx;
        const int per_group = 8 * tiles_n;
        const int g = lin / per_group, first = g * 8;
        const int gsz = min(tiles_m - first, 8);
        const int in = lin - g * per_group;
        tm = first + in % gsz;
        tn = in / gsz;
    };

    // ---- DMA duty of this wave per part: 16-row blocks wid and wid + 8.  Lane -> row lane >> 2 of the block, chunk position
    // lane & 3, which holds logical chunk (lane & 3) ^ swz(row), swz(row) = (4 - ((row >> 2) & 3)) & 3 and (row >> 2) & 3 = lane >> 4.
#if defined(EGOTAP_ABL) && (EGOTAP_ABL & 1)      // timing-only: every DMA wave-instruction reads 8 rows x 128 B instead of 16 rows x 64 B (wrong data)
    const int drow = lane >> 3, dchunk = lane & 7;
#else
    const int drow = lane >> 2, dchunk = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
#endif
    typename XL::Row xr0, xr1;
    const __bf16 *pw0, *pw1;
    int lx_tile = 0, lx_kt = 0, lw_tile = 0, lw_kt = 0;       // position of the X / W issue streams (the W stream runs one phase ahead)
    int xkb = 0;                                               // k offset of the X stream's tile (split * K; 0 without split-K)
    auto set_x = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        xr0 = xl.row(min(tm * BM + wid * 16 + drow, M - 1));
        xr1 = xl.row(min(tm * BM + (wid + 8) * 16 + drow, M - 1));
        xkb = ksplit > 1 ? (tn / tiles_n_real) * K : 0;
    };
    auto set_w = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        long koff = 0;
        if (ksplit > 1) { const int sp = tn / tiles_n_real; tn -= sp * tiles_n_real; koff = (long)sp * K; }
        constexpr int WBLK = BN / 16;                // 16-row blocks of the W part
        pw0 = Wb + (long)(tn * BN + (wid % WBLK) * 16 + drow) * ldw + koff + dchunk * 8;
        pw1 = NI == 4 ? Wb + (long)(tn * BN + (wid + 8) * 16 + drow) * ldw + koff + dchunk * 8 : pw0;
    };
    set_x(0);
    set_w(0);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_s;
    // inline asm: the compiler's waitcnt pass would drain vmcnt(0) before every LDS read after __builtin_amdgcn_global_load_lds;
    // the waits are counted by hand below (a constant number of DMA instructions per phase, unconditionally)
    auto dma1 = [&](const __bf16* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    auto issue_w = [&](int st) __attribute__((always_inline)) {        // W part of the next K-tile of the W stream -> stage st
        const unsigned sa = lds0 + st * STAGE + PART + (wid % (BN / 16)) * 1024;
        const int k0 = lw_kt * BK;
        dma1(pw0 + k0, sa);
        if constexpr (NI == 4) dma1(pw1 + k0, sa + 8 * 1024);
        if (lw_tile < my_n && ++lw_kt == KT) {
            lw_kt = 0;
            if (++lw_tile < my_n) set_w(lw_tile);
            else lw_kt = KT - 1;                 // stream exhausted: keep re-reading the last K-tile into a stage nobody reads
        }
    };
    auto issue_x = [&](int st) __attribute__((always_inline)) {
        const unsigned sa = lds0 + st * STAGE + wid * 1024;
        const int k0 = lx_kt * BK + xkb;
        dma1(xl.ptr(xr0, k0, dchunk * 8), sa);
        dma1(xl.ptr(xr1, k0, dchunk * 8), sa + 8 * 1024);
        if (lx_tile < my_n && ++lx_kt == KT) {
            lx_kt = 0;
            if (++lx_tile < my_n) set_x(lx_tile);
            else lx_kt = KT - 1;
        }
    };

    f32x4 acc[8][NI];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment byte offset of this lane inside a 16-row block: row lane & 15, logical chunk lane >> 4
    const int l15 = lane & 15;
    const int foff = l15 * ROWB + (((lane >> 4) ^ ((4 - (l15 >> 2)) & 3)) << 4);
    const int x_base = (grp * 128) * ROWB + foff;             // + mi * 16 * ROWB
    const int w_base = PART + (wc * 16 * NI) * ROWB + foff;   // + ni * 16 * ROWB

    int c_tile = 0, c_kt = 0;
    // number of K-tiles (from the start of a tile) whose waits must leave the previous epilogue's stores in flight: vmcnt is ONE
    // in-order counter, so a wait that retires a DMA also waits for every older store; the epilogue's NST stores are younger than
    // the (up to) four half stages in flight and older than the ones issued after it, so for the first two K-tiles (four phases)
    // of the next tile the allowance is 8 + NST, by which time the stores have had ~2 us to drain.  Only after a FULL tile (every
    // lane active in every store: the count is exact); after a ragged tile the plain allowance makes the first wait drain them.
    constexpr bool CS = s_epi_colsum<Epi>::value;
    static_assert(!CS || (Epi::W == 8 && NI == 4), "column sums ride on the bf16-output epilogues of the 256-column tile");
    constexpr int IT = Epi::W == 4 ? 4 : 2, NST = 8 * IT * Epi::STORES + (CS ? 2 : 0);
    constexpr int SLK = s_epi_exact<Epi>::value ? VMC + NST : VMC;      // vmcnt allowance of the first K-tiles after an (exact) epilogue
    static_assert(SLK <= 63, "vmcnt is a 6-bit counter");
    int slack_kt = 0;
    // [r4] FULL (every row of the tile exists): unconditional stores, so that hipcc counts them exactly and its waits for the aux rows of
    // the next 16-row block leave this block's stores in flight (gemm_bf16s64.h has the derivation: under `if (m < M)` every block waited
    // for the previous block's stores to complete)
    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        int tm, tn;
        tile_of(c_tile, tm, tn);
        const int m_wave = tm * BM + grp * 128, n_wave = tn * BN + wc * 16 * NI;
        const int q = lane >> 4;
        // lane -> (row of the 16-row block, first column) of its IT pieces; with NI < 4 the wave owns 16 NI columns: the lanes of the
        // columns past them idle in the epilogue (cact)
        const int er = Epi::W == 4 ? q : (lane >> 3), ecol = Epi::W == 4 ? l15 * 4 : (lane & 7) * 8;
        const bool cact = ecol < 16 * NI;
        const int en = n_wave + (cact ? ecol : 0);
        const typename Epi::Col cc = epi.col(en);
        typename Epi::Aux ax[IT], an[IT];
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // column sums of this lane's 8 columns over its 16 rows (CS epilogues)
#pragma unroll
        for (int it = 0; it < IT; ++it) ax[it] = epi.fetch(min(m_wave + (16 / IT) * it + er, M - 1), en);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            if (mi + 1 < 8) {
#pragma unroll
                for (int it = 0; it < IT; ++it) an[it] = epi.fetch(min(m_wave + (mi + 1) * 16 + (16 / IT) * it + er, M - 1), en);
            }
            // accumulators -> patch[16 m][64 n] (fp32), 16-byte chunk ch of row r at position ch ^ r
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                *(f32x4*)(Es + l15 * 64 + (((4 * ni + q) ^ l15) << 2)) = acc[mi][ni];
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int m0 = m_wave + mi * 16;
            if (mi == 0) s_keep(cc);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = (16 / IT) * it + er;
                s_keep(ax[it]);
                if constexpr (Epi::W == 4) {
                    const f32x4 v = *(const f32x4*)(Es + r * 64 + ((l15 ^ r) << 2));
                    float vv[4] = {v[0], v[1], v[2], v[3]};
#if !(defined(EGOTAP_ABL) && (EGOTAP_ABL & 2))      // timing-only: the epilogue without its global stores / functor
                    if ((FULL || m0 + r < M) && cact) epi.emit(vv, cc, ax[it], m0 + r, en);
#else
                    asm volatile("" ::"v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]));
#endif
                } else {
                    const int c2 = lane & 7;
                    const f32x4 v0 = *(const f32x4*)(Es + r * 64 + (((2 * c2) ^ r) << 2));
                    const f32x4 v1 = *(const f32x4*)(Es + r * 64 + (((2 * c2 + 1) ^ r) << 2));
                    float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#if !(defined(EGOTAP_ABL) && (EGOTAP_ABL & 2))
                    if ((FULL || m0 + r < M) && cact) {
                        epi.emit(vv, cc, ax[it], m0 + r, en);              // (leaves the values it stored in vv)
                        if constexpr (CS) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) cs[i] += (float)(__bf16)vv[i];      // the sum of what was STORED (bf16)
                        }
                    }
#else
                    asm volatile("" ::"v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]), "v"(vv[4]), "v"(vv[5]), "v"(vv[6]), "v"(vv[7]));
#endif
                }
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) ax[it] = an[it];
        }
        if constexpr (CS) {
            // lanes with equal lane & 7 hold the same 8 columns for different rows: fold the 8 row groups (fixed order), lanes 0..7 store
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cs[i] += __shfl_xor(cs[i], 8, 64);
                cs[i] += __shfl_xor(cs[i], 16, 64);
                cs[i] += __shfl_xor(cs[i], 32, 64);
            }
            if (lane < 8) {
                float* cp = epi.colpart + (long)(2 * tm + grp) * N + en;
                *(f32x4*)cp = f32x4{cs[0], cs[1], cs[2], cs[3]};
                *(f32x4*)(cp + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
            }
        }
        slack_kt = (s_epi_exact<Epi>::value && (tm + 1) * BM <= M && KT >= 4) ? 2 : 0;
    };

    // Tried and dropped (round 2): all 8 waves in lockstep with one barrier per K-tile in its middle and every DMA issue / fragment
    // read placed between the MFMAs of a burst (the schedule that gained 2 % on gemm_f32_dma.h).  bf16-output GEMMs within +-3 % of
    // the two-group schedule below (864 / 1069 / 1110 / 1170 TF against 857 / 1072 / 1068 / 1215 at N x K = 3072x1024, 1024x1024,
    // 4096x1024, 1024x4096), but the fp32-output + residual epilogue, which the two groups run side by side under each other's MFMAs,
    // falls from 710-1100 TF to 545-790.
    // [r3] Epilogues that move 8 bytes per element both ways (fp32 residual in, fp32 out: 512 KB per tile) are paced by the memory
    // system when all 256 workgroups reach them together -- they all start together and every tile takes the same time.  Starting
    // the workgroups a quarter tile apart (four phases, ~8 k cycles each, once per launch) spreads the bursts: +7-9 % on the
    // attention-output and MLP-down GEMMs (profiles/r03_gemm_ablation.md); bf16-output epilogues (2 bytes, one way) gain nothing.
    if constexpr (s_epi_stagger<Epi>::value) {
        for (int i = 0; i < (int)((blockIdx.x >> 3) & 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
    // ---- prologue: K-tiles 0, 1, 2 complete (W then X each), then the steady-state issues of phases 0.. pick up W(3), X(3)
    issue_w(0); issue_x(0);
    issue_w(1); issue_x(1);
    issue_w(2); issue_x(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");      // K-tile 0 has landed (K-tiles 1, 2 may be in flight) ...
    __builtin_amdgcn_s_barrier();                          // ... for every wave: phase 0 may read it
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    bf16x8 wf[NI], xf[4];
    int st = 0;
    for (int T = 0; T < total; ++T) {
        const char* sa = smem_s + st * STAGE;
        const int st3 = (st + 3) & 3;
        // ---------------- phase 2T: row half 0
        issue_w(st3);
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8*)(sa + w_base + j * 16 * ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(sa + x_base + i * 16 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        // retire the half stage issued four phases ago (this phase's and the three before it may stay in flight): it is read from
        // the next phase on.  The wait sits behind this phase's own DMA and fragment reads, which do not depend on it.
        if (slack_kt > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLK) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- phase 2T + 1: row half 1
        issue_x(st3);
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(sa + x_base + (4 + i) * 16 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        if (slack_kt > 0) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLK) : "memory"); --slack_kt; }
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        st = (st + 1) & 3;
        if (++c_kt == KT) {
            // One extra barrier per tile and group lets the two groups' epilogues run side by side: group 0 waits here for group 1's
            // last MFMA phase (256 cycles idle), then both store their tiles concurrently; without it each group sits at a barrier
            // through most of the other's epilogue (they are latency-bound: LDS round trips, residual loads, store issue).
            if (grp == 0) __builtin_amdgcn_s_barrier();
            {
                int tm_, tn_;
                tile_of(c_tile, tm_, tn_);
                // (fp32-output epilogues only: with the bf16-output GELU-grad epilogue the unconditional copy let hipcc hoist aux loads across
            // blocks -- 254 VGPRs -- and ran 11 % slower)
            if (Epi::W == 4 && (tm_ + 1) * BM <= M) epilogue(std::integral_constant<bool, Epi::W == 4>{});
                else epilogue(std::false_type{});
            }
            if (grp == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            c_kt = 0;
            ++c_tile;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (dummy) DMAs must not outlive the workgroup's LDS
}

template <class XL, class Epi, int NI = 4>
static hipError_t gemm_bf16s_launch(const XL& xl, const __bf16* Wb, long ldw, const Epi& epi, int M, int N, int K, int num_cu, hipStream_t stream) {
    using Cfg = SCfg;
    constexpr int BN = 64 * NI;
    if (M <= 0) return hipSuccess;
    if (N % BN != 0 || K % Cfg::BK != 0 || ldw % 8 != 0) return hipErrorInvalidValue;
    auto kern = gemm_bf16s_kernel<XL, Epi, NI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / BN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xl, Wb, ldw, epi, M, N, K, tiles_m, tiles_n, 1);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- split-K for few-row GEMMs
// out = epi(sum over splits of slab[m][split * N + n]) in a fixed order (bitwise reproducible); Epi is one of the W = 4 (fp32 output)
// epilogues above, used through col() / emit() exactly as the GEMM's own epilogue would
template <class Epi>
static __global__ __launch_bounds__(256) void splitk_rows_reduce_kernel(const float* __restrict__ slab, Epi epi, int M, int N, int ksplit) {
    constexpr int W = Epi::W;                     // columns per thread: 4 (fp32 outputs) or 8 (bf16 outputs), as in the GEMM's own epilogue
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = N / W;
    if (i >= (long)M * nw) return;
    const int m = (int)(i / nw), n = (int)(i - (long)m * nw) * W;
    const float* p = slab + (long)m * N * ksplit + n;
    float v[W];
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        f32x4 acc = *(const f32x4*)(p + 4 * q);
        for (int sp = 1; sp < ksplit; ++sp) acc += *(const f32x4*)(p + (long)sp * N + 4 * q);
        v[4 * q] = acc[0]; v[4 * q + 1] = acc[1]; v[4 * q + 2] = acc[2]; v[4 * q + 3] = acc[3];
    }
    const typename Epi::Aux ax = epi.fetch(m, n);
    epi.emit(v, epi.col(n), ax, m, n);
}
// chooses the split count: enough column tiles to fill the chip, at least 8 K-tiles per split, the slab within slab_floats; 1 = not worth it
static inline int gemm_bf16s_ksplit(int M, int N, int K, int num_cu, size_t slab_floats) {
    if (N % 256 != 0 || M <= 0) return 1;
    const int tiles = ((M + SCfg::BM - 1) / SCfg::BM) * (N / 256);
    if (tiles * 2 > num_cu) return 1;
    int sp = num_cu / tiles;
    while (sp > 1 && (K % (sp * SCfg::BK) != 0 || K / sp < 8 * SCfg::BK || (size_t)M * N * sp > slab_floats)) --sp;
    return sp;
}
template <class XL, class Epi>
static hipError_t gemm_bf16s_splitk_launch(const XL& xl, const __bf16* Wb, long ldw, const Epi& epi, float* slab, int ksplit, int M, int N, int K, int num_cu,
                                           hipStream_t stream) {
    using Cfg = SCfg;
    if (M <= 0) return hipSuccess;
    if (N % 256 != 0 || ksplit < 2 || K % (ksplit * Cfg::BK) != 0 || ldw % 8 != 0) return hipErrorInvalidValue;
    auto kern = gemm_bf16s_kernel<XL, SEpiRawF32, 4>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / 256;
    const int ntiles = tiles_m * tiles_n * ksplit;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xl, Wb, ldw, SEpiRawF32{slab, (long)N * ksplit}, M, N * ksplit, K / ksplit,
                       tiles_m, tiles_n, ksplit);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long items = (long)M * (N / Epi::W);
    hipLaunchKernelGGL(splitk_rows_reduce_kernel<Epi>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, (const float*)slab, epi, M, N, ksplit);
    return hipGetLastError();
}
