// [r4] bf16-STORAGE GEMM, 64-deep K-tiles:  OUT = epi( X[M,K] * W[N,K]^T ), X and W plain row-major bf16 -- the ViT's nn.Linear forward and
// input-gradient products (modeling_vit.py:226-230, 271, 319-344), which are 95 % of the NT GEMM time of the bf16 training step.  Same
// tile, wave layout, MFMA shape, accumulator layout, tile walk and epilogue functors as gemm_bf16s_kernel (gemm_bf16s.h); what changed
// is the operand pipeline (profiles/r03_gemm_ablation.md, column 3: fetching 128-byte row segments instead of 64-byte ones is worth
// +4-7 % on every shape; the round-3 verdict's item 1a):
//
//   * K-tile = 64 k.  An LDS row is the 128 contiguous bytes of one operand row: one global_load_lds_dwordx4 wave instruction moves
//     8 rows x 128 B (BK = 32: 16 rows x 64 B), i.e. whole 128-byte lines.  The 16-byte chunk c of local row r sits at chunk position
//     c ^ ((r >> 1) & 7): the 16 lanes a ds_read_b128 serves together (16 consecutive rows, one logical chunk) then touch 16 distinct
//     16-byte bank groups of the 256-byte bank row.  The permutation is applied on the global side (which chunk a lane fetches); the
//     DMA itself writes lane i at base + 16 i.
//   * LDS holds TWO K-tiles of four 16 KB parts each (128 rows x 128 B):
//        XA0 = rows a0 (the first 64 of each wave group's 128 rows), XA1 = rows a1, WB0 / WB1 = the first / second 32 of each wave's 64 columns.
//     A K-tile is four PHASES of 16 MFMAs per wave, one output QUADRANT each, over the whole K = 64:
//        phase 0: a0 x b0   (reads XA0: 8 fragments, WB0: 4)      phase 1: a0 x b1   (reads WB1: 4)
//        phase 2: a1 x b1   (reads XA1: 8)                        phase 3: a1 x b0   (no reads: b0 stayed in registers)
//     so every part is read in exactly ONE phase and its region is free again right after it -- with only two K-tiles of LDS a part is
//     then requested five to six phases ahead of its first use (the BK = 32 ring's distance), not one to three:
//        phase 4T+0 issues WB1(T+1)    4T+1: XA1(T+1)    4T+2: XA0(T+2)    4T+3: WB0(T+2)
//     Write-after-read: the regions were last read in phases 4T-3, 4T-2, 4T, 4T -- at least two phases back, what two wave groups one
//     barrier apart need (gemm_bf16s.h).  Read-after-write: every phase ends [own DMA; own fragment reads; vmcnt(8)] barrier: the four newest
//     parts may stay in flight, the part read in the NEXT phase was issued five or six phases ago.
//   * 64 fragment registers (XA: 8 x 4, WB0 / WB1: 4 x 4 each) + 128 accumulators; the X / W operand addresses are a wave-uniform 64-bit
//     base (scalar unit) + one 32-bit lane offset per row block (global_load_lds ..., s[base] form): no per-lane pointer arithmetic.
//   * Epilogue, store allowance after an exact epilogue, staggered start: as gemm_bf16s_kernel.
#pragma once
#include <type_traits>

#include "gemm_bf16s.h"
#include "lds_dma.h"

struct S64Cfg {
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
    static constexpr int ROWB = 128;                      // bytes per LDS row (64 bf16)
    static constexpr int PART = 128 * ROWB;               // 16 KiB
    static constexpr int KBUF = 4 * PART;                 // one K-tile: XA0 | WB0 | WB1 | XA1
    static constexpr int O_XA0 = 0, O_WB0 = PART, O_WB1 = 2 * PART, O_XA1 = 3 * PART;
    static constexpr int EPATCH = 16 * 64 * 4;            // per-wave epilogue patch: 16 rows x 64 fp32
    static constexpr int LDS_BYTES = 2 * KBUF + (THREADS / 64) * EPATCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------------------------------------------- X-operand loaders
// SBASE loaders: row m of the operand is A + m * lda (plain row-major): the DMA address is a wave-uniform base + a 32-bit lane offset.
// Pointer loaders: row(m) -> per-lane state, ptr(row, kt, chunk) -> address of the 8 k of 16-byte chunk `chunk` (0..7) of the row's
// 64-deep K-tile kt (kt wave-uniform: the index arithmetic on it stays on the scalar unit).
struct X64Plain {
    static constexpr bool SBASE = true;
    const __bf16* A;
    long lda;
};
// 3x3 convolution (pad 1, stride 1) on a channels-last bf16 map [Nimg * S * S, Cp], Cp a multiple of 64 (conv_bf16s.h's XConv3 for this
// kernel): k = (ci / 64, tap, ci % 64), so a 64-deep K-tile is ONE tap of one 64-channel slab and row m of it is the 128 contiguous
// bytes of input pixel (y + dy, x + dx), channels 64 s .. 64 s + 63 -- whole lines again; a page of zeros where the tap leaves the image.
struct X64Conv3 {
    static constexpr bool SBASE = false;
    const __bf16* in;
    const __bf16* zero;      // >= 128 bytes of zeros
    int Cp, log2S;
    struct Row { const __bf16* p; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int S = 1 << log2S, x = m & (S - 1), y = (m >> log2S) & (S - 1);
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (y + dy >= 0 && y + dy < S && x + dx >= 0 && x + dx < S) mask |= 1u << t;
        }
        return Row{in + (long)m * Cp, mask};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int kt, int chunk) const {
        const int slab = (kt * 7282) >> 16, tap = kt - 9 * slab;                     // kt / 9 exactly for kt < 7000
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        const int off = ((dy << log2S) + dx) * Cp + 64 * slab;
        return ((r.mask >> tap) & 1u) ? r.p + off + chunk * 8 : zero + chunk * 8;
    }
};

// BasicBlock 3x3 convolution (pad 1, stride 1 or 2) on the backbone's eye-interleaved maps [B * Si * Si, 2 C], Si = So * stride (conv_bf16s.h's
// XConvE for this kernel: image n = 2 b + eye reads the C-channel slice eye * C of rows b * Si * Si + pixel), C a multiple of 64,
// k = (ci / 64, tap, ci % 64)
struct X64ConvE {
    static constexpr bool SBASE = false;
    const __bf16* in;
    const __bf16* zero;      // >= 128 bytes of zeros
    int C, log2So, stride;
    struct Row { const __bf16* p; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int So = 1 << log2So, xo = m & (So - 1), yo = (m >> log2So) & (So - 1), n = m >> (2 * log2So);
        const int Si = So * stride, yi = yo * stride, xi = xo * stride;
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (yi + dy >= 0 && yi + dy < Si && xi + dx >= 0 && xi + dx < Si) mask |= 1u << t;
        }
        return Row{in + (((long)(n >> 1) * Si * Si + (long)yi * Si + xi) * 2 + (n & 1)) * C, mask};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int kt, int chunk) const {
        const int slab = (kt * 7282) >> 16, tap = kt - 9 * slab;                     // kt / 9 exactly for kt < 7000
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        const int off = (dy * (stride << log2So) + dx) * (2 * C) + 64 * slab;
        return ((r.mask >> tap) & 1u) ? r.p + off + chunk * 8 : zero + chunk * 8;
    }
};

// [r5] SCALAR-ORIGIN forms of the two convolution loaders: the same rows, taps and zero page, addressed as ONE wave-uniform 64-bit origin
// (at or below everything the loader touches: the map's first pixel minus one image row + 1 pixels, or the zero page, whichever is lower) + a 32-bit
// lane offset -- the global_load_lds s[base] form the plain operands use -- instead of a 64-bit pointer per lane picked by a 64-bit select.  The
// launcher takes this form when everything lies inside 4 GB of the origin (every shipped chunk size); same bytes fetched, same bits out.
//   row(m)        -> byte offset of the row's centre pixel (+ eye slice) from the origin, and its 9-bit tap mask
//   ktile(kt,...) -> wave-uniform: signed byte offset of the K-tile's tap / slab from the centre pixel, and the tap
struct X64Conv3S {
    static constexpr bool SBASE = false, SORG = true;
    const __bf16* org;
    unsigned in_off, zero_off;     // bytes from org to the map / to the zero page
    int Cp, log2S;
    struct Row { unsigned off; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int S = 1 << log2S, x = m & (S - 1), y = (m >> log2S) & (S - 1);
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (y + dy >= 0 && y + dy < S && x + dx >= 0 && x + dx < S) mask |= 1u << t;
        }
        return Row{in_off + (unsigned)m * (unsigned)(Cp * 2), mask};
    }
    __device__ __forceinline__ void ktile(int kt, int& koff, int& tap) const {
        const int slab = (kt * 7282) >> 16;
        tap = kt - 9 * slab;
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        koff = (((dy << log2S) + dx) * Cp + 64 * slab) * 2;
    }
};
struct X64ConvES {
    static constexpr bool SBASE = false, SORG = true;
    const __bf16* org;
    unsigned in_off, zero_off;
    int C, log2So, stride;
    struct Row { unsigned off; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int So = 1 << log2So, xo = m & (So - 1), yo = (m >> log2So) & (So - 1), n = m >> (2 * log2So);
        const int Si = So * stride, yi = yo * stride, xi = xo * stride;
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (yi + dy >= 0 && yi + dy < Si && xi + dx >= 0 && xi + dx < Si) mask |= 1u << t;
        }
        return Row{in_off + (((unsigned)(n >> 1) * (unsigned)(Si * Si) + (unsigned)(yi * Si + xi)) * 2u + (unsigned)(n & 1)) * (unsigned)(C * 2), mask};
    }
    __device__ __forceinline__ void ktile(int kt, int& koff, int& tap) const {
        const int slab = (kt * 7282) >> 16;
        tap = kt - 9 * slab;
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        koff = ((dy * (stride << log2So) + dx) * (2 * C) + 64 * slab) * 2;
    }
};
// [r5] the fc1 gathers in the same form: a row's byte offset from the tensor + a wave-uniform offset per K-tile (every lane valid: mask 1, tap 0)
struct X64TokensS {
    static constexpr bool SBASE = false, SORG = true;
    const __bf16* org;               // = Y
    unsigned zero_off;               // unused (no lane is ever invalid)
    int T, D, seq, side, ppd, grid;
    struct Row { unsigned off; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / T, i = m - b * T;
        return Row{(unsigned)(((long)b * seq + (long)(ppd * (i / grid)) * side + ppd * (i % grid)) * D * 2), 1u};
    }
    __device__ __forceinline__ void ktile(int kt, int& koff, int& tap) const {
        const int k0 = kt * 64, s = k0 / D, c = k0 - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        koff = ((prl * side + pcl) * D + c) * 2;
        tap = 0;
    }
};
struct X64RotS {
    static constexpr bool SBASE = false, SORG = true;
    const __bf16* org;               // = hm
    unsigned zero_off;
    int C, J, HW;
    struct Row { unsigned off; unsigned mask; };
    __device__ __forceinline__ Row row(int m) const {
        const int T = 2 * J;
        const int b = m / T, t = m - b * T;
        const int eye = t / J, j = t - eye * J;
        return Row{(unsigned)((long)(b * C + 2 * J + eye * 2 * J + j) * HW * 2), 1u};
    }
    __device__ __forceinline__ void ktile(int kt, int& koff, int& tap) const {
        const int k0 = kt * 64, cs = k0 / HW;
        koff = (cs * J * HW + (k0 - cs * HW)) * 2;
        tap = 0;
    }
};
template <class XL, class = void> struct s64_sorg { static constexpr bool value = false; };
template <class XL> struct s64_sorg<XL, std::enable_if_t<XL::SORG>> { static constexpr bool value = true; };

// [r5] fc1 of the two encoders on this kernel: rows gathered as gemm_bf16s.h's XTokens / XRot gather them (net_architecture.py:388-406, 690-694).
// D and HW are multiples of 64, so a 64-deep K-tile lies inside one patch token / one map: 128 contiguous bytes per row.
struct X64Tokens {
    static constexpr bool SBASE = false;
    const __bf16* Y;
    int T, D, seq, side, ppd, grid;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / T, i = m - b * T;
        return Row{Y + ((long)b * seq + (long)(ppd * (i / grid)) * side + ppd * (i % grid)) * D};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int kt, int chunk) const {       // kt wave-uniform: scalar arithmetic
        const int k0 = kt * 64, s = k0 / D, c = k0 - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        return r.p + (long)(prl * side + pcl) * D + c + chunk * 8;
    }
};
struct X64Rot {
    static constexpr bool SBASE = false;
    const __bf16* hm;
    int C, J, HW;
    struct Row { const __bf16* p; };
    __device__ __forceinline__ Row row(int m) const {
        const int T = 2 * J;
        const int b = m / T, t = m - b * T;
        const int eye = t / J, j = t - eye * J;
        return Row{hm + (long)(b * C + 2 * J + eye * 2 * J + j) * HW};
    }
    __device__ __forceinline__ const __bf16* ptr(const Row& r, int kt, int chunk) const {
        const int k0 = kt * 64, cs = k0 / HW;
        return r.p + (long)cs * J * HW + (k0 - cs * HW) + chunk * 8;
    }
};

// Epilogues whose per-column constants are just the bias (Col = SBias8 or f32x4 read from `bias`): the kernel stages the wave's 64 bias
// values in its (idle) epilogue patch by one 4-byte LDS DMA per tile, issued in the tile's first phase and counted like every other DMA.
// The epilogue then starts with two LDS reads instead of a global load whose wait -- vmcnt is in order -- drained every DMA in flight
// and exposed a full memory round trip per tile.
template <class E, class C = typename E::Col> struct s64_lds_bias { static constexpr bool value = false; };
template <class E> struct s64_lds_bias<E, SBias8> { static constexpr bool value = true; };
template <class E> struct s64_lds_bias<E, f32x4> { static constexpr bool value = true; };
template <class E> __device__ __forceinline__ const float* s64_bias_ptr(const E& e, std::true_type) { return e.bias; }
template <class E> __device__ __forceinline__ const float* s64_bias_ptr(const E&, std::false_type) { return nullptr; }
template <class C> struct s64_col_lds;
template <> struct s64_col_lds<SBias8> { static __device__ __forceinline__ SBias8 get(const float* p) { return SBias8{*(const f32x4*)p, *(const f32x4*)(p + 4)}; } };
template <> struct s64_col_lds<f32x4> { static __device__ __forceinline__ f32x4 get(const float* p) { return *(const f32x4*)p; } };

template <class XL, bool SB = XL::SBASE> struct S64RowState { };                                   // SBASE loaders keep no per-lane row state
template <class XL> struct S64RowState<XL, false> { typename XL::Row r[2][2]; };                    // [a][g]

// NJ = 16-column MFMA tiles per wave and half: 2 -> the 256 x 256 output tile (default); [r4] 1 -> 256 x 128 for N = 128 (layer2 of the bf16
// estimators ran on the 32-deep kernel's 128-column tile at MFMA busy 0.165: a K-tile there costs what it costs at 256 columns).  The W
// parts shrink to 64 rows (ONE DMA instruction per wave: row block wid), a phase to 8 MFMAs per wave; the X parts, the part offsets in
// LDS, the phase order and the barriers are unchanged.  Any four consecutive issues of the cycle WB1, XA1, XA0, WB0 hold two X and two W
// parts, so the allowance for "the four newest parts" is the constant 4 + 2 NJ.
template <class XL, class Epi, int NJ = 2>
__global__ __launch_bounds__(S64Cfg::THREADS, 2) void gemm_bf16s64_kernel(XL xl, const __bf16* __restrict__ Wb, long ldw, Epi epi,
                                                                           int M, int N, int K, int tiles_m, int tiles_n) {
    using Cfg = S64Cfg;
    static_assert(NJ == 2 || NJ == 1, "n-tiles per wave and half");
    constexpr int BM = Cfg::BM, BN = 128 * NJ, BK = Cfg::BK, ROWB = Cfg::ROWB, PART = Cfg::PART, KBUF = Cfg::KBUF;
    constexpr int WCOLS = 32 * NJ;                   // output columns per wave
    constexpr int VMC = 4 + 2 * NJ;                  // vmcnt allowance: the four newest parts (X: two DMA instructions per wave, W: NJ) may stay in flight
    extern __shared__ __attribute__((aligned(16))) char smem_s64[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wc = wid & 3;
    float* Es = (float*)(smem_s64 + 2 * KBUF + wid * Cfg::EPATCH);

    // tiles of this workgroup: XCD-aware chunk of the grouped tile order (as gemm_bf16s_kernel)
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

    // ---- DMA duty of this wave per part: row blocks wid and wid + 8 of the part's 16 (8 rows x 128 B each).  Lane -> row lane >> 3 of
    // the block, chunk position lane & 7, which holds logical chunk (lane & 7) ^ ((r >> 1) & 7), r = 8 blk + (lane >> 3) the local row:
    // (r >> 1) & 7 = (4 blk + (lane >> 4)) & 7, the same for blk = wid and wid + 8.
    const int drow = lane >> 3, dchunk = (lane & 7) ^ ((4 * wid + (lane >> 4)) & 7);
    // X parts: local row r of XA_a -> tile row (r >> 6) * 128 + a * 64 + (r & 63): block wid is group 0's rows, block wid + 8 group 1's.
    // W parts: local row r of WB_b -> tile column (r >> 5) * 64 + b * 32 + (r & 31): block wid -> wave column wid >> 2, block wid + 8 -> 2 + (wid >> 2).
    // Lane offsets in bytes from the wave-uniform base of the part (rows past M re-read row M - 1: their results are never stored).
    unsigned xo[2][2];                               // [a][g]  (SBASE loaders: lane offsets)
    S64RowState<XL> xrow;                            //         (pointer loaders: per-lane row state)
    // (NJ = 1: local row r of WB_b -> tile column (r >> 4) * 32 + b * 16 + (r & 15): block wid -> wave column wid >> 1, rows 8 (wid & 1) ..)
    const unsigned wo = (unsigned)((8 * (NJ == 2 ? (wid & 3) : (wid & 1)) + drow) * ldw * 2 + dchunk * 16);
    struct Stream { int tile, kt; };                 // position of an issue stream: (tile index of this workgroup, K-tile inside it)
    Stream s_xa[2] = {{0, 0}, {0, 0}}, s_wb[2] = {{0, 0}, {0, 0}};
    unsigned long long xbase[2], wbase[2];           // wave-uniform bases of the streams' current tiles (bytes)
    auto uniform64 = [](unsigned long long v) __attribute__((always_inline)) { return lds_dma_base(v); };      // lds_dma.h
    auto set_x = [&](int a, int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        const int r0 = tm * BM + a * 64 + 8 * wid + drow;      // group 0's row of this lane; group 1's is 128 further
        if constexpr (XL::SBASE) {
            xbase[a] = uniform64((unsigned long long)(size_t)xl.A + (unsigned long long)tm * BM * xl.lda * 2);
#pragma unroll
            for (int g = 0; g < 2; ++g) xo[a][g] = (unsigned)((long)(min(r0 + g * 128, M - 1) - tm * BM) * xl.lda * 2 + dchunk * 16);
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) xrow.r[a][g] = xl.row(min(r0 + g * 128, M - 1));
        }
    };
    auto set_w = [&](int b, int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        wbase[b] = uniform64((unsigned long long)(size_t)Wb + ((unsigned long long)(tn * BN + (NJ == 2 ? (wid >> 2) * 64 + b * 32 : (wid >> 1) * 32 + b * 16)) * ldw) * 2);
    };
    set_x(0, 0); set_x(1, 0);
    set_w(0, 0); set_w(1, 0);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_s64;
    // inline asm: the compiler's waitcnt pass would drain vmcnt(0) before every LDS read after __builtin_amdgcn_global_load_lds;
    // the waits are counted by hand below (a constant number of DMA instructions per phase, unconditionally)
    auto dma1 = [&](unsigned voff, unsigned long long sbase, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    auto advance = [&](Stream& st, int which, bool is_x) __attribute__((always_inline)) {
        if (__builtin_expect(st.tile < my_n && ++st.kt == KT, 0)) {          // (rare: the common path falls through without a taken branch)
            st.kt = 0;
            if (++st.tile < my_n) { if (is_x) set_x(which, st.tile); else set_w(which, st.tile); }
            else st.kt = KT - 1;                     // stream exhausted: keep re-reading the last K-tile into a region nobody reads
        }
    };
    unsigned long long xorg = 0;                     // scalar-origin loaders: the origin, and this lane's offset into the zero page
    unsigned zoff = 0;
    const unsigned dch16 = (unsigned)dchunk * 16;
    if constexpr (s64_sorg<XL>::value) { xorg = uniform64((unsigned long long)(size_t)xl.org); zoff = xl.zero_off + dch16; }
    auto dma1v = [&](const __bf16* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    auto issue_x = [&](int a, int buf) __attribute__((always_inline)) {        // XA_a of the stream's next K-tile -> K-tile buffer buf
        const unsigned sa = lds0 + buf * KBUF + (a ? Cfg::O_XA1 : Cfg::O_XA0) + wid * 1024;
        if constexpr (XL::SBASE) {
            const unsigned long long kb = xbase[a] + (unsigned long long)s_xa[a].kt * (BK * 2);
            dma1(xo[a][0], kb, sa);
            dma1(xo[a][1], kb, sa + 8 * 1024);
        } else if constexpr (s64_sorg<XL>::value) {
            int koff, tap;
            xl.ktile(s_xa[a].kt, koff, tap);                                   // scalar unit
            const unsigned kc = (unsigned)koff + dch16;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const typename XL::Row& r = xrow.r[a][g];
                dma1(((r.mask >> tap) & 1u) ? r.off + kc : zoff, xorg, sa + g * 8 * 1024);
            }
        } else {
            dma1v(xl.ptr(xrow.r[a][0], s_xa[a].kt, dchunk), sa);
            dma1v(xl.ptr(xrow.r[a][1], s_xa[a].kt, dchunk), sa + 8 * 1024);
        }
        advance(s_xa[a], a, true);
    };
    auto issue_w = [&](int b, int buf) __attribute__((always_inline)) {
        const unsigned sa = lds0 + buf * KBUF + (b ? Cfg::O_WB1 : Cfg::O_WB0) + wid * 1024;
        const unsigned long long kb = wbase[b] + (unsigned long long)s_wb[b].kt * (BK * 2);
        dma1(wo, kb, sa);
        if constexpr (NJ == 2) dma1(wo, kb + (unsigned long long)128 * ldw * 2, sa + 8 * 1024);
        advance(s_wb[b], b, false);
    };

    f32x4 acc[8][2 * NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment byte offset of this lane inside a 16-row block: row lane & 15, logical chunk 4 s + (lane >> 4) of k-step s
    const int l15 = lane & 15;
    const int foff0 = l15 * ROWB + (((lane >> 4) ^ (l15 >> 1)) << 4);           // k-step 0; k-step 1 is the chunk position xor 4: offset xor 64
    const int foff1 = foff0 ^ 64;
    const int xa_base = (grp * 64) * ROWB;            // + i * 16 * ROWB: m-tile i of the half
    const int wb_base = (wc * 16 * NJ) * ROWB;        // + j * 16 * ROWB: n-tile j of the half

    int c_tile = 0;
    // vmcnt is ONE in-order counter for DMA, loads and stores: for the first four phases after an exact epilogue the parts a wait
    // must retire were all issued before the epilogue's NST stores, so the allowance is 8 + NST and the stores stay in flight
    // (derivation in gemm_bf16s.h); after a ragged tile the plain allowance makes the first wait drain them.
    constexpr bool CS = s_epi_colsum<Epi>::value;
    static_assert(!CS || Epi::W == 8, "column sums ride on the bf16-output epilogues");
    static_assert(NJ == 2 || !CS, "column sums: 256-column tiles only");
    constexpr int LPR = (Epi::W == 4 ? 8 : 4) * NJ;  // epilogue: lanes per patch row (W columns each)
    constexpr int IT = LPR / 4, NST = 8 * IT * Epi::STORES + (CS ? 2 : 0);
    constexpr int SLK = s_epi_exact<Epi>::value ? VMC + NST : VMC;
    static_assert(SLK + 1 <= 63, "vmcnt is a 6-bit counter");
    bool slack_on = false;           // the next tile's first K-tile leaves the stores of the epilogue before it in flight
    constexpr bool LB = s64_lds_bias<Epi>::value;
    static_assert(NJ == 2 || !LB, "the bias staging covers 64 columns per wave");
    const float* bias_g = s64_bias_ptr(epi, std::integral_constant<bool, LB>{});
    const bool has_bias = LB && bias_g != nullptr;    // the same for every wave of the launch: the DMA count per phase stays uniform
    const unsigned patch_lds = lds0 + 2 * KBUF + wid * Cfg::EPATCH;
    auto issue_bias = [&]() __attribute__((always_inline)) {      // the 64 bias values of this wave's columns of tile c_tile -> the head of its patch
        int tm, tn;
        tile_of(c_tile, tm, tn);
        // (no bias: the DMA still runs, from any valid 256 bytes -- the count per phase is a compile-time constant -- and is never read)
        const unsigned long long b = has_bias ? uniform64((unsigned long long)(size_t)bias_g + (unsigned long long)(tn * BN + wc * 64) * 4)
                                              : uniform64((unsigned long long)(size_t)Wb);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"((unsigned)(lane * 4)), "s"(b), "s"(__builtin_amdgcn_readfirstlane(patch_lds)) : "memory");
    };
    // per-column constants of this lane: staged in the patch during the tile's first K-tile (LDS operations of a wave stay in order: the
    // epilogue's patch writes cannot overtake these reads), or from global memory for the epilogues with other constants / no bias
    auto epi_cols = [&](int en, int ecol) __attribute__((always_inline)) -> typename Epi::Col {
        if constexpr (LB) {
            if (has_bias) return s64_col_lds<typename Epi::Col>::get(Es + ecol);
        }
        return epi.col(en);
    };
    // FULL (every row of the tile exists): the stores are unconditional, so hipcc counts them exactly and its waits for the aux rows of the
    // NEXT 16-row block (requested before this block's stores) leave this block's stores in flight.  Under `if (m < M)` it had to assume the
    // stores might not have been issued and waited as if only loads followed the ones it needs -- vmcnt is in order: every block then sat
    // until the previous block's stores had completed (eight store round trips per tile in the residual / GELU-grad epilogues).
    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        int tm, tn;
        tile_of(c_tile, tm, tn);
        const int m_wave = tm * BM + grp * 128, n_wave = tn * BN + wc * WCOLS;
        const int q = lane >> 4;
        const int er = lane / LPR, ec = lane % LPR, ecol = ec * Epi::W;
        const int en = n_wave + ecol;
        typename Epi::Col cc = epi_cols(en, ecol);
        typename Epi::Aux ax[IT], an[IT];
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
            for (int ni = 0; ni < 2 * NJ; ++ni) {
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
                    const f32x4 v = *(const f32x4*)(Es + r * 64 + ((ec ^ r) << 2));
                    float vv[4] = {v[0], v[1], v[2], v[3]};
#if defined(EGOTAP_ABL) && (EGOTAP_ABL & 2)      // timing-only: the epilogue without its functor and global stores
                    asm volatile("" ::"v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]));
#else
                    if (FULL || m0 + r < M) epi.emit(vv, cc, ax[it], m0 + r, en);
#endif
                } else {
                    const int c2 = ec;
                    const f32x4 v0 = *(const f32x4*)(Es + r * 64 + (((2 * c2) ^ r) << 2));
                    const f32x4 v1 = *(const f32x4*)(Es + r * 64 + (((2 * c2 + 1) ^ r) << 2));
                    float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#if defined(EGOTAP_ABL) && (EGOTAP_ABL & 2)
                    asm volatile("" ::"v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]), "v"(vv[4]), "v"(vv[5]), "v"(vv[6]), "v"(vv[7]));
                    if (false) {
#else
                    if (FULL || m0 + r < M) {
#endif
                        epi.emit(vv, cc, ax[it], m0 + r, en);              // (leaves the values it stored in vv)
                        if constexpr (CS) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) cs[i] += (float)(__bf16)vv[i];      // the sum of what was STORED (bf16)
                        }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) ax[it] = an[it];
        }
        if constexpr (CS) {
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
        slack_on = s_epi_exact<Epi>::value && (tm + 1) * BM <= M;       // (K >= 128: two K-tiles at least, so the allowance ends before the next epilogue)
    };

    if constexpr (s_epi_stagger<Epi>::value) {       // workgroups start a quarter tile apart (gemm_bf16s.h)
        for (int i = 0; i < (int)((blockIdx.x >> 3) & 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
    // ---- prologue: K-tile 0 complete, then XA0 and WB0 of K-tile 1 -- what the steady-state schedule would have issued by now
    issue_x(0, 0); issue_w(0, 0); issue_w(1, 0); issue_x(1, 0);
    issue_x(0, 1); issue_w(0, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + NJ) : "memory");   // K-tile 0 has landed (the two parts of K-tile 1 may be in flight) ...
    __builtin_amdgcn_s_barrier();                          // ... for every wave: phase 0 may read it
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    bf16x8 xf[4][2], wf0[NJ][2], wf1[NJ][2];
    int buf = 0;
#define S64_PHASE_MFMA(A, WF, B)                                                                  \
    do {                                                                                        \
        __builtin_amdgcn_s_barrier();                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        __builtin_amdgcn_s_setprio(1);                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                      \
                _Pragma("unroll") for (int s = 0; s < 2; ++s)                                   \
                    acc[4 * (A) + i][NJ * (B) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[j][s], xf[i][s], acc[4 * (A) + i][NJ * (B) + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        __builtin_amdgcn_s_barrier();                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
    // One K-tile = four phases.  The vmcnt allowance WC of its four waits is a compile-time constant of the COPY of this body that runs:
    // the steady-state copy (WC = 8, no branch anywhere between a phase's fragment reads and its barrier: per-phase wait selection by
    // branches cost 0.1-0.25 us per K-tile, measured) and two copies for a tile's FIRST K-tile, whose waits also let the bias DMA
    // (one more among the four newest phases) and, after an exact epilogue, that epilogue's stores stay in flight.
    auto ktile = [&](auto waitc, auto first) __attribute__((always_inline)) {
        constexpr int WC = decltype(waitc)::value;
        constexpr bool FIRST = decltype(first)::value;
        const char* kb = smem_s64 + buf * KBUF;
        // ---------------- phase 0: a0 x b0
        issue_w(1, buf ^ 1);
        if constexpr (FIRST && LB) issue_bias();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            wf0[j][0] = *(const bf16x8*)(kb + Cfg::O_WB0 + wb_base + j * 16 * ROWB + foff0);
            wf0[j][1] = *(const bf16x8*)(kb + Cfg::O_WB0 + wb_base + j * 16 * ROWB + foff1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[i][0] = *(const bf16x8*)(kb + Cfg::O_XA0 + xa_base + i * 16 * ROWB + foff0);
            xf[i][1] = *(const bf16x8*)(kb + Cfg::O_XA0 + xa_base + i * 16 * ROWB + foff1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC) : "memory");
        S64_PHASE_MFMA(0, wf0, 0);
        // ---------------- phase 1: a0 x b1
        issue_x(1, buf ^ 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            wf1[j][0] = *(const bf16x8*)(kb + Cfg::O_WB1 + wb_base + j * 16 * ROWB + foff0);
            wf1[j][1] = *(const bf16x8*)(kb + Cfg::O_WB1 + wb_base + j * 16 * ROWB + foff1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC) : "memory");
        S64_PHASE_MFMA(0, wf1, 1);
        // ---------------- phase 2: a1 x b1
        issue_x(0, buf);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[i][0] = *(const bf16x8*)(kb + Cfg::O_XA1 + xa_base + i * 16 * ROWB + foff0);
            xf[i][1] = *(const bf16x8*)(kb + Cfg::O_XA1 + xa_base + i * 16 * ROWB + foff1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC) : "memory");
        S64_PHASE_MFMA(1, wf1, 1);
        // ---------------- phase 3: a1 x b0
        issue_w(0, buf);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC) : "memory");
        S64_PHASE_MFMA(1, wf0, 0);
        buf ^= 1;
    };
    constexpr int XB = LB ? 1 : 0;                      // the bias DMA of a tile's first K-tile (issued whether or not there is a bias: a constant count)
    for (c_tile = 0; c_tile < my_n; ++c_tile) {
        if (slack_on) ktile(std::integral_constant<int, SLK + XB>{}, std::true_type{});
        else ktile(std::integral_constant<int, VMC + XB>{}, std::true_type{});
#if defined(EGOTAP_ABL) && (EGOTAP_ABL & 1)      // timing-only: the store allowance never ends (waits may pass before their data has landed: wrong results)
        for (int kt = 1; kt < KT; ++kt) ktile(std::integral_constant<int, SLK>{}, std::false_type{});
#else
        for (int kt = 1; kt < KT; ++kt) ktile(std::integral_constant<int, VMC>{}, std::false_type{});
#endif
        // one extra barrier per tile and group lets the two groups' epilogues run side by side (gemm_bf16s.h)
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
    }
#undef S64_PHASE_MFMA
    if (grp == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (dummy) DMAs must not outlive the workgroup's LDS
}

// shapes this kernel takes; the caller falls back to gemm_bf16s_launch otherwise
static inline bool gemm_bf16s64_ok(int M, int N, int K, long ldx, long ldw) {
    return M > 0 && N % 256 == 0 && K % 64 == 0 && K >= 128 && ldx % 8 == 0 && ldw % 8 == 0 && (long)256 * (ldx > ldw ? ldx : ldw) * 2 < (1L << 31);
}
template <class XL, class Epi, int NJ = 2>
static hipError_t gemm_bf16s64_launch_x(const XL& xl, const __bf16* Wb, long ldw, const Epi& epi, int M, int N, int K, int num_cu, hipStream_t stream) {
    using Cfg = S64Cfg;
    if (M <= 0) return hipSuccess;
    if (N % (128 * NJ) != 0 || K % 64 != 0 || K < 128 || ldw % 8 != 0 || (long)256 * ldw * 2 >= (1L << 31)) return hipErrorInvalidValue;
    auto kern = gemm_bf16s64_kernel<XL, Epi, NJ>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / (128 * NJ);
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xl, Wb, ldw, epi, M, N, K, tiles_m, tiles_n);
    return hipGetLastError();
}
template <class Epi>
static hipError_t gemm_bf16s64_launch(const __bf16* X, long ldx, const __bf16* Wb, long ldw, const Epi& epi, int M, int N, int K, int num_cu, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    if (!gemm_bf16s64_ok(M, N, K, ldx, ldw)) return hipErrorInvalidValue;
    return gemm_bf16s64_launch_x(X64Plain{X, ldx}, Wb, ldw, epi, M, N, K, num_cu, stream);
}

// Scalar-origin addressing of a convolution operand (X64Conv3S / X64ConvES): the origin is the lower of (map - reach) and the zero page; false when
// the map's end or the zero page lies 4 GB or more above it (the caller keeps the per-lane pointer form).  g_conv_addressing
// (egotap_debug_conv_addressing; defined once, in part 3 of the library): 0 = scalar origin where it fits (default), 1 = per-lane pointers always.
extern int g_conv_addressing;
static inline bool s64_conv_origin(const __bf16* in, size_t bytes, size_t reach, const __bf16* zero, const __bf16*& org, unsigned& in_off, unsigned& zero_off) {
    const unsigned long long a = (unsigned long long)(size_t)in, z = (unsigned long long)(size_t)zero;
    if (g_conv_addressing == 1 || a < reach) return false;
    const unsigned long long lo = a - reach < z ? a - reach : z, hi = a + bytes > z + 256 ? a + bytes : z + 256;
    if (hi - lo >= (1ull << 32) - 4096) return false;
    org = (const __bf16*)(size_t)lo;
    in_off = (unsigned)(a - lo);
    zero_off = (unsigned)(z - lo);
    return true;
}

// Dispatch of a plain-operand NT product: the 64-deep kernel where its shape rules hold, the 32-deep one otherwise (ragged K, narrow
// leading dimensions).  g_gemm_bf16s_bk (egotap_debug_gemm_bk, egotap_debug.h; defined once, in part 3 of the library) pins one of the two
// for same-process A/B timing and for the bit-equality test: both kernels run the same MFMAs in the same k order per output element.
extern int g_gemm_bf16s_bk;          // 0 = choose by shape (default), 32 / 64 = force
template <class Epi>
static hipError_t gemm_bf16s_plain_launch(const XPlain& xl, const __bf16* Wb, long ldw, const Epi& epi, int M, int N, int K, int num_cu, hipStream_t stream) {
    if (g_gemm_bf16s_bk != 32 && gemm_bf16s64_ok(M, N, K, xl.lda, ldw)) return gemm_bf16s64_launch(xl.A, xl.lda, Wb, ldw, epi, M, N, K, num_cu, stream);
    if (g_gemm_bf16s_bk == 64) return hipErrorInvalidValue;
    return gemm_bf16s_launch(xl, Wb, ldw, epi, M, N, K, num_cu, stream);
}
