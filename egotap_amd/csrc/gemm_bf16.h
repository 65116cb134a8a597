// GEMM on the gfx950 bf16 matrix cores with fp32 operands in HBM:  C[M,N] = epi( A[M,K] * W[N,K]^T ).
//
// Same loaders, epilogues, 256x256 tile, persistent XCD-aware tile walk and LDS-transposed dwordx4 C stores as
// gemm_f32_persist_kernel (gemm_f32.h); what changes is the arithmetic.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of
// the exact-f32 MFMA (2.5 PFLOP/s dense against 157 TFLOP/s), so an fp32 product can be bought cheaper as a SUM of bf16
// products with fp32 accumulation:
//
//   NP = 3  ("bf16x3")  x = hi + lo with hi = bf16(x), lo = bf16(x - hi)  (16 significant bits kept per operand),
//                       a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi           (dropped terms <= 2^-16 |a||b|)
//                       3 bf16 MFMAs per 16 k  = 96 matrix-pipe cycles against 512 for 8 v_mfma_f32_32x32x2_f32.
//   NP = 1  ("bf16")    plain bf16 operands (round to nearest even), fp32 accumulate: the training configurations.
//
// Operands stay fp32 in HBM (no second copy of any tensor, weights are the caller's live nn.Parameters): the split /
// rounding happens in registers between the global load and the LDS write.  An LDS row of either operand holds two
// 32-byte groups + 16 bytes of padding (80 bytes: ds_read_b128 / ds_write_b128 of 16 consecutive rows touch 16 distinct
// 16-byte bank groups):  NP = 3: group 0 = hi(k0..15), group 1 = lo(k0..15), slab depth 16;
//                        NP = 1: group 0 = k0..15,     group 1 = k16..31,    slab depth 32.
// A lane's fragment of one 32x32x16 MFMA is the 16 bytes (8 bf16) at  row*80 + group*32 + (lane>>5)*16.
//
// Pipeline per slab and wave (64 x 128 sub-tile = 2 x 4 MFMA tiles): the A fragments of a slab live in registers, the B
// fragments of tile column j+1 are read while column j multiplies; global loads run FOUR slabs ahead in two register
// sets (a slab lasts ~0.6 us here, a fifth of the fp32 kernel's, so one set in flight no longer covers the L2 latency);
// one barrier per slab.  Register budget is the constraint (8 waves = 256 VGPRs each: 128 accumulators, 32 staging,
// 32 fragments): a spill reload inside the slab loop is a vector-memory operation that the in-order vmcnt makes wait
// for every older global load.
#pragma once
#include "gemm_f32.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NP_, int SCHED_ = 0>
struct BfCfg {
    static constexpr int NP = NP_, SCHED = SCHED_;
    static constexpr int BM = 256, BN = 256, WM = 4, WN = 2, MINW = 2;
    static constexpr int BK = NP_ == 1 ? 32 : 16;          // floats of k per slab
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int LDR = 20;                          // LDS row stride in dwords (80 bytes)
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int NS = 3;                            // LDS slabs
    static constexpr int V4 = BK / 8;                       // float4 loads per thread per operand per slab (thread = row x group)
    static constexpr int LDS_BYTES = NS * (BM + BN) * LDR * 4;
    static_assert(NP_ == 1 || NP_ == 3, "NP: 1 (bf16) or 3 (bf16x3 split)");
    static_assert(BM == 4 * 64 && BN == 2 * 128 && THREADS == 2 * BM && BM == BN, "staging assigns one (row, group) per thread");
};

// 8 consecutive floats -> 8 bf16 (round to nearest even)
__device__ __forceinline__ bf16x8 bf16_round8(const f32x4& x0, const f32x4& x1) {
    bf16x8 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (__bf16)x0[i];
        h[4 + i] = (__bf16)x1[i];
    }
    return h;
}
// 8 consecutive floats -> hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void bf16_split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h0 = (__bf16)x0[i], h1 = (__bf16)x1[i];
        hi[i] = h0;
        hi[4 + i] = h1;
        lo[i] = (__bf16)(x0[i] - (float)h0);
        lo[4 + i] = (__bf16)(x1[i] - (float)h1);
    }
}

template <class Cfg, class ALoad, class Epi, class WMat = SegMat>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void gemm_bf16_persist_kernel(
    ALoad al, WMat W, Epi epi, float* C, long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr bool WB = std::is_same<WMat, SegMatB>::value;      // W already bf16 (NP = 1 only): loaded as 2 x 16 bytes, no conversion
    static_assert(!WB || Cfg::NP == 1, "bf16 weights make sense for the plain-bf16 arithmetic only");
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDR = Cfg::LDR, NS = Cfg::NS, NP = Cfg::NP;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, V4 = Cfg::V4, ELD = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [NS][BM][LDR]
    float* Bs = smem + NS * BM * LDR;                   // [NS][BN][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float* Es = smem + NS * (BM + BN) * LDR + wid * 32 * ELD;      // this wave's 32 x 32 transpose patch
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;
    // staging: thread -> (row sr of the 256-row tile, group sg); rows on the low lane bits (conflict-free b128 writes)
    const int sr = wid * 32 + l31, sg = lh;

    // tiles of this block (same walk as gemm_f32_persist_kernel)
    const int ntiles = tiles_m * tiles_n, nb = gridDim.x, x8 = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (x8 < (nb & 7) ? 1 : 0);
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int lo_t = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8;
    const int cnt = q8 + (x8 < r8 ? 1 : 0);
    const int my_n = cnt > jb ? (cnt - jb + nbx - 1) / nbx : 0;
    const int KT = K / BK;
    const int total = my_n * KT;
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

    typename ALoad::Row arow;
    decltype(W.row(0)) brow;
    int l_tile = 0, l_kt = 0;           // load position in this block's slab stream
    auto set_rows = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        arow = al.row(min(tm * BM + sr, M - 1));
        brow = W.row(tn * BN + sr);
    };
    set_rows(0);

    f32x4 ga[2][V4], gb[2][V4];         // two global-load register sets (slab s uses set s & 1)
    bf16x8 fa[2][TM];                   // [group][i]   A fragments of the current slab
    bf16x8 fb[2][2];                    // [j parity][group]
    bf16x8 cv0_, cv1_;                  // converted staging data on its way to LDS
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = (wm * (TM * 32) + l31) * LDR + 4 * lh;      // dwords
    const int b_off = (wn * (TN * 32) + l31) * LDR + 4 * lh;
    const int s_off = sr * LDR;

#define GLOAD(SET)                                                                                   \
    {                                                                                                \
        const int k0_ = l_kt * BK + sg * (BK / 2);                                                   \
        _Pragma("unroll") for (int v = 0; v < V4; ++v) ga[SET][v] = al.load(arow, k0_ + 4 * v);      \
        if (WB) {                                                                                    \
            gb[SET][0] = *(const f32x4*)(brow + k0_);                                                \
            gb[SET][1] = *(const f32x4*)(brow + k0_ + 8);                                            \
        } else {                                                                                     \
            _Pragma("unroll") for (int v = 0; v < V4; ++v) gb[SET][v] = *(const f32x4*)(brow + k0_ + 4 * v); \
        }                                                                                            \
        if (++l_kt == KT) {                                                                          \
            l_kt = 0;                                                                                \
            if (++l_tile < my_n) set_rows(l_tile);                                                   \
        }                                                                                            \
    }
    // registers -> bf16, half H (0 / 1) of this thread's floats of one operand.  NP = 3: floats 4H..4H+3 of its 8 give
    // elements 4H.. of hi (cv0_) and lo (cv1_); NP = 1: float4 pair H of its 16 floats gives cv0_ (H = 0) or cv1_ (H = 1).
#define CONV_HALF(G, H)                                                                              \
    {                                                                                                \
        if (NP == 3) {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
                const __bf16 h_ = (__bf16)G[H][e];                                                   \
                cv0_[4 * (H) + e] = h_;                                                              \
                cv1_[4 * (H) + e] = (__bf16)(G[H][e] - (float)h_);                                   \
            }                                                                                        \
        } else if ((H) == 0) {                                                                       \
            cv0_ = bf16_round8(G[0], G[1]);                                                          \
        } else {                                                                                     \
            cv1_ = bf16_round8(G[V4 - 2], G[V4 - 1]);                                                \
        }                                                                                            \
    }
#define CONV_HALF_B(G, H)                                                                            \
    {                                                                                                \
        if (WB) {                                                                                    \
            if ((H) == 0) cv0_ = __builtin_bit_cast(bf16x8, G[0]); else cv1_ = __builtin_bit_cast(bf16x8, G[1]); \
        } else CONV_HALF(G, H)                                                                       \
    }
    // ... and into LDS: NP = 3: hi -> group 0, lo -> group 1, each at half sg; NP = 1: the two halves of group sg
#define CWRITE(BASE)                                                                                 \
    {                                                                                                \
        float* p_ = (BASE) + s_off;                                                                  \
        if (NP == 3) {                                                                               \
            *(bf16x8*)(p_ + 4 * sg) = cv0_;                                                          \
            *(bf16x8*)(p_ + 8 + 4 * sg) = cv1_;                                                      \
        } else {                                                                                     \
            *(bf16x8*)(p_ + 8 * sg) = cv0_;                                                          \
            *(bf16x8*)(p_ + 8 * sg + 4) = cv1_;                                                      \
        }                                                                                            \
    }
#define LSTORE(SET, BUF)                                                                             \
    {                                                                                                \
        CONV_HALF(ga[SET], 0) CONV_HALF(ga[SET], 1) CWRITE(As + (BUF) * BM * LDR)                    \
        CONV_HALF_B(gb[SET], 0) CONV_HALF_B(gb[SET], 1) CWRITE(Bs + (BUF) * BN * LDR)                    \
    }
#define AFRAG(BUF, I)                                                                                \
    {                                                                                                \
        const float* p_ = As + (BUF) * BM * LDR + a_off + (I) * 32 * LDR;                            \
        fa[0][I] = *(const bf16x8*)(p_);                                                             \
        fa[1][I] = *(const bf16x8*)(p_ + 8);                                                         \
    }
#define BFRAGS(PAR, BUF, J)                                                                          \
    {                                                                                                \
        const float* p_ = Bs + (BUF) * BN * LDR + b_off + (J) * 32 * LDR;                            \
        fb[PAR][0] = *(const bf16x8*)(p_);                                                           \
        fb[PAR][1] = *(const bf16x8*)(p_ + 8);                                                       \
    }
#define MFMAS_I(I, BPAR, J)                                                                          \
    {                                                                                                \
        acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][I], fb[BPAR][0], acc[I][J], 0, 0, 0); \
        if (NP == 3) {                                                                               \
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][I], fb[BPAR][1], acc[I][J], 0, 0, 0); \
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][I], fb[BPAR][0], acc[I][J], 0, 0, 0); \
        } else {                                                                                     \
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][I], fb[BPAR][1], acc[I][J], 0, 0, 0); \
        }                                                                                            \
    }
    // pattern for the scheduler inside one fenced region: NM times (1 MFMA, NV VALU)
#define INTERLEAVE(NM, NV)                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < (NM); ++q_) {                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                          \
    }

    // prologue: slabs 0 and 1 staged, slabs 2 and 3 in flight, first fragments of slab 0 in registers.
    // GLOAD / LSTORE are UNCONDITIONAL everywhere: past the end of this block's slab stream the loader re-reads slabs of
    // its last tile (valid memory) and the LDS write lands in a slab nobody reads.  With `if (gs + 2 < total)` guards the
    // compiler's waitcnt pass loses count at the control-flow joins and drains EVERY outstanding load (s_waitcnt
    // vmcnt(0)) before each LDS write.
    GLOAD(0)
    GLOAD(1)
    LSTORE(0, 0)
    LSTORE(1, 1)
    GLOAD(0)
    GLOAD(1)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) AFRAG(0, i)
    BFRAGS(0, 0, 0)

    int buf = 0, c_tile = 0, c_kt = 0;
    auto epilogue = [&]() __attribute__((always_inline)) {
        // identical to gemm_f32_persist_kernel: accumulators -> per-wave LDS patch -> row-major float4 -> epilogue -> dwordx4
        int tm, tn;
        tile_of(c_tile, tm, tn);
        const int er = lane >> 3, ec = (lane & 7) * 4;
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
                    const bool ok = mb + s4 * 8 < M;
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (ok) o = epi.apply4(*(const f32x4*)(Es + (s4 * 8 + er) * ELD + ec), cc, rs[s4], mb + s4 * 8, n0);
                    if (more) rs[s4] = epi.res4(min(mb0 + in * 32 + s4 * 8, M - 1), nb0 + jn * 32);
                    if (ok) *(f32x4*)(C + (long)(mb + s4 * 8) * ldc + n0) = o;
                }
            }
        }
    };

    // One slab.  The A fragments are single buffered: row block i of the NEXT slab is re-read into the same registers right
    // after the last MFMAs that use them have issued (tile column TN-1), under the MFMAs of the other row block / the
    // first ones of the next slab.  SCHED = 1: the staged slab gs+2 is converted under the MFMAs of the last two tile
    // columns (A operand under column TN-2, W operand under TN-1), ~4 VALU per MFMA -- an MFMA holds the issue port 8 of
    // its 32 cycles.  Done as one block mid-slab (SCHED = 0) the two co-resident waves of a SIMD convert at the same time
    // and the matrix pipe idles (bf16x3: 303 -> 355 TFLOP/s); plain bf16 is load-bound and prefers the early request.
    auto body = [&](auto PT, int gs) __attribute__((always_inline)) {
        constexpr int P = decltype(PT)::value;          // parity of gs: global-load register set of slabs gs+2 / gs+4
        constexpr int NMI = NP == 3 ? 3 : 2, NV = NP == 3 ? 4 : 2;
        const int b1 = buf + 1 >= NS ? buf + 1 - NS : buf + 1;
        const int b2 = b1 + 1 >= NS ? b1 + 1 - NS : b1 + 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) BFRAGS((j + 1) & 1, buf, j + 1) else BFRAGS(0, b1, 0)   // next column / next slab (published)
            __builtin_amdgcn_sched_barrier(0);
            const bool conv = Cfg::SCHED == 1 && j >= TN - 2;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                MFMAS_I(i, j & 1, j)
                if (conv) {
                    if (j == TN - 2) CONV_HALF(ga[P], i) else CONV_HALF_B(gb[P], i)
                    INTERLEAVE(NMI, NV)
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j == TN - 1) {
                    AFRAG(b1, i)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (Cfg::SCHED == 1) {
                if (j == TN - 2) CWRITE(As + b2 * BM * LDR)
                if (j == TN - 1) {
                    CWRITE(Bs + b2 * BN * LDR)
                    GLOAD(P)                            // request slab gs+4 into the registers just freed
                }
                if (j >= TN - 2) __builtin_amdgcn_sched_barrier(0);
            } else if (j == TN / 2 - 1) {
                LSTORE(P, b2)                           // slab gs+2 (requested two slabs ago) -> the free LDS slab
                GLOAD(P)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf = b1;
        if (++c_kt == KT) {
            epilogue();
            c_kt = 0;
            ++c_tile;
        }
    };
    int gs = 0;
    for (; gs + 1 < total; gs += 2) {
        body(std::integral_constant<int, 0>{}, gs);
        body(std::integral_constant<int, 1>{}, gs + 1);
    }
    if (gs < total) body(std::integral_constant<int, 0>{}, gs);
#undef GLOAD
#undef CONV_HALF
#undef CONV_HALF_B
#undef CWRITE
#undef LSTORE
#undef AFRAG
#undef BFRAGS
#undef MFMAS_I
#undef INTERLEAVE
}

// f32 -> bf16 (round to nearest even), 8 elements per thread: the scratch copy of W for the plain-bf16 GEMM
static __global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    *(bf16x8*)(dst + i * 8) = bf16_round8(*(const f32x4*)(src + i * 8), *(const f32x4*)(src + i * 8 + 4));
}

template <class Cfg, class ALoad, class Epi, class WMat = SegMat>
static hipError_t gemm_bf16_persist_launch(const ALoad& al, const WMat& W, const Epi& epi, float* C, long ldc, int M, int N,
                                           int K, int num_cu, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0) return hipErrorInvalidValue;
    auto kern = gemm_bf16_persist_kernel<Cfg, ALoad, Epi, WMat>;
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
