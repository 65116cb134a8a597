// Weight-gradient GEMM ("TN") on the bf16 matrix cores with fp32 operands in HBM:  dW[N,K] = sum_m dY[m,n] * X[m,k].
// Arithmetic as in gemm_bf16.h: NP = 3 splits every fp32 operand into hi + lo bf16 in registers and takes
// hi*hi + hi*lo + lo*hi (fp32-grade gradients at a third of the bf16 rate), NP = 1 rounds to bf16 (training configs).
//
// The contracted index m is the ROW of both operands in memory, while a v_mfma_f32_32x32x16_bf16 fragment wants 8
// consecutive m for one n (or k) per lane: a transposed read.  The slab of BKM rows of dY and of X is staged exactly as it
// lies in memory (coalesced float4 rows -> bf16 -> ds_write_b64 into a row-major [m][n] image) and the fragments come
// back through ds_read_b64_tr_b16, gfx950's 4 x 16 block transpose read (cdna_hip_programming.md T10): the 16-lane group
// of lanes with the same (lane >> 4) reads rows r0..r0+3 x 16 columns and every lane receives its column's 4 rows, so
// two reads (r0 = 8h, 8h+4) are one fragment (lane half h holds m = 8h..8h+7 of the 16-m step).  Row stride 576 bytes:
// the four rows of a block sit 64 bytes apart modulo 256, so the 32 lanes of a half read 256 distinct bytes (no bank
// conflict); the instruction needs EXEC all ones -- there is no divergent code around the reads.
// M is split over blocks and the partial [N,K] slabs are summed in a fixed order by reduce_slabs_kernel, exactly like
// gemm_tn_f32.h (bitwise reproducible, no float atomics).
#pragma once
#include "gemm_bf16.h"
#include "gemm_tn_f32.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

template <int NP_>
struct TnBfCfg {
    static constexpr int NP = NP_, NIMG = NP_ == 3 ? 2 : 1;      // images per operand (hi, lo)
    static constexpr int BN = 256, BK = 256, BKM = 32, WN = 4, WK = 2;
    static constexpr int THREADS = 64 * WN * WK;
    static constexpr int STR = 288;                               // image row stride in bf16 (576 bytes)
    static constexpr int TN = BN / WN / 32, TK = BK / WK / 32;
    static constexpr int V4 = BKM * BN / 4 / THREADS;             // float4 per thread per operand per slab
    static constexpr int IMG = BKM * STR;                         // bf16 per image
    static constexpr int LDS_BYTES = 2 * 2 * NIMG * IMG * 2;      // 2 buffers x 2 operands x images
    static_assert(BN == BK && (BKM * BN / 4) % THREADS == 0, "staging must divide evenly");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class Cfg, class XLoad>
__global__ __launch_bounds__(Cfg::THREADS, 2) void gemm_tn_bf16_kernel(const float* __restrict__ dY, long ldy, XLoad xl,
                                                                         float* __restrict__ slabs, int M, int N, int K,
                                                                         int tiles_n, int tiles_k, int splits, float* __restrict__ csum) {
    constexpr int BN = Cfg::BN, BK = Cfg::BK, BKM = Cfg::BKM, STR = Cfg::STR, IMG = Cfg::IMG, NIMG = Cfg::NIMG, NP = Cfg::NP;
    constexpr int TN = Cfg::TN, TK = Cfg::TK, V4 = Cfg::V4, THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) __bf16 simg[];
    // image (buf, operand, part): operand 0 = dY rows, 1 = X rows; part 0 = hi, 1 = lo
    auto img = [&](int buf, int op, int part) __attribute__((always_inline)) { return simg + ((buf * 2 + op) * NIMG + part) * IMG; };
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / Cfg::WK, wk = wid % Cfg::WK;
    const int l31 = lane & 31, lh = lane >> 5;

    const int bid = blockIdx.x;
    const int split = bid % splits, tile = bid / splits;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int n0 = tn * BN, k0 = tk * BK;
    const int rows_per = ((M + splits - 1) / splits + BKM - 1) / BKM * BKM;
    const int m_lo = split * rows_per, m_hi = min(M, m_lo + rows_per);
    const int nslab = m_hi > m_lo ? (m_hi - m_lo + BKM - 1) / BKM : 0;

    // staging: thread -> (row r, float4 column c); 64 float4 per 256-wide row, 8 rows per pass
    constexpr int C4 = BN / 4;
    f32x4 pa[V4], pb[V4];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int mb = m_lo + s * BKM;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / C4, c = idx - r * C4;
            const int m = mb + r;
            pa[i] = m < m_hi ? *(const f32x4*)(dY + (long)m * ldy + n0 + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / C4, c = idx - r * C4;
            const int m = mb + r;
            pb[i] = m < m_hi ? xl.load(xl.row(m), k0 + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto put = [&](const f32x4& v, __bf16* hi_img, __bf16* lo_img, int off) __attribute__((always_inline)) {
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (__bf16)v[e];
            l[e] = (__bf16)(v[e] - (float)h[e]);
        }
        *(bf16x4*)(hi_img + off) = h;
        if (NP == 3) *(bf16x4*)(lo_img + off) = l;
    };
    // [r3] csum != nullptr: the workgroups of the first k-tile column also sum the fp32 dY values they stage (before the rounding): csum[split][N]
    // = the layer's bias gradient over the split's rows, fixed order (a thread always stages the same four columns: 512 % 64 == 0)
    const bool do_cs = csum != nullptr && tk == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int idx = tid + i * THREADS, r = idx / C4, c = idx - r * C4;
            if (do_cs) cs += pa[i];
            put(pa[i], img(buf, 0, 0), img(buf, 0, NIMG - 1), r * STR + c * 4);
            put(pb[i], img(buf, 1, 0), img(buf, 1, NIMG - 1), r * STR + c * 4);
        }
    };

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read address of this lane inside a 32-column block: row q of the 4-row block, columns 16g + 4p
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int t_off = (8 * lh + tq) * STR + 16 * tg + 4 * tp;          // bf16 elements; + 4*STR for the second read
    auto frag = [&](const __bf16* image, int step, int col0) __attribute__((always_inline)) {
        const __bf16* p = image + step * 16 * STR + col0 + t_off;
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
        const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * STR));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = b[e];
        }
        return f;
    };

    if (nslab > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload(s + 1);
#pragma unroll
        for (int step = 0; step < BKM / 16; ++step) {
            bf16x8 fa[NIMG][TN];
#pragma unroll
            for (int part = 0; part < NIMG; ++part)
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[part][i] = frag(img(buf, 0, part), step, wn * (TN * 32) + i * 32);
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                bf16x8 fb[NIMG];
#pragma unroll
                for (int part = 0; part < NIMG; ++part) fb[part] = frag(img(buf, 1, part), step, wk * (TK * 32) + j * 32);
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0], acc[i][j], 0, 0, 0);
                    if (NP == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[NIMG - 1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[NIMG - 1][i], fb[0], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
    float* out = slabs + (long)split * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int kk = k0 + wk * (TK * 32) + j * 32 + l31;
            const int nb = n0 + wn * (TN * 32) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + (r & 3) + 8 * (r >> 2);
                if (n < N && kk < K) out[(long)n * K + kk] = acc[i][j][r];
            }
        }
    if (do_cs) {                                   // (uniform over the workgroup; the loop above ended with a barrier: the images are free)
        f32x4* red = (f32x4*)simg;
        red[tid] = cs;                             // [wave][lane = float4 column]
        __syncthreads();
        if (wid == 0) {
            f32x4 t = red[lane];
#pragma unroll
            for (int w = 1; w < THREADS / 64; ++w) t += red[w * 64 + lane];
            *(f32x4*)(csum + (long)split * N + n0 + lane * 4) = t;
        }
    }
}

// db != nullptr: also db[N] (+)= column sums of dY (the bias gradient of the same Linear layer)
template <class Cfg, class XLoad>
static hipError_t gemm_tn_bf16_launch(const float* dY, long ldy, const XLoad& xl, float* dW, float* slabs, size_t slab_bytes,
                                      int M, int N, int K, int num_cu, int accumulate, hipStream_t stream, float* db = nullptr) {
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0) return hipErrorInvalidValue;
    const int tiles_n = N / Cfg::BN, tiles_k = K / Cfg::BK;
    const int tiles = tiles_n * tiles_k;
    int splits = (num_cu + tiles - 1) / tiles;                   // one 512-thread block per CU (144 KB of LDS each)
    const int max_by_rows = (M + 4 * Cfg::BKM - 1) / (4 * Cfg::BKM);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    const size_t per_split = ((size_t)N * K + (db ? (size_t)N : 0)) * 4;      // with db: the split's column-sum row behind the slabs
    while ((size_t)splits * per_split > slab_bytes && splits > 1) --splits;
    const bool direct = splits == 1 && !accumulate;              // a single slab that is not added to anything IS the gradient
    if ((direct ? (db ? (size_t)N * 4 : 0) : (size_t)splits * per_split) > slab_bytes) return hipErrorOutOfMemory;
    float* csum = db ? slabs + (direct ? 0 : (size_t)splits * N * K) : nullptr;
    auto kern = gemm_tn_bf16_kernel<Cfg, XLoad>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, dY, ldy, xl, direct ? dW : slabs, M, N, K,
                       tiles_n, tiles_k, splits, csum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (db) hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)csum, db, (long)N, splits, accumulate);
    if (direct) return hipGetLastError();
    const long n = (long)N * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, dW, n, splits,
                       accumulate);
    return hipGetLastError();
}
