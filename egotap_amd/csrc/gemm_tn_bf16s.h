// bf16-STORAGE weight-gradient GEMM ("TN"):  dW[N,K] (+)= sum_m dY[m,n] * X[m,k],  dY, X bf16 in HBM, fp32 accumulate, dW fp32
// (nn.Linear's weight gradient of the reduced-precision training step; model/egotap_autoencoder_model.py:299-323).
//
// Same skeleton as gemm_bf16s.h -- 256 x 256 output tile, 8 waves = 2 groups x 4 one barrier apart, ring of four 32 KB LDS
// stages filled by global_load_lds_dwordx4, v_mfma_f32_16x16x32_bf16, two phases of 16 MFMAs per 32-deep step -- with the
// CONTRACTED index m being the memory row of both operands:
//   * a stage = [32 rows of dY | 32 rows of X] x 256 columns (512-byte rows), staged exactly as the rows lie in memory;
//   * MFMA A operand = X^T (row = k of dW), B operand = dY (column = n): a lane's 4 accumulator registers are 4 consecutive k of
//     one n, i.e. 16 contiguous bytes of dW[n][k..k+3] -- the partial tile is stored without a transpose;
//   * both fragments need 8 consecutive m for one column per lane: two ds_read_b64_tr_b16 each (4 x 16 block transpose,
//     cdna_hip_programming.md T10).  A 32-lane half of that instruction reads 8 row segments of 32 bytes: rows 8g + q (g = 16-lane
//     group, q = 0..3) of one 16-column block.  Chunk pair cb (32 bytes) of row r is stored at pair position
//     cb ^ f(r), f(r) = (r & 3) | (((r >> 3) & 1) << 2): the 8 segments then fall in 8 distinct 32-byte bank groups.  The swizzle is
//     applied on the global side of the DMA (which 16 bytes a lane fetches).
//   * M is split over workgroups (grid = tiles x splits); each split accumulates a whole number of 32-row steps and writes a
//     partial fp32 slab, reduce_slabs_kernel adds the slabs in a fixed order (bitwise reproducible, no float atomics).  Rows past M
//     are fetched from a caller-supplied page of zeros.
#pragma once
#include <algorithm>
#include <type_traits>

#include "gemm_bf16s.h"
#include "gemm_tn_bf16.h"
#include "lds_dma.h"

struct TnSCfg {
    static constexpr int BN = 256, BK = 256, BKM = 32, NS = 4, THREADS = 512;
    static constexpr int ROWB = 512;                      // bytes per LDS row (256 bf16)
    static constexpr int PART = BKM * ROWB;               // dY part or X part of a stage: 16 KiB
    static constexpr int STAGE = 2 * PART;
    static constexpr int LDS_BYTES = NS * STAGE;
};

// X-operand loaders for the TN kernel: address of 8 consecutive columns (16 bytes) k .. k+7 of row m = base + rowpart(m) + kpart(k) (elements).
// [r5] The column of a lane's DMA piece is fixed for the whole kernel (k0 of the workgroup's tile + the lane's chunk), so kpart is computed ONCE
// per lane; only rowpart runs per DMA instruction.  Through round 4 the gather loaders recomputed ptr(m, k) -- five integer divisions by run-time
// divisors, ~100 VALU instructions -- in front of every one of the kernel's DMA issues: the fc1 weight gradients ran at MFMA busy 0.35-0.37.
struct TXPlain {
    const __bf16* A;
    long lda;
    __device__ __forceinline__ const __bf16* base() const { return A; }
    __device__ __forceinline__ long kpart(int k) const { return k; }
    __device__ __forceinline__ long rowpart(int m) const { return (long)m * lda; }
    __device__ __forceinline__ const __bf16* ptr(int m, int k) const { return A + (long)m * lda + k; }
    long span(int M) const { return (long)M * lda; }      // (host) elements the M rows span from base()
};
struct TXTokens {            // rows gathered as XTokens (fc1 of the position encoder)
    const __bf16* Y;
    int T, D, seq, side, ppd, grid;
    __device__ __forceinline__ const __bf16* base() const { return Y; }
    __device__ __forceinline__ long kpart(int k) const {
        const int s = k / D, c = k - s * D;
        const int prl = s / ppd, pcl = s - prl * ppd;
        return (long)(prl * side + pcl) * D + c;
    }
    __device__ __forceinline__ long rowpart(int m) const {
        const int b = m / T, i = m - b * T;
        const int gr = i / grid;
        return ((long)b * seq + (long)(ppd * gr) * side + ppd * (i - gr * grid)) * D;
    }
    __device__ __forceinline__ const __bf16* ptr(int m, int k) const { return Y + rowpart(m) + kpart(k); }
    long span(int M) const { return (long)((M + T - 1) / T) * seq * D; }
};
struct TXRot {               // rows gathered as XRot (fc1 of the rotation encoder)
    const __bf16* hm;
    int C, J, HW;
    __device__ __forceinline__ const __bf16* base() const { return hm; }
    __device__ __forceinline__ long kpart(int k) const {
        const int cs = k / HW;
        return (long)cs * J * HW + (k - cs * HW);
    }
    __device__ __forceinline__ long rowpart(int m) const {
        const int T = 2 * J;
        const int b = m / T, t = m - b * T;
        const int eye = t / J, j = t - eye * J;
        return (long)(b * C + 2 * J + eye * 2 * J + j) * HW;
    }
    __device__ __forceinline__ const __bf16* ptr(int m, int k) const { return hm + rowpart(m) + kpart(k); }
    long span(int M) const { return (long)((M + 2 * J - 1) / (2 * J)) * C * HW; }
};

// FAST (plain X operand, M a multiple of 32: every 32-row step of every split is whole): the DMA addresses are a wave-uniform 64-bit base
// (scalar unit) + one 32-bit lane offset per row block, no per-lane pointer arithmetic, no zero-page selects (round 4: the issue path of
// the general form costs 8 v_mul_lo, 4 v_mad_u64 and 4 exec-mask regions per phase; the NT kernel's counterpart change was worth 5-7 %).
template <class XL, bool FAST = false>
__global__ __launch_bounds__(TnSCfg::THREADS, 2) void gemm_tn_bf16s_kernel(const __bf16* __restrict__ dY, long ldy, XL xl, const __bf16* __restrict__ zeros,
                                                                            float* __restrict__ slabs, int M, int N, int K, int tiles_n, int splits,
                                                                            int rows_per, int* __restrict__ sync) {
    using Cfg = TnSCfg;
    constexpr int BN = Cfg::BN, BK = Cfg::BK, BKM = Cfg::BKM, ROWB = Cfg::ROWB, PART = Cfg::PART, STAGE = Cfg::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wc = wid & 3;
    const int l15 = lane & 15;

    // XCD-aware order: an XCD takes a contiguous range of (split, tile) pairs with the tile index fastest, so the tiles_n x tiles_k tiles
    // of one split -- which read the same 32-row steps of dY and X -- share them in one L2 instead of fetching them once per XCD
    // (measured before: 6.9 GB of L2-side reads per launch against 3.6 GB algorithmic)
    const int tiles_k = K / BK;
    const int tiles = tiles_n * tiles_k;
    const int bid = xcd_lin(blockIdx.x, gridDim.x);
    const int split = bid / tiles, tile = bid - split * tiles;
    // [r3] tile order inside a split: an XCD's 32 workgroups read a dY column slabs and b X column slabs of the split (a x b tiles);
    // with the n index fastest a 16 x 4 tile grid (MLP-up weight gradient) gave an XCD 16 + 2 slabs and made the second XCD fetch
    // the 16 dY slabs again (36 slab reads per split against 20 algorithmic: the 2.1 x of the round-2 profile); walking the grid in
    // 8 x 4 blocks makes it 2 x (8 + 4) = 24.  Blocks of PA x PB tiles, PA * PB = 32, both dividing the grid (else the plain order).
    int tn, tk;
    {
        int pa = 0, pb = 0;
        if (tiles > 32) {
            if (tiles_n % 8 == 0 && tiles_k % 4 == 0) { pa = 8; pb = 4; }
            else if (tiles_n % 4 == 0 && tiles_k % 8 == 0) { pa = 4; pb = 8; }
        }
        if (pa == 0) { tn = tile % tiles_n; tk = tile / tiles_n; }
        else {
            const int blk = tile >> 5, in = tile & 31;
            const int bn = tiles_n / pa;                    // blocks along n
            tn = (blk % bn) * pa + in % pa;
            tk = (blk / bn) * pb + in / pa;
        }
    }
    const int n0 = tn * BN, k0 = tk * BK;
    // [r3] The workgroups of one XCD that work on one split stream the same dY / X slabs and share them through that XCD's L2 only
    // while they stay within ~20 steps of each other (4 MB over 12 slab streams of 16 KB per step).  Nothing kept them there: measured
    // 1.2 x the algorithmic bytes at 37 k rows, 1.35 x at 147 k, 2.6 x at 590 k (tools/tn_prof.sh).  Every SYNC_STEPS steps wave 0 of each
    // of them checks in at a counter and waits (bounded: a missing partner costs 60 us, never a hang) until all have; the other waves
    // wait for it at the phase's barrier as they always do.  sync == nullptr: off.
    constexpr int SYNC_STEPS = 128;
    int sync_size = 0, sync_slot = 0;
    if (sync != nullptr && (gridDim.x & 7) == 0) {
        const int chunk_sz = gridDim.x >> 3, chunk = bid / chunk_sz;
        const int lo = max(chunk * chunk_sz, split * tiles), hi = min((chunk + 1) * chunk_sz, (split + 1) * tiles);
        sync_size = hi - lo;
        sync_slot = chunk * splits + split;
    }
    const int m_lo = split * rows_per, m_hi = min(M, m_lo + rows_per);
    const int total = m_hi > m_lo ? (m_hi - m_lo + BKM - 1) / BKM : 0;      // 32-row steps of this split
    float* out = slabs + (long)split * N * K;

    // ---- DMA duty per part: 2-row blocks wid and wid + 8 (a wave instruction = 1024 bytes = two 512-byte rows).  Lane -> row
    // lane >> 5 of the block, chunk position lane & 31, holding logical chunk pos ^ (2 f(row)).
    const int r0 = 2 * wid + (lane >> 5), r1 = r0 + 16;
    auto fsw = [](int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); };
    const int c0 = (lane & 31) ^ fsw(r0), c1 = (lane & 31) ^ fsw(r1);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_t;
    auto dma1 = [&](const __bf16* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    int ly = 0, lx = 0;            // next step of the dY / X issue streams; past the end they keep re-reading the zero page
    // FAST: lane offsets (bytes) from the step's wave-uniform base; past the end the streams re-read the split's last step (valid memory,
    // landing in a stage nobody reads)
    unsigned oy0 = 0, oy1 = 0, ox0 = 0, ox1 = 0;
    unsigned long long ybase = 0, xbase = 0;
    auto uniform64 = [](unsigned long long v) __attribute__((always_inline)) { return lds_dma_base(v); };      // lds_dma.h
    auto dma_s = [&](unsigned voff, unsigned long long sbase, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    constexpr bool PLAINX = std::is_same<XL, TXPlain>::value;
    if constexpr (FAST) {
        oy0 = (unsigned)(((long)r0 * ldy + c0 * 8) * 2); oy1 = (unsigned)(((long)r1 * ldy + c1 * 8) * 2);
        ybase = uniform64((unsigned long long)(size_t)dY + ((unsigned long long)m_lo * ldy + n0) * 2);
        if constexpr (PLAINX) {
            ox0 = (unsigned)(((long)r0 * xl.lda + c0 * 8) * 2); ox1 = (unsigned)(((long)r1 * xl.lda + c1 * 8) * 2);
            xbase = uniform64((unsigned long long)(size_t)xl.A + ((unsigned long long)m_lo * xl.lda + k0) * 2);
        } else {
            // [r5] gathered rows (fc1 of the two encoders): every row whole and in range, the tensor under 4 GB -> wave-uniform tensor base + one
            // 32-bit byte offset per lane = the lane's fixed column part + the row's gather offset; no 64-bit pointer selects, no zero page
            ox0 = (unsigned)(xl.kpart(k0 + c0 * 8) * 2); ox1 = (unsigned)(xl.kpart(k0 + c1 * 8) * 2);
            xbase = uniform64((unsigned long long)(size_t)xl.base());
        }
    }
    // the lane's two column pieces of the X tile, fixed for the kernel (general path)
    const __bf16* const xk0 = xl.base() + xl.kpart(k0 + c0 * 8);
    const __bf16* const xk1 = xl.base() + xl.kpart(k0 + c1 * 8);
    auto issue_y = [&](int st) __attribute__((always_inline)) {
        const unsigned sa = lds0 + st * STAGE + wid * 1024;
        if constexpr (FAST) {
            const unsigned long long b = ybase + (unsigned long long)min(ly, total - 1) * (BKM * 2) * ldy;
            dma_s(oy0, b, sa);
            dma_s(oy1, b, sa + 8 * 1024);
            ++ly;
            return;
        }
        const int mb = m_lo + ly * BKM;
        const bool in = ly < total;
        const int ma = mb + r0, mc = mb + r1;
        dma1(in && ma < m_hi ? dY + (long)ma * ldy + n0 + c0 * 8 : zeros + c0 * 8, sa);
        dma1(in && mc < m_hi ? dY + (long)mc * ldy + n0 + c1 * 8 : zeros + c1 * 8, sa + 8 * 1024);
        ++ly;
    };
    auto issue_x = [&](int st) __attribute__((always_inline)) {
        const unsigned sa = lds0 + st * STAGE + PART + wid * 1024;
        if constexpr (FAST && PLAINX) {
            const unsigned long long b = xbase + (unsigned long long)min(lx, total - 1) * (BKM * 2) * xl.lda;
            dma_s(ox0, b, sa);
            dma_s(ox1, b, sa + 8 * 1024);
            ++lx;
            return;
        }
        if constexpr (FAST && !PLAINX) {
            const int mb = m_lo + min(lx, total - 1) * BKM;
            dma_s(ox0 + (unsigned)(xl.rowpart(mb + r0) * 2), xbase, sa);
            dma_s(ox1 + (unsigned)(xl.rowpart(mb + r1) * 2), xbase, sa + 8 * 1024);
            ++lx;
            return;
        }
        const int mb = m_lo + lx * BKM;
        const bool in = lx < total;
        const int ma = mb + r0, mc = mb + r1;
        dma1(in && ma < m_hi ? xk0 + xl.rowpart(ma) : zeros + c0 * 8, sa);
        dma1(in && mc < m_hi ? xk1 + xl.rowpart(mc) : zeros + c1 * 8, sa + 8 * 1024);
        ++lx;
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read address of this lane for a 16-column block cb: rows 8 g4 + q' (first read) and + 4 (second read),
    // 8 bytes at chunk 2 cb + (p >> 1), half p & 1; the swizzle term f depends on (q', g4 & 1) only, identical for both reads
    const int g4 = lane >> 4, qp = l15 >> 2, pp = l15 & 3;
    const int fr = qp | ((g4 & 1) << 2);
    const int t_row = (8 * g4 + qp) * ROWB + (pp & 1) * 8;
    auto frag = [&](const char* part, int cb) __attribute__((always_inline)) {
        const char* p = part + t_row + (((2 * cb + (pp >> 1)) ^ (2 * fr)) << 4);
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
        const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * ROWB));
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = b[e];
        }
        return f;
    };

    if (total > 0) {
        issue_y(0); issue_x(0);
        issue_y(1); issue_x(1);
        issue_y(2); issue_x(2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) __builtin_amdgcn_s_barrier();

        bf16x8 yf[4], xf[4];
        int st = 0;
        for (int T = 0; T < total; ++T) {
            const char* sa = smem_t + st * STAGE;
            const int st3 = (st + 3) & 3;
            if (sync_size > 1 && T > 0 && (T & (SYNC_STEPS - 1)) == 0 && wid == 0) {        // wave-uniform
                int met = 1;
                if (lane == 0) {
                    __hip_atomic_fetch_add(sync + sync_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int target = sync_size * (T / SYNC_STEPS);
                    met = 0;
                    for (int it = 0; it < 256 && !met; ++it) {
                        met = __hip_atomic_load(sync + sync_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
                        if (!met) __builtin_amdgcn_s_sleep(8);
                    }
                }
                // a partner that is not resident (other kernels on the device: a collective overlapped with the backward, a shared GPU)
                // costs this workgroup ONE bounded wait: after a miss it stops checking in (the counter then runs behind for its
                // partners, which give up the same way)
                if (!__builtin_amdgcn_readfirstlane(met)) sync_size = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the counter traffic must not sit in the in-order vmcnt count of the DMA waits
            }
            // ---------------- phase 2T: k half 0 of this wave
            issue_y(st3);
#pragma unroll
            for (int j = 0; j < 4; ++j) yf[j] = frag(sa, wc * 4 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = frag(sa + PART, grp * 8 + i);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- phase 2T + 1: k half 1
            issue_x(st3);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = frag(sa + PART, grp * 8 + 4 + i);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            st = (st + 1) & 3;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // partial tile: acc[i][j] = D[k = 16 i' + 4 q + r][n = 16 j' + l15]  ->  out[n][k .. k+3]
    const int q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l15, kk = k0 + grp * 128 + i * 16 + 4 * q;
            *(f32x4*)(out + (long)n * K + kk) = acc[i][j];
        }
}

static inline long total_steps_hint(int M, int splits) { return (long)M / splits / TnSCfg::BKM; }
static __global__ void zero_ints_kernel(int* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
// dY bf16 [M, N] (row stride ldy), X through xl, zeros: >= 512 bytes of zeros (16-byte aligned), dW fp32 [N, K]
template <class XL>
static hipError_t gemm_tn_bf16s_launch(const __bf16* dY, long ldy, const XL& xl, const __bf16* zeros, float* dW, float* slabs, size_t slab_bytes,
                                       int M, int N, int K, int num_cu, int accumulate, hipStream_t stream) {
    using Cfg = TnSCfg;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || ldy % 8 != 0 || M <= 0) return hipErrorInvalidValue;
    const int tiles_n = N / Cfg::BN, tiles_k = K / Cfg::BK;
    const int tiles = tiles_n * tiles_k;
    // one 512-thread workgroup per CU (128 KB of LDS each): the launch runs ceil(tiles * splits / CUs) rounds of M / splits rows each --
    // take the split count with the least rounds x rows (the smallest on ties: fewer slabs to reduce).  [r4: ceil(CUs / tiles) gave a
    // 12 x 4 tile grid 6 splits = 288 workgroups, two rounds for 1.1 rounds of work; 5 splits = 240 workgroups run 40 % faster]
    const int max_by_rows = (M + 8 * Cfg::BKM - 1) / (8 * Cfg::BKM);
    int splits = 1;
    {
        const int s_hi = std::min(max_by_rows, std::max(1, 2 * ((num_cu + tiles - 1) / tiles)));
        double best = 1e30;
        for (int sp = 1; sp <= s_hi; ++sp) {
            const double cost = (double)(((long)tiles * sp + num_cu - 1) / num_cu) / sp;
            if (cost < best * (1.0 - 1e-9)) { best = cost; splits = sp; }
        }
    }
    while ((size_t)splits * N * K * 4 > slab_bytes && splits > 1) --splits;
    const bool direct = splits == 1 && !accumulate;              // a single slab that is not added to anything IS the gradient
    if (!direct && (size_t)splits * N * K * 4 > slab_bytes) return hipErrorOutOfMemory;
    const int rows_per = ((M + splits - 1) / splits + Cfg::BKM - 1) / Cfg::BKM * Cfg::BKM;
    // plain operand and whole 32-row steps everywhere: the scalar-base form of the DMA addresses
    bool fast = false;
#if !(defined(EGOTAP_ABL) && (EGOTAP_ABL & 16))      // A/B: the general addressing everywhere
    if constexpr (std::is_same<XL, TXPlain>::value) fast = M % Cfg::BKM == 0 && (long)32 * (ldy > xl.lda ? ldy : xl.lda) * 2 < (1L << 31) && xl.lda % 8 == 0;
    else fast = M % Cfg::BKM == 0 && (long)32 * ldy * 2 < (1L << 31) && xl.span(M) * 2 < (1L << 32);      // gathered rows: 32-bit byte offsets from the tensor base
#endif
    auto kern = gemm_tn_bf16s_kernel<XL>;
    if (fast) kern = gemm_tn_bf16s_kernel<XL, true>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[fast]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done[fast] = true;
    }
    // check-in counters of the L2-sharing groups (8 XCD chunks x splits) behind the slabs, cleared per launch
    int* sync = nullptr;
    const size_t sync_off = ((size_t)splits * N * K * 4 + 255) & ~(size_t)255;
    if (!direct && slabs != nullptr && sync_off + (size_t)8 * splits * 4 <= slab_bytes && (long)total_steps_hint(M, splits) > 256) {
        sync = (int*)((char*)slabs + sync_off);
        hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(256), 0, stream, sync, 8 * splits);
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, dY, ldy, xl, zeros, direct ? dW : slabs, M, N, K,
                       tiles_n, splits, rows_per, sync);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || direct) return e;
    const long n = (long)N * K;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, dW, n, splits, accumulate);
    return hipGetLastError();
}
