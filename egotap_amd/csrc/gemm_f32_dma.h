// Exact-fp32 GEMM with global -> LDS DMA staging:  C[M,N] = epi( A[M,K] * W[N,K]^T ),  everything fp32.
//
// The staging scheme of gemm_bf16_dma.h applied to the exact-fp32 kernel: a 16-float slab row is 64 bytes = four 16-byte chunks,
// the same geometry as 32 bf16, so the unpadded xor-swizzled LDS image, the lane -> (row, chunk) duty of the DMA and the fragment
// addressing carry over unchanged; a fragment float4 feeds four v_mfma_f32_32x32x2_f32 (k-permutation of gemm_f32.h).  Against
// gemm_f32_persist_kernel this removes the staging registers and every ds_write, and parks FOUR slabs in LDS (three in flight).
// Any loader that is pure address math (HAS_PTR: plain rows, the per-heatmap token regroup, the cos/sin maps) can feed the DMA;
// k order per output element is that of every other fp32 tile: bit-identical results.
#pragma once
#include <type_traits>
#include "gemm_f32.h"
#include "lds_dma.h"

struct DmaF32Cfg {
    static constexpr int BM = 256, BN = 256, BK = 16, WM = 4, WN = 2, THREADS = 512, NS = 4, TM = 2, TN = 4;
    static constexpr int ROWB = 64;                               // bytes per LDS row
    static constexpr int STAGE = (BM + BN) * ROWB;                // 32 KiB
    static constexpr int ELD = 36, EPATCH = 16 * ELD * 4;         // per-wave epilogue patch: 16 rows x 36 floats
    static constexpr int LDS_BYTES = NS * STAGE + (THREADS / 64) * EPATCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class ALoad, class Epi>
__global__ __launch_bounds__(DmaF32Cfg::THREADS, 2) void gemm_f32_dma_kernel(ALoad al, SegMat W, Epi epi, float* C,
                                                                          long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    using Cfg = DmaF32Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS, TM = Cfg::TM, TN = Cfg::TN, ROWB = Cfg::ROWB, ELD = Cfg::ELD;
    extern __shared__ __attribute__((aligned(16))) char smem_dmaf[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;
    float* Es = (float*)(smem_dmaf + NS * Cfg::STAGE + wid * Cfg::EPATCH);

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

    // DMA duty of this wave per slab: 16-row chunks wid and wid + 8 of A and of W.  Lane -> (row lane >> 2 of the chunk, chunk
    // position lane & 3), which holds logical chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ (lane >> 4).
    const int drow = lane >> 2, dchunk = (lane & 3) ^ (lane >> 4);
    typename ALoad::Row ra0, ra1;
    const float *pb0, *pb1;
    // [r5] plain row-major operands: a wave-uniform 64-bit base per tile (scalar registers) + a 32-bit byte offset per lane -- the global_load_lds
    // s[base] form -- instead of a 64-bit pointer per lane (-DEGOTAP_F32_DMA_FLAT keeps the pointer form for the A/B).  W is always plain rows.
#ifdef EGOTAP_F32_DMA_FLAT
    constexpr bool SBA = false, SBW = false;
#else
    constexpr bool SBA = std::is_same<ALoad, ALoadPlain>::value, SBW = true;
#endif
    unsigned long long abase = 0, wbase = 0;
    unsigned ao0 = 0, ao1 = 0;
    auto uniform64 = [](const void* p) __attribute__((always_inline)) { return lds_dma_base(p); };      // lds_dma.h
    const unsigned wo0 = (unsigned)(((long)(wid * 16 + drow) * W.ld + dchunk * 4) * 4), wo1 = (unsigned)(((long)((wid + 8) * 16 + drow) * W.ld + dchunk * 4) * 4);
    int l_tile = 0, l_kt = 0;
    auto set_rows = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        if constexpr (SBA) {
            abase = uniform64(al.A + (long)tm * BM * al.lda);
            const long ld = al.lda * 4;                                                                      // bytes per row
            ao0 = (unsigned)((long)(min(tm * BM + wid * 16 + drow, M - 1) - tm * BM) * ld + dchunk * 16);
            ao1 = (unsigned)((long)(min(tm * BM + (wid + 8) * 16 + drow, M - 1) - tm * BM) * ld + dchunk * 16);
        } else {
        ra0 = al.row(min(tm * BM + wid * 16 + drow, M - 1));
        ra1 = al.row(min(tm * BM + (wid + 8) * 16 + drow, M - 1));
        }
        // the segment of W is uniform over a tile (seg % 256 == 0): scalar selects, no indexed (vector) load of W.p[] whose
        // vmcnt wait would drain the DMA pipeline at every tile switch
        const int n0 = tn * BN, sidx = n0 / W.seg;
        const float* wp = (sidx == 0 ? W.p[0] : (sidx == 1 ? W.p[1] : W.p[2])) + (long)(n0 - sidx * W.seg) * W.ld;
        if constexpr (SBW) wbase = uniform64(wp);
        else {
        pb0 = wp + (long)(wid * 16 + drow) * W.ld + dchunk * 4;
        pb1 = wp + (long)((wid + 8) * 16 + drow) * W.ld + dchunk * 4;
        }
    };
    set_rows(0);
    // The DMA goes through inline asm: the compiler's waitcnt pass treats __builtin_amdgcn_global_load_lds as a store to LDS that
    // any later ds_read may alias and drains vmcnt(0) in front of every fragment read, which serialises the pipeline.  vmcnt for
    // these instructions is counted by hand (constant number in flight, see the loop).  M0 (the LDS base of the DMA) is a reserved
    // register: the compiler keeps no value in it across statements, so writing it here needs no clobber.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_dmaf;
    auto dma1 = [&](const float* g, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    // one slab of this block's slab stream -> stage st.  Unconditional (past the end it re-reads the last slab into a stage
    // nobody reads) so that the number of DMA instructions in flight is a constant the waits below can count on.
    auto dma1s = [&](unsigned voff, unsigned long long sbase, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    auto dma_part = [&](int st, int part) __attribute__((always_inline)) {       // one of the four DMA instructions of a slab
        const unsigned sa = lds0 + st * Cfg::STAGE + wid * 1024;
        const int k0 = l_kt * BK;
        if (part == 0) { if constexpr (SBA) dma1s(ao0, abase + (unsigned long long)k0 * 4, sa); else dma1(al.ptr(ra0, k0 + dchunk * 4), sa); }
        else if (part == 1) { if constexpr (SBA) dma1s(ao1, abase + (unsigned long long)k0 * 4, sa + 8 * 1024); else dma1(al.ptr(ra1, k0 + dchunk * 4), sa + 8 * 1024); }
        else if (part == 2) { if constexpr (SBW) dma1s(wo0, wbase + (unsigned long long)k0 * 4, sa + BM * ROWB); else dma1(pb0 + k0, sa + BM * ROWB); }
        else { if constexpr (SBW) dma1s(wo1, wbase + (unsigned long long)k0 * 4, sa + BM * ROWB + 8 * 1024); else dma1(pb1 + k0, sa + BM * ROWB + 8 * 1024); }
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        if (l_tile < my_n && ++l_kt == KT) {
            l_kt = 0;
            if (++l_tile < my_n) set_rows(l_tile);
            else l_kt = KT - 1;                 // stream exhausted: keep pointing at the last slab
        }
    };
    auto dma = [&](int st) __attribute__((always_inline)) {
        dma_part(st, 0); dma_part(st, 1); dma_part(st, 2); dma_part(st, 3);
        dma_advance();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment byte offsets inside a stage: row l31 of a 32-row block, logical chunk 2 t + lh (floats 8 t + 4 lh ..)
    const int sw = (l31 >> 2) & 3;
    const int f0 = l31 * ROWB + ((lh ^ sw) << 4), f1 = l31 * ROWB + (((2 + lh) ^ sw) << 4);
    const int a_base = (wm * (TM * 32)) * ROWB, b_base = (BM + wn * (TN * 32)) * ROWB;

    // Tried and dropped: issuing the epilogue of block q-1 behind the MFMAs of block q in a tile's last slab (to keep the matrix
    // pipe busy through the epilogue).  As a second, differently ordered copy of the MFMA phase it made the register allocator
    // shuffle the 128 accumulators between the copies and spill (32 TF); as one block-major sequence for every slab the four
    // back-to-back MFMAs on one accumulator cost more than the epilogue saves (144 -> 132 TF).  Also tried: storing the accumulators
    // as they stand (global_store_dword, two 128-byte row segments per instruction, no LDS patch): 135 TF against 140 at K = 1024,
    // and 86 with a residual read the same way.  Per tile the epilogue costs ~9 us (3.8 % at K = 1024, 1 % at K = 4096).
    int c_tile = 0, c_kt = 0;
    // FULL (every row of the tile exists: all tiles but a ragged last tile row): the stores are unconditional, so hipcc counts them exactly
    // and its waits for the residual rows of the NEXT half block (requested before this one's stores) leave this one's stores in flight;
    // with `if (m < M)` around them it had to assume they might not have been issued and waited as if they were the only thing in flight
    // behind the loads -- in order, i.e. for the previous half block's stores to complete: a store round trip per half block.
    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        int tm, tn;
        tile_of(c_tile, tm, tn);
        const int er = lane >> 3, ec = (lane & 7) * 4;
        const int nb0 = tn * BN + wn * (TN * 32) + ec, mb0 = tm * BM + wm * (TM * 32) + er;
        // 16 half blocks (32 columns x 16 rows) per wave; the residual rows of half block q+1 are requested before half block q
        // goes through the LDS patch, so their latency hides behind it
        f32x4 rs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, rn[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (Epi::HAS_RES) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) rs[s2] = epi.res4(min(mb0 + s2 * 8, M - 1), nb0);
        }
        // [r4] the per-column constants (bias ...) of half block q + 1 are requested while half block q is processed, like its residual
        // rows: loaded and waited for inside one half block they cost a memory round trip each -- and more: vmcnt counts stores too, in
        // order, so that wait also sat until the previous half block's stores had completed (sixteen serialised round trips per tile, the
        // ~9 us per tile measured in round 3).
        typename Epi::Col4 cc = epi.col4(nb0), cn = cc;
#pragma unroll
        for (int q = 0; q < TN * TM * 2; ++q) {
            const int j = q / (TM * 2), i = (q / 2) % TM, hf = q & 1;
            const int n0 = nb0 + j * 32, mb = mb0 + i * 32;
            if (q + 1 < TN * TM * 2 && (q + 1) / (TM * 2) != j) cn = epi.col4(nb0 + ((q + 1) / (TM * 2)) * 32);
            if (Epi::HAS_RES && q + 1 < TN * TM * 2) {
                const int jn = (q + 1) / (TM * 2), in = ((q + 1) / 2) % TM, hn = (q + 1) & 1;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) rn[s2] = epi.res4(min(mb0 + in * 32 + hn * 16 + s2 * 8, M - 1), nb0 + jn * 32);
            }
#pragma unroll
            for (int r = 8 * hf; r < 8 * hf + 8; ++r) Es[((r & 3) + 8 * ((r >> 2) & 1) + 4 * lh) * ELD + l31] = acc[i][j][r];
#pragma unroll
            for (int r = 8 * hf; r < 8 * hf + 8; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int m = mb + hf * 16 + s2 * 8;
                if (FULL || m < M) {
                    const f32x4 o = epi.apply4(*(const f32x4*)(Es + (s2 * 8 + er) * ELD + ec), cc, rs[s2], m, n0);
                    *(f32x4*)(C + (long)m * ldc + n0) = o;
                }
            }
            if (Epi::HAS_RES) { rs[0] = rn[0]; rs[1] = rn[1]; }
            cc = cn;
        }
    };

    // Tried in round 2 and dropped: the two-group schedule of gemm_bf16s.h (two groups of four waves one barrier apart, a phase =
    // [DMA of half a slab; fragment reads; vmcnt] barrier [32 MFMAs] barrier, so that one wave's DMA issue, fragment reads and barrier
    // skew sit under the other wave's MFMA burst).  Correct and bit-identical, but 130.6 TF against 139.0 on the same box (129.8 with
    // the fragments prefetched one phase ahead): with v_mfma_f32_32x32x2_f32 a SIMD's matrix pipe is fed better by two waves
    // interleaving their MFMAs than by one wave at a time -- a single wave leaves a few idle cycles between its back-to-back MFMAs,
    // which costs more than the ~8 % of exposed issue work the lockstep schedule pays per slab.
    // Pipeline.  The barrier sits in the MIDDLE of a slab: [fragments t=1 of slab gs] [32 MFMAs t=0] wait + barrier (slab gs+1
    // has landed for everyone, nobody reads stage gs-1 any more) [DMA slab gs+3 -> stage gs-1] [fragments t=0 of slab gs+1]
    // [32 MFMAs t=1].  Every fragment read is issued one MFMA group ahead of its use, so no LDS latency is exposed after the
    // barrier; a slab is in flight for two slab times (7 us at the fp32 MFMA rate).
    dma(0);
    dma(1);
    dma(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a0[i] = *(const f32x4*)(smem_dmaf + a_base + i * 32 * ROWB + f0);
#pragma unroll
    for (int j = 0; j < TN; ++j) b0[j] = *(const f32x4*)(smem_dmaf + b_base + j * 32 * ROWB + f0);
    int st = 0;
    bool after_epi = false;
    for (int gs = 0; gs < total; ++gs) {
        const char* sa = smem_dmaf + st * Cfg::STAGE;
        const int st1 = st + 1 == NS ? 0 : st + 1;
        // the six fragment reads of the NEXT burst are issued one by one behind the first MFMAs of this one (same reason as the DMA
        // issues below: in front of the burst they are issue cycles with the matrix pipe idle)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            acc[q & 1][q >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q & 1][0], b0[q >> 1][0], acc[q & 1][q >> 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < TM) a1[q] = *(const f32x4*)(sa + a_base + q * 32 * ROWB + f1);
            else b1[q - TM] = *(const f32x4*)(sa + b_base + (q - TM) * 32 * ROWB + f1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    if (u > 0 || j >= 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][u], b0[j][u], acc[i][j], 0, 0, 0);
        // slab gs+1 has landed once at most the 4 DMA instructions of slab gs+2 are outstanding (vmcnt is in order; stores of an
        // epilogue in between only make the wait longer)
        // ... which is why the wait is skipped right after an epilogue: it was done in front of the epilogue's stores (below), and
        // repeated here it would sit until the 32 stores behind slab gs+1 in the queue have drained to HBM
        if (!after_epi) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        after_epi = false;
        __syncthreads();
        // The four DMA instructions of slab gs+3 are issued BETWEEN the first MFMAs of this burst (their operands a1 / b1 were read
        // before the barrier): a DMA costs the issuing SIMD ~56 issue cycles, and issued in front of the burst -- by both waves of
        // the SIMD at once, they leave the barrier together -- those 224 cycles per slab were matrix-pipe idle time.
        const int sd = st == 0 ? NS - 1 : st - 1;
        const char* sn = smem_dmaf + st1 * Cfg::STAGE;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q & 1][q >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q & 1][0], b1[q >> 1][0], acc[q & 1][q >> 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma_part(sd, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        dma_advance();
#pragma unroll
        for (int q = 0; q < 6; ++q) {             // (i, j) = (0, 2), (1, 2), (0, 3), (1, 3) at u = 0, then (0, 0), (1, 0) at u = 1
            const int i = q & 1, j = q < 4 ? 2 + (q >> 1) : 0, u = q < 4 ? 0 : 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][u], b1[j][u], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < TM) a0[q] = *(const f32x4*)(sn + a_base + q * 32 * ROWB + f0);
            else b0[q - TM] = *(const f32x4*)(sn + b_base + (q - TM) * 32 * ROWB + f0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    if (u > 1 || (u == 1 && j >= 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][u], b1[j][u], acc[i][j], 0, 0, 0);
        st = st1;
        if (++c_kt == KT) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // slab gs+2 (only slab gs+3 may still be in flight)
            after_epi = true;
            {
                int tm_, tn_;
                tile_of(c_tile, tm_, tn_);
                // (epilogues with wide per-column constants -- patch embedding, BatchNorm -- keep the one predicated copy: two copies of
                // their epilogue pushed the kernel past 256 registers)
                // (... and the two training epilogues with a residual operand and no constants, EpiAccum / EpiGeluGrad, spilled with two)
                constexpr bool DUAL = sizeof(typename Epi::Col4) <= 16 && !(Epi::HAS_RES && sizeof(typename Epi::Col4) < 16);
                if (DUAL && (tm_ + 1) * BM <= M) epilogue(std::integral_constant<bool, DUAL>{});
                else epilogue(std::false_type{});
            }
            c_kt = 0;
            ++c_tile;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (dummy) DMAs must not outlive the workgroup's LDS
}

// al: a loader with HAS_PTR (16-byte aligned 4-float groups, al.dma_ok()), W: fp32 rows (ld multiple of 4)
template <class ALoad, class Epi>
static hipError_t gemm_f32_dma_launch(const ALoad& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N, int K,
                                       int num_cu, hipStream_t stream) {
    using Cfg = DmaF32Cfg;
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0 || !al.dma_ok() || W.ld % 4 != 0) return hipErrorInvalidValue;
    if (W.ld >= (1L << 21) || K >= (1 << 21)) return hipErrorInvalidValue;      // 256 rows of an operand within the 32-bit lane offsets of the scalar-base DMA
    auto kern = gemm_f32_dma_kernel<ALoad, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, al, W, epi, C, ldc, M, N, K, tiles_m, tiles_n);
    return hipGetLastError();
}
