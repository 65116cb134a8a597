// Plain-bf16 GEMM with bf16 operands in HBM and LDS-DMA staging:  C[M,N] = epi( A[M,K] * W[N,K]^T ),  A, W bf16, C fp32.
//
// Why a second bf16 kernel.  gemm_bf16_persist_kernel<NP = 1> keeps fp32 tensors in HBM and converts in registers; at the
// bf16 MFMA rate a 256x256x32 slab lasts 0.43 us, so covering the L2-miss latency (~2 us loaded) needs ~200 KB of loads in
// flight per CU and the register file holds 96 KB of staging at most: measured MFMA busy 0.11-0.20 (0.5 PFLOP/s), with
// spills on top in the epilogues that need more registers.  Here the operands are bf16 already (a scratch copy of the
// activations and of W, written by f32_to_bf16_kernel or by a producer), so a slab is half the bytes, and it goes
// global -> LDS directly (global_load_lds_dwordx4): no staging registers, no conversion VALU, no ds_write, and the data
// in flight is parked in LDS -- FOUR 32 KB stages, three slabs in flight.
//
// LDS image of a stage: [A rows 0..255][W rows 0..255], one row = 32 bf16 of k = 64 bytes = four 16-byte chunks, unpadded
// (the DMA writes lane i of a wave at base + 16 i, so a wave instruction fills 16 whole rows).  Chunk c of row r is stored at
// chunk position c ^ ((r >> 2) & 3): the ds_read_b128 of a fragment (16 consecutive rows, same logical chunk) then touches
// 16 distinct 16-byte bank groups.  The swizzle is applied on the global side (which 16 bytes a lane fetches).
// Wave tiling, persistent XCD-aware tile walk and the epilogue functors are those of gemm_bf16_persist_kernel; the epilogue's
// per-wave transpose patch is 16 rows (two passes per 32x32 block) to leave LDS for the fourth stage.
#pragma once
#include "gemm_bf16.h"
#include "lds_dma.h"

struct DmaCfg {
    static constexpr int BM = 256, BN = 256, BK = 32, WM = 4, WN = 2, THREADS = 512, NS = 4, TM = 2, TN = 4;
    static constexpr int ROWB = 64;                               // bytes per LDS row
    static constexpr int STAGE = (BM + BN) * ROWB;                // 32 KiB
    static constexpr int ELD = 36, EPATCH = 16 * ELD * 4;         // per-wave epilogue patch: 16 rows x 36 floats
    static constexpr int LDS_BYTES = NS * STAGE + (THREADS / 64) * EPATCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class Epi>
__global__ __launch_bounds__(DmaCfg::THREADS, 2) void gemm_bf16_dma_kernel(const __bf16* __restrict__ A, long lda, SegMatB W, Epi epi, float* C,
                                                                          long ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    using Cfg = DmaCfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS, TM = Cfg::TM, TN = Cfg::TN, ROWB = Cfg::ROWB, ELD = Cfg::ELD;
    extern __shared__ __attribute__((aligned(16))) char smem_dma[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int l31 = lane & 31, lh = lane >> 5;
    float* Es = (float*)(smem_dma + NS * Cfg::STAGE + wid * Cfg::EPATCH);

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
    // [r5] wave-uniform 64-bit base per tile (scalar registers) + 32-bit byte offset per lane: the global_load_lds s[base] form
    unsigned long long abase = 0, wbase = 0;
    unsigned ao0 = 0, ao1 = 0;
    auto uniform64 = [](const void* p) __attribute__((always_inline)) { return lds_dma_base(p); };      // lds_dma.h
    const unsigned wo0 = (unsigned)(((long)(wid * 16 + drow) * W.ld + dchunk * 8) * 2), wo1 = (unsigned)(((long)((wid + 8) * 16 + drow) * W.ld + dchunk * 8) * 2);
    int l_tile = 0, l_kt = 0;
    auto set_rows = [&](int i) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(i, tm, tn);
        abase = uniform64(A + (long)tm * BM * lda);
        ao0 = (unsigned)(((long)(min(tm * BM + wid * 16 + drow, M - 1) - tm * BM) * lda + dchunk * 8) * 2);
        ao1 = (unsigned)(((long)(min(tm * BM + (wid + 8) * 16 + drow, M - 1) - tm * BM) * lda + dchunk * 8) * 2);
        // the segment of W is uniform over a tile (seg % 256 == 0): scalar selects, no indexed (vector) load of W.p[] whose
        // vmcnt wait would drain the DMA pipeline at every tile switch
        const int n0 = tn * BN, sidx = n0 / W.seg;
        const __bf16* wp = (sidx == 0 ? W.p[0] : (sidx == 1 ? W.p[1] : W.p[2])) + (long)(n0 - sidx * W.seg) * W.ld;
        wbase = uniform64(wp);
    };
    set_rows(0);
    // The DMA goes through inline asm: the compiler's waitcnt pass treats __builtin_amdgcn_global_load_lds as a store to LDS that
    // any later ds_read may alias and drains vmcnt(0) in front of every fragment read, which serialises the pipeline.  vmcnt for
    // these instructions is counted by hand (constant number in flight, see the loop).  M0 (the LDS base of the DMA) is a reserved
    // register: the compiler keeps no value in it across statements, so writing it here needs no clobber.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_dma;
    auto dma1 = [&](unsigned voff, unsigned long long sbase, unsigned lds_addr) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
    };
    // one slab of this block's slab stream -> stage st.  Unconditional (past the end it re-reads the last slab into a stage
    // nobody reads) so that the number of DMA instructions in flight is a constant the waits below can count on.
    auto dma = [&](int st) __attribute__((always_inline)) {
        const unsigned sa = lds0 + st * Cfg::STAGE + wid * 1024;
        const int k0 = l_kt * BK;
        const unsigned long long ka = abase + (unsigned long long)k0 * 2, kw = wbase + (unsigned long long)k0 * 2;
        dma1(ao0, ka, sa);
        dma1(ao1, ka, sa + 8 * 1024);
        dma1(wo0, kw, sa + BM * ROWB);
        dma1(wo1, kw, sa + BM * ROWB + 8 * 1024);
        if (l_tile < my_n && ++l_kt == KT) {
            l_kt = 0;
            if (++l_tile < my_n) set_rows(l_tile);
            else l_kt = KT - 1;                 // stream exhausted: keep pointing at the last slab
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment byte offsets inside a stage: row l31 of a 32-row block, logical chunk 2 ks + lh
    const int sw = (l31 >> 2) & 3;
    const int f0 = l31 * ROWB + ((lh ^ sw) << 4), f1 = l31 * ROWB + (((2 + lh) ^ sw) << 4);
    const int a_base = (wm * (TM * 32)) * ROWB, b_base = (BM + wn * (TN * 32)) * ROWB;

    int c_tile = 0, c_kt = 0;
    auto epilogue = [&]() __attribute__((always_inline)) {
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
#pragma unroll
        for (int q = 0; q < TN * TM * 2; ++q) {
            const int j = q / (TM * 2), i = (q / 2) % TM, hf = q & 1;
            const int n0 = nb0 + j * 32, mb = mb0 + i * 32;
            const typename Epi::Col4 cc = epi.col4(n0);
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
                if (m < M) {
                    const f32x4 o = epi.apply4(*(const f32x4*)(Es + (s2 * 8 + er) * ELD + ec), cc, rs[s2], m, n0);
                    *(f32x4*)(C + (long)m * ldc + n0) = o;
                }
            }
            if (Epi::HAS_RES) { rs[0] = rn[0]; rs[1] = rn[1]; }
        }
    };

    dma(0);
    dma(1);
    dma(2);
    int st = 0;
    for (int gs = 0; gs < total; ++gs) {
        // slab gs has landed once at most the 8 DMA instructions of slabs gs+1, gs+2 are outstanding (vmcnt is in order; stores of
        // an epilogue in between only make the wait longer); the barrier publishes it and retires every read of stage st-1
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        dma(st == 0 ? NS - 1 : st - 1);
        const char* sa = smem_dma + st * Cfg::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks == 0 ? f0 : f1;
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(sa + a_base + i * 32 * ROWB + fo);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(sa + b_base + j * 32 * ROWB + fo);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        st = st + 1 == NS ? 0 : st + 1;
        if (++c_kt == KT) {
            epilogue();
            c_kt = 0;
            ++c_tile;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (dummy) DMAs must not outlive the workgroup's LDS
}

// A: bf16 [M, K] with row stride lda (elements, multiple of 8), W: bf16 rows, C fp32
template <class Epi>
static hipError_t gemm_bf16_dma_launch(const __bf16* A, long lda, const SegMatB& W, const Epi& epi, float* C, long ldc, int M, int N, int K,
                                       int num_cu, hipStream_t stream) {
    using Cfg = DmaCfg;
    if (M <= 0) return hipSuccess;
    if (N % Cfg::BN != 0 || K % Cfg::BK != 0 || W.seg % Cfg::BN != 0 || lda % 8 != 0 || W.ld % 8 != 0 || lda >= (1L << 22) || W.ld >= (1L << 22)) return hipErrorInvalidValue;
    auto kern = gemm_bf16_dma_kernel<Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = N / Cfg::BN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, A, lda, W, epi, C, ldc, M, N, K, tiles_m, tiles_n);
    return hipGetLastError();
}
