// [r3] The ResNet stem of the bf16 estimators as ONE kernel on the bf16 matrix cores:
//     conv 7x7 / 2 pad 3 (3 -> 64) + BatchNorm(eval) + ReLU + MaxPool2d(3, 2, 1)      (torchvision resnet18.conv1 / bn1 / relu / maxpool
// via net_architecture.py:69-70), fp32 NCHW RGB in, bf16 channels-last eye-interleaved out ([B * HP * HP, 2 x 64], HP = S0 / 4: the
// layout conv_bf16s.h's stages read).  It replaces stem_conv7_mfma_kernel<true> (fp32 MFMA at 1/16 of the bf16 rate; wrote the
// 128 x 128 x 64 map, 1.07 GB per 256 stereo frames) + maxpool3s2_nhwc_bf16s_kernel (read it back): the stem's output feeds nothing
// but the max-pool (AfterBackbone never uses layer0, net_architecture.py:146-171), so it never has to reach HBM.
//
//   * Implicit GEMM on v_mfma_f32_32x32x16_bf16 with A = weights (row = output channel), B = input patch (column = stem pixel).  K is
//     ordered (c, ky, kx) with kx padded from 7 to 8, so the 8 k of a lane's B fragment are 8 CONSECUTIVE input pixels of one
//     (channel, row) of the patch: four ds_read_b32 at 4-byte alignment, consecutive lanes on consecutive dwords (conflict-free).
//     21 (c, ky) rows pad to 22 = 11 MFMA steps of two rows (lane half h takes row 2 step + h); the padded taps carry zero weights.
//     All A fragments (64 channels x 176 k) live in registers for the whole kernel: 88 VGPRs, no LDS traffic for weights.
//   * A workgroup (4 waves, 74 KB of LDS: two per CU, one stages while the other multiplies) is persistent over runs (image, group of
//     R = 8 pooled rows) and walks a run in 64-column segments of the stem map; per segment the input patch (39 rows x 136 columns x 3
//     channels, zero halo, converted to bf16 on the way) sits in LDS.  A segment walks 17 stem rows in pairs (waves 0-1 / 2-3 take
//     the two rows, one 32-pixel tile each): 22 MFMAs per wave and row, BatchNorm + ReLU on the accumulators, bf16 rows into a four-slot
//     LDS ring; after every pair the whole workgroup pools one output row out of three ring rows (one thread = 8 channels of one pooled
//     pixel, nine 16-byte LDS reads, unsigned 16-bit max: post-ReLU bf16 values are ordered like their bit patterns) and stores it.
//   * Max-pool padding: post-ReLU values are >= 0 and every window holds a valid element, so a zero column / row stands in for -inf.
//     The last stem column of a segment is the left neighbour of the next one's first pooling window: carried in LDS.
#pragma once
#include "gemm_bf16s.h"

struct StemPoolCfg {
    static constexpr int R = 8, SR = 2 * R + 1, PR = 2 * SR + 5;       // pooled rows per item, stem rows, patch rows (39)
    static constexpr int XS = 64, PCOLS = 136, PITCH = 320;            // stem columns per segment, staged patch columns, bytes per patch row
    static constexpr int PATCH_BYTES = 3 * PR * PITCH;                  // 37 440
    static constexpr int RING_ROW = (XS + 1) * 128, RING_BYTES = 4 * RING_ROW;      // [slot][1 + 64 columns][64 channels] bf16
    static constexpr int CARRY_BYTES = 2 * SR * 128;
    static constexpr int OFF_RING = PATCH_BYTES, OFF_CARRY = OFF_RING + RING_BYTES, OFF_BN = OFF_CARRY + CARRY_BYTES;
    static constexpr int LDS_BYTES = OFF_BN + 2 * 128 * 4;            // scale | shift, 64 channels x 2 eyes (MODE 2: per-eye batch statistics)
    static constexpr int THREADS = 256;
    static constexpr int NPRE = (3 * PR * (PCOLS / 2) + THREADS - 1) / THREADS;      // column pairs per thread: 32
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

typedef unsigned short u16x8s __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

// [r5] MODE 0: BatchNorm folded from the running statistics (gamma, beta, mean, var: eval mode).
// MODE 1: STATISTICS ONLY -- the same staging and MFMAs, no BatchNorm / pool / store: every lane sums its 32 (channel, pixel-column) accumulators and
//         their squares over the stem rows its workgroup OWNS (ys >= 1: row ys = 0 of a run is the row group above's last row, computed again only as
//         pooling halo), and the workgroup writes one partial row part[block][eye * 64 + c][2] (`out`, as floats; the other eye's 64 columns zero).
//         The launcher makes the grid a multiple of 2 x (row groups per image), so every run of a workgroup belongs to the same eye.
// MODE 2: BatchNorm with per-eye scale / shift tables (gamma = scale[2][64], beta = shift[2][64], from bn_finish_bf16s_kernel over MODE 1's partials):
//         batch-statistics BatchNorm of the frozen estimators under train.py:91 (bn_bf16s.h).
template <int MODE>
static __global__ __launch_bounds__(StemPoolCfg::THREADS, 2) void stem_pool_bf16s_kernel(
    const float* __restrict__ left, const float* __restrict__ right, const float* __restrict__ w, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, __bf16* __restrict__ out, int HIN, int nimg) {
    using Cfg = StemPoolCfg;
    constexpr int R = Cfg::R, SR = Cfg::SR, PR = Cfg::PR, XS = Cfg::XS, PCOLS = Cfg::PCOLS, PITCH = Cfg::PITCH, THREADS = Cfg::THREADS, NPRE = Cfg::NPRE;
    extern __shared__ __attribute__((aligned(16))) char sp_sm[];
    char* patch = sp_sm;
    char* ring = sp_sm + Cfg::OFF_RING;
    char* carry = sp_sm + Cfg::OFF_CARRY;
    float* bn_sc = (float*)(sp_sm + Cfg::OFF_BN);
    float* bn_sh = bn_sc + 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, xl = lane & 31, h = lane >> 5;
    const int HO = HIN / 2, HP = HIN / 4, xsegs = HO / XS, groups = HP / R;
    // a workgroup takes whole (image, row group) runs -- run blockIdx.x + k gridDim.x -- and walks a run's segments left to right, so a
    // carried column always comes from the item it processed just before; q = position in that sequence
    const long runs = (long)nimg * groups;
    const long my_runs = (long)blockIdx.x < runs ? (runs - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nq = my_runs * xsegs;
    auto decode = [&](long q, int& seg, int& g, int& n) __attribute__((always_inline)) {
        const long run = blockIdx.x + (q / xsegs) * (long)gridDim.x;
        seg = (int)(q % xsegs);
        g = (int)(run % groups);
        n = (int)(run / groups);
    };

    if (MODE == 0 && tid < 64) {
        const float sc = gamma[tid] / sqrtf(var[tid] + 1e-5f);
        bn_sc[tid] = sc;
        bn_sh[tid] = beta[tid] - mean[tid] * sc;
    }
    if (MODE == 2 && tid < 128) { bn_sc[tid] = gamma[tid]; bn_sh[tid] = beta[tid]; }
    f32x16 ssum[2], ssq[2];                                  // MODE 1 only (dead otherwise)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[0][r] = 0.f; ssum[1][r] = 0.f; ssq[0][r] = 0.f; ssq[1][r] = 0.f; }
    // ---- A fragments: channel mt * 32 + xl, k row rr = 2 step + h (c = rr / 7, ky = rr % 7), element j = kx (7 -> zero)
    bf16x8 af[2][11];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int st = 0; st < 11; ++st) {
            const int rr = 2 * st + h;
            const float* wp = w + (mt * 32 + xl) * 147 + rr * 7;       // rr * 7 = c * 49 + ky * 7
#pragma unroll
            for (int j = 0; j < 8; ++j) af[mt][st][j] = (__bf16)((j < 7 && rr < 21) ? wp[j] : 0.f);
        }
    // byte offset of this lane's (c, ky) row per step inside the patch (the padded row 21 reads row 20: its weights are zero)
    int roff[11];
#pragma unroll
    for (int st = 0; st < 11; ++st) {
        const int rr = min(2 * st + h, 20);
        roff[st] = ((rr / 7) * PR + rr % 7) * PITCH;
    }

    // ---- patch staging: pair i = tid + j * THREADS -> (channel, patch row, column pair); patch row pr <-> input row 4 py0 - 5 + pr,
    // patch column pc <-> input column 2 c0 - 3 + pc (c0 = first stem column of the segment)
    auto stage = [&](int seg, int g, int n) __attribute__((always_inline)) {
        const float* src = ((n & 1) ? right : left) + (long)(n >> 1) * 3 * HIN * HIN;
        const int iy0 = 4 * g * R - 5, ix0 = 2 * seg * XS - 3;
        // chunks of 8 column pairs per thread: 16 loads in flight, then convert and write (the index arithmetic is redone per chunk on
        // purpose: hoisted out of the segment loop it would hold 100+ registers next to the 88 of the weights)
#pragma unroll 1
        for (int j0 = 0; j0 < NPRE; j0 += 8) {
            float pre[8][2];
            int dst[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = tid + (j0 + j) * THREADS;
                const int c = i / (PR * (PCOLS / 2)), rem = i - c * (PR * (PCOLS / 2)), pr = rem / (PCOLS / 2), pp = rem - pr * (PCOLS / 2);
                const int y = iy0 + pr, x = ix0 + 2 * pp;
                const bool rowok = c < 3 && y >= 0 && y < HIN;
                const float* rp = src + ((long)c * HIN + y) * HIN;
                pre[j][0] = (rowok && x >= 0 && x < HIN) ? rp[x] : 0.f;
                pre[j][1] = (rowok && x + 1 >= 0 && x + 1 < HIN) ? rp[x + 1] : 0.f;
                dst[j] = c < 3 ? (c * PR + pr) * PITCH + pp * 4 : -1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
                bf16x2s v;
                v[0] = (__bf16)pre[j][0];
                v[1] = (__bf16)pre[j][1];
                if (dst[j] >= 0) *(bf16x2s*)(patch + dst[j]) = v;
            }
        }
    };

    const int t = wid & 1, par = wid >> 1;                   // 32-pixel tile of the segment, row of the pair
    const int pcol = 1 + 32 * t + xl;                        // ring column of this lane's pixel (column 0 = the pixel left of the segment)
    const int key = (pcol & 7) << 1;                         // 8-byte chunk swizzle of a ring pixel: chunk ch sits at ch ^ key (pairs stay in order)
    // pooling duty: pooled column ppx, channels 8 u .. 8 u + 7
    const int ppx = tid >> 3, u = tid & 7;

    int cbuf = 0;                                            // carry buffer the current segment READS (its left neighbour's last column)
    for (long q = 0; q < nq; ++q) {
        int seg, g, n;
        decode(q, seg, g, n);
        __syncthreads();                                     // the previous segment's MFMAs and pooling are done with patch and ring
        stage(seg, g, n);
        __syncthreads();
        const int py0 = g * R;
        for (int pair = 0; pair <= R; ++pair) {
            const int ys = 2 * pair - 1 + par;               // stem row 2 py0 - 1 + ys; pair 0: only ys = 0 (waves 2-3)
            const int ystem = 2 * py0 - 1 + ys;
            f32x16 acc[2];
            const bool rowact = ys >= 0;
            if (rowact) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
                const char* bp = patch + (2 * ys) * PITCH + (32 * t + xl) * 4;
#pragma unroll
                for (int st = 0; st < 11; ++st) {
                    const unsigned* q = (const unsigned*)(bp + roff[st]);
                    u32x4s raw = {q[0], q[1], q[2], q[3]};
                    const bf16x8 bfrag = __builtin_bit_cast(bf16x8, raw);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][st], bfrag, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][st], bfrag, acc[1], 0, 0, 0);
                }
            }
            if constexpr (MODE == 1) {
                if (ys >= 1) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { ssum[mt][r] += acc[mt][r]; ssq[mt][r] += acc[mt][r] * acc[mt][r]; }
                }
                continue;
            }
            const int eo = MODE == 2 ? (n & 1) * 64 : 0;      // this run's eye selects the scale / shift table
            __syncthreads();                                 // A: the previous pair's pooling has read its three ring rows
            if (rowact) {
                char* rrow = ring + (ys & 3) * Cfg::RING_ROW;
                const bool zero_row = ystem < 0;             // the row above the image: max-pool padding
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int co0 = mt * 32 + 8 * gq + 4 * h;
                        const f32x4 sc = *(const f32x4*)(bn_sc + eo + co0), sh = *(const f32x4*)(bn_sh + eo + co0);
                        bf16x4s o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = fmaxf(acc[mt][4 * gq + e] * sc[e] + sh[e], 0.f);
                            o[e] = (__bf16)(zero_row ? 0.f : v);
                        }
                        const int ch = co0 >> 2;
                        *(bf16x4s*)(rrow + pcol * 128 + ((ch ^ key) << 3)) = o;
                        if (t == 1 && xl == 31) *(bf16x4s*)(carry + ((cbuf ^ 1) * SR + ys) * 128 + (ch << 3)) = o;       // last column: the next segment's left neighbour
                    }
                if (t == 0 && lane < 16) {                   // ring column 0: zero (image edge) or the carried column of the previous segment
                    bf16x4s c0v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    if (seg > 0) c0v = *(const bf16x4s*)(carry + (cbuf * SR + ys) * 128 + (lane << 3));
                    *(bf16x4s*)(rrow + (lane << 3)) = c0v;  // key(0) = 0
                }
            }
            __syncthreads();                                 // B: both rows of the pair are in the ring
            if (pair >= 1) {
                // pooled row py0 + pair - 1 from stem rows ys = 2 pair - 2, 2 pair - 1, 2 pair; pooled column ppx from ring columns 2 ppx .. 2 ppx + 2
                u16x8s m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const char* rrow = ring + ((2 * pair - 2 + dy) & 3) * Cfg::RING_ROW;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int pc = 2 * ppx + dx;
                        const u16x8s v = *(const u16x8s*)(rrow + pc * 128 + (((2 * u) ^ ((pc & 7) << 1)) << 3));
                        m = __builtin_elementwise_max(m, v);
                    }
                }
                const long prow = ((long)(n >> 1) * HP + py0 + pair - 1) * HP + seg * (XS / 2) + ppx;
                *(u16x8s*)((char*)out + (prow * 128 + (n & 1) * 64 + u * 8) * 2) = m;
            }
        }
        cbuf ^= 1;                                           // the column this segment saved is the next segment's neighbour
    }
    if constexpr (MODE == 1) {
        // lanes of equal h hold the same 32 channels for 32 pixel columns: fold the 32 columns (fixed order), then the four waves through LDS
        __syncthreads();
        float* red = (float*)sp_sm;                          // [wave][64 channels][2]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = ssum[mt][r], b = ssq[mt][r];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (xl == 0) {
                    const int co = mt * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    red[(wid * 64 + co) * 2] = a;
                    red[(wid * 64 + co) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < 128) {
            const int co = tid >> 1, qq = tid & 1;
            const float a = (nq > 0) ? red[(0 * 64 + co) * 2 + qq] + red[(1 * 64 + co) * 2 + qq] + red[(2 * 64 + co) * 2 + qq] + red[(3 * 64 + co) * 2 + qq] : 0.f;
            const int groups_ = HP / R;
            const int eye = (int)((blockIdx.x / groups_) & 1);
            float* part = (float*)out + (long)blockIdx.x * 256;
            part[(eye * 64 + co) * 2 + qq] = a;
            part[((eye ^ 1) * 64 + co) * 2 + qq] = 0.f;
        }
    }
}

template <int MODE>
static inline hipError_t stem_pool_bf16s_launch_mode(const float* left, const float* right, const float* w, const float* gamma, const float* beta,
                                                     const float* mean, const float* var, __bf16* out, int HIN, int nimg, int num_cu, hipStream_t s, int* grid_out = nullptr) {
    using Cfg = StemPoolCfg;
    const int HO = HIN / 2, HP = HIN / 4;
    if (HO % Cfg::XS != 0 || HP % Cfg::R != 0 || nimg <= 0) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_pool_bf16s_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int groups = HP / Cfg::R;
    const long runs = (long)nimg * groups;
    long grid = runs < 2L * num_cu ? runs : 2L * num_cu;
    if (MODE == 1) {           // every run of a workgroup in ONE eye: run = block + k * grid, image = run / groups -> the grid a multiple of 2 * groups
        if (nimg % 2 != 0) return hipErrorInvalidValue;
        grid = grid / (2 * groups) * (2 * groups);
        if (grid <= 0) return hipErrorInvalidValue;
    }
    if (grid_out) *grid_out = (int)grid;
    hipLaunchKernelGGL(stem_pool_bf16s_kernel<MODE>, dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, left, right, w, gamma, beta, mean, var, out, HIN, nimg);
    return hipGetLastError();
}
static inline hipError_t stem_pool_bf16s_launch(const float* left, const float* right, const float* w, const float* gamma, const float* beta,
                                                const float* mean, const float* var, __bf16* out, int HIN, int nimg, int num_cu, hipStream_t s) {
    return stem_pool_bf16s_launch_mode<0>(left, right, w, gamma, beta, mean, var, out, HIN, nimg, num_cu, s);
}
