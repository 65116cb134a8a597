// RETIRED from the shipped library in round 4 (kept for the record; not compiled): the 64-key-step forward of the DMA-staged bf16 attention
// (generation 2) and its launcher.  Same-box A/B against the 32-key, three-workgroups-per-CU forward that ships (attention_bf16s3_kernel):
// 2.10 -> 1.74 ms per layer at B = 1024, N = 576; 0.90 -> 1.04 PF at N = 2304; bit-identical context and log-sum-exp
// (tools/attn_fwd_gen_probe.py, round 3).  Needs the helpers of egotap_amd/csrc/attention_bf16s2.h (namespace att2).
// ------------------------------------------------------------------------------------------------- forward, generation 2
// S^T = K Q^T (K rows from the image, Q^T in registers), online softmax per lane (the query is the lane), O^T += V^T P^T (V transposed
// from its image, the probability accumulators as the B operand where they stand).  64 keys per step, staged as in the dQ kernel.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attention_bf16s2_kernel(const __bf16* __restrict__ QKV, __bf16* __restrict__ CTX, int N, int heads,
                                                                     int qgroups, float scale_log2e, float* __restrict__ LSE) {
    using namespace att2;
    static_assert(NW == 4, "four waves share the DMA duty of a 64-key step");
    constexpr unsigned KVBUF = 4 * TILEB;
    constexpr float RESC = 6.0f;                               // deferred running-maximum update (attention_bf16s.h)
    extern __shared__ __attribute__((aligned(16))) char sm2[];
    const int lin = xcd_lin(blockIdx.x, gridDim.x);
    const int bh = lin / qgroups, qg = lin - bh * qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    const int D = heads * DH;
    const int ld3 = 3 * D;
    const __bf16* qkv = QKV + (long)b * N * ld3 + h * DH;
    const int qb = qg * NW + wid;
    const bool valid = qb * 32 < N;
    const int q0 = min(qb * 32, N - 32);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm2;
    int ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = 4 * wid + i;
        ok[i] = 32 * (e >> 3) * ld3 + D + dma_src(e & 7, lane, ld3);
    }
    auto issue = [&](int kt, unsigned boff) __attribute__((always_inline)) {
        const __bf16* src = qkv + (long)(kt * 64) * ld3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a = lds0 + boff + 1024 * (4 * wid + i);
            dma16(src + ok[i], a);
            dma16(src + ok[i] + D, a + 2 * TILEB);
        }
    };
    issue(0, 0);
    Frags<1> qf;
    attns::load_row_frags(qf, qkv + (long)(q0 + l31) * ld3, lh);
    const LaneAddr la = lane_addr(lane);
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                      // m_run in base-2 units (score * scale_log2e)
    const int ntiles = N / 64;
    auto step = [&](int kt, unsigned boff) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ntiles) issue(kt + 1, boff ^ KVBUF);
        if (!valid) return;                                     // (wave-uniform) stage only
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 s = rows_x_frags(sm2 + boff + sub * TILEB, la, qf);     // S^T[key][q]
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
            const bool raise = mx > m_run + RESC;
            if (__builtin_amdgcn_ballot_w64(raise) != 0) {                   // rare after the first tile: wave-uniform branch
                const float m_new = raise ? mx : m_run;
                const float alpha = exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
                psum += s[r];
            }
            l_run += psum;
            imgT_x_p(o, sm2 + boff + 2 * TILEB + sub * TILEB, la, s);       // O^T[d][q] += V^T P^T
        }
    };
    for (int kt = 0; kt < ntiles; kt += 2) {
        step(kt, 0);
        if (kt + 1 < ntiles) step(kt + 1, KVBUF);
    }
    __syncthreads();
    if (valid) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (LSE != nullptr && lh == 0) LSE[(long)bh * N + q0 + l31] = m_run * 0.6931471805599453f + logf(l_tot);
        attns::store_rows_bf16(o, 1.0f / l_tot, (float*)sm2 + wid * 32 * OLD, CTX + ((long)b * N + q0) * D + h * DH, D, lane);
    }
}


static hipError_t attention_bf16s2_fwd_launch(const __bf16* QKV, __bf16* CTX, float* LSE, int B, int N, int heads, hipStream_t stream) {
    using namespace att2;
    constexpr int NW = 4;
    constexpr size_t img = 2 * 4 * (size_t)TILEB, patch = (size_t)NW * 32 * OLD * 4;
    constexpr size_t lds = img > patch ? img : patch;
    auto kern = attention_bf16s2_kernel<NW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int groups = (N / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3(B * heads * groups), dim3(64 * NW), lds, stream, QKV, CTX, N, heads, groups, 1.4426950408889634f / sqrtf(128.0f), LSE);
    return hipGetLastError();
}
