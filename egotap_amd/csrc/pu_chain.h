// Propagation-Unit recurrence and the pose head (custom_cells.py:94-197,
// net_architecture.py:513-576, 732-751).
//
// The reference's PU writes its new state into the caller's tensors, so SkelNet's per-joint
// states alias one tensor and the "tree" is a chain over joint index: a 2-layer modified LSTM
// over J steps.  Everything that does not depend on the state (x2f, x2h,
// b2h of all J steps) is batched into ordinary GEMMs by the caller; what is left per step is
//     hp    = sigmoid(F_t[:, 0:H]) * h_{t-1}
//     gates = Gin_t + hp * Whh^T + bhh                       (gate order: forget, input, cell, output)
//     c_t   = c_{t-1} * sig(f) + sig(i) * tanh(g) ;  h_t = sig(o) * tanh(c_t)
// which pu_step_r16_kernel does in one launch per step (30 dependent steps per forward).
#pragma once
#include "common.h"
#include "layernorm.h"   // wave_sum

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#define PU_FAULT_WORDS 16      // fault words of one pu_chain_launch: one per row-block chunk (the workspace reserves 64 bytes)
static __global__ void fill_u32_kernel(unsigned* p, long n, unsigned v, unsigned* zero_words = nullptr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
    if (i < PU_FAULT_WORDS && zero_words) zero_words[i] = 0u;
}

// One recurrent step of a PU layer.  The gated state hp_t = sigmoid(F_t[:, 0:H]) * h_{t-1} is produced by the PREVIOUS step's pointwise
// stage (which owns h_{t-1} element by element and knows F_t, a state-independent GEMM output): one sigmoid per element and step
// instead of one per element, gate wave and unit block (64 x redundant when every GEMM wave gated its own operand slice: 7-8k VALU
// cycles per wave in front of 2k cycles of MFMA -- that, not the matrix pipe, was the 28 us step).
// A block owns 16 batch rows (v_mfma_f32_16x16x4_f32) x 32 units x 4 gates, so B = 256 is 256 workgroups; wave = (gate, quarter of K),
// every operand of a wave's 64 MFMAs is requested up front in two rounds, partial sums meet in LDS, pointwise on 512 threads.
// k order inside a quarter: MFMA (i, u) contracts k = 16 i + 4 g + u over the four lane groups g; the quarters are added in the
// order 0..3.  One kernel for every batch size: a row's result does not depend on the batch it is in.
static __global__ __launch_bounds__(1024) void pu_step_r16_kernel(const float* __restrict__ hp_in, const float* __restrict__ Gin_t,
                                                                  const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                                  const float* c_prev, float* c_out, float* __restrict__ h_out,
                                                                  const float* __restrict__ F_next, int ldf_next, float* __restrict__ hp_out,
                                                                  float* __restrict__ gpre_out, int B, int H) {
    __shared__ f32x4 red[4 * 4 * 2 * 64];        // [quarter][gate][unit tile][lane] (32 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int g = wid & 3, kq4 = wid >> 2;
    // workgroups go round-robin over the 8 XCDs: XCD x gets the unit blocks {2x, 2x + 1} of every row block, so each L2 holds one
    // eighth of Whh (512 KiB, resident from step to step) instead of all 4 MiB
    const int lin = blockIdx.x;
    const int r0 = (lin >> 4) * 16, u0 = (((lin & 7) << 1) | ((lin >> 3) & 1)) * 32;
    const int arow = min(r0 + l15, B - 1);
    const int k0 = kq4 * 128;                    // H = 512 (checked by the launcher): a quarter of K is 128
    // operands of this thread's pointwise element (threads 0..511: row tid >> 5, unit tid & 31), requested up front
    const int prow = r0 + (tid >> 5), punit = u0 + (tid & 31);
    const int prc = min(prow, B - 1);
    const bool pw = tid < 512;
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f}, cpv = 0.f, fnext = 0.f;
    if (pw) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gin[q] = Gin_t[(long)prc * 4 * H + (long)q * H + punit];
            bg[q] = bhh[q * H + punit];
        }
        cpv = c_prev[(long)prc * H + punit];
        if (F_next) fnext = F_next[(long)prc * ldf_next + punit];
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* hp = hp_in + (long)arow * H + k0 + 4 * lg;
    const float* wp = Whh + (long)g * H * H + (long)(u0 + l15) * H + k0 + 4 * lg;
    // H = 512: 8 iterations of 16 k; every operand of the wave is in flight before its first MFMA
    f32x4 a[8], w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = *(const f32x4*)(hp + 16 * i);
        w0[i] = *(const f32x4*)(wp + 16 * i);
        w1[i] = *(const f32x4*)(wp + (long)16 * H + 16 * i);
    }
    __builtin_amdgcn_sched_barrier(0);           // keep the 24 requests ahead of the MFMAs (the scheduler would pair them up)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w0[i][u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w1[i][u], acc[1], 0, 0, 0);
        }
    }
    red[((kq4 * 4 + g) * 2 + 0) * 64 + lane] = acc[0];
    red[((kq4 * 4 + g) * 2 + 1) * 64 + lane] = acc[1];
    __syncthreads();
    if (pw && prow < B) {
        // element (row_local, unit_local) of the 16 x 16 accumulator tiles: tile unit_local >> 4, lane (unit_local & 15) + 16 * (row_local >> 2), register row_local & 3
        const int rl = tid >> 5, ul = tid & 31;
        const int idx = (ul >> 4) * 64 + (ul & 15) + 16 * (rl >> 2), reg = rl & 3;
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = red[(0 * 4 + q) * 128 + idx][reg];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[(w * 4 + q) * 128 + idx][reg];
            pre[q] = v + bg[q];
        }
        const float pf = pre[0] + gin[0], pi = pre[1] + gin[1], pc = pre[2] + gin[2], po = pre[3] + gin[3];
        if (gpre_out) {      // training: keep the gate pre-activations for the backward pass
            float* gp = gpre_out + (long)prow * 4 * H + punit;
            gp[0] = pf; gp[H] = pi; gp[2 * H] = pc; gp[3 * H] = po;
        }
        const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
        const float cn = cpv * fg + ig * cg;
        const float hn = og * tanhf(cn);
        c_out[(long)prow * H + punit] = cn;
        h_out[(long)prow * H + punit] = hn;
        if (hp_out) hp_out[(long)prow * H + punit] = sigmoidf_(fnext) * hn;      // the next step's gated state
    }
}

// One PU step: hp_in = gated state of this step (zeros at t = 0), F_next = the next step's gate logits F_{t+1}[:, 0:H] (row stride
// ldf_next) or nullptr at the last step, hp_out = where the next step's gated state goes (not hp_in: blocks finish at different times).
static inline void pu_step_launch(hipStream_t s, int B, int H, const float* hp_in, const float* Gin_t, const float* Whh, const float* bhh,
                                  const float* c_prev, float* c_out, float* h_out, const float* F_next, int ldf_next, float* hp_out,
                                  float* gpre_out) {
    hipLaunchKernelGGL(pu_step_r16_kernel, dim3(((B + 15) / 16) * 16), dim3(1024), 0, s, hp_in, Gin_t, Whh, bhh, c_prev, c_out, h_out, F_next,
                       ldf_next, hp_out, gpre_out, B, H);
}

// The whole J-step recurrence of one PU layer in ONE launch.  What a step kernel re-reads every step -- its 256 KiB slice of Whh, at the
// 66-73 GB/s per CU an XCD's L2 serves (4 us of a 14 us step, 8 us as measured with half-used lines) -- lives in the waves' registers
// here (64 VGPRs per lane for UT = 2), and the launch boundary becomes a hand-off among the workgroups that share a row block:
// the only cross-workgroup dependence of step t is on the gated state hp_t of the SAME 16 rows, all H units (H / (16 UT) workgroups).
//   Hand-off (cdna_hip_programming.md Guideline 16, form R2 "the data is the flag"): every step has its own hp buffer, armed with a
//   sentinel word before the launch; owners store hp write-through (agent-scope atomic stores), consumers re-read their fragments
//   with sc1 loads until no word is the sentinel.  No flag, no L2 write-back / invalidate, no drain-barrier-publish-poll sequence.
//   (Earlier forms, per step at B = 256: release / acquire fences from every wave 85 us, from one wave 19.7; one atomic counter per row
//   block 18.1 -- 16 XCD-crossing adders queue on a line; one flag word per workgroup 9.9; this 9.5.)
// Workgroups of a row block must be co-resident (they wait for each other): the launcher sizes the grid from the occupancy query and
// walks larger batches in row chunks; the spin is bounded so that a wave always reaches its exit.
// Arithmetic, k order and reduction order are those of pu_step_r16_kernel (bit-identical results).
constexpr unsigned PU_SENTINEL = 0x7fc0deadu;   // "not stored yet" in the gated-state buffers (a NaN payload arithmetic never produces)
struct PuChain {
    const float* F; long f_step; int ldf;       // gate logits: F + t * f_step + row * ldf + unit  (sigmoid gates h_{t-1})
    const float* G; long g_step;                // Gin_t [rows, 4H] at G + t * g_step (read only: a faulted launch can be redone from it)
    float* GP;                                  // training: the gate pre-activations of step t go to GP + t * g_step (nullptr: not kept)
    const float* Whh; const float* bhh;
    float* C; long c_step;                      // c_t of every step (training) or nullptr
    float* HS; long hs_step;                    // h_t at HS + t * hs_step
    float* HP; long hp_stride;                  // gated state hp_{t+1} at HP + t * hp_stride (J - 1 buffers), PU_SENTINEL-filled on entry
    int rows, H, J;
    unsigned* fault;                            // device word, zero on entry: set by a workgroup whose wait ran out (-> pu_solo_kernel redoes the launch)
    unsigned* fault_host;                       // host-mapped word (or nullptr): set by pu_solo_kernel when it had to run, read by the next ABI call
};

template <int UT>
static __global__ __launch_bounds__(1024) void pu_chain_kernel(const PuChain p) {
    constexpr int NBG = 32 / UT;                 // workgroups per row block (H = 512)
    constexpr int PW = 256 * UT;                 // pointwise threads: 16 rows x 16 UT units
    __shared__ f32x4 red[4 * 4 * UT * 64 + 1];   // [quarter][gate][unit tile][lane] + one slot for the workgroup's "gave up" flag
    int* wg_dead = (int*)&red[4 * 4 * UT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int g = wid & 3, kq4 = wid >> 2;
    const int H = p.H;
    if (tid == 0) *wg_dead = 0;
    __syncthreads();
    const int rb = blockIdx.x / NBG, r0 = rb * 16, u0 = (blockIdx.x % NBG) * (16 * UT);
    const int arow = min(r0 + l15, p.rows - 1);
    const int k0 = kq4 * 128;
    const int rl = tid / (16 * UT), ul = tid % (16 * UT);
    const int prow = r0 + rl, punit = u0 + ul;
    const int prc = min(prow, p.rows - 1);
    const bool pw = tid < PW;
    const int idx = (ul >> 4) * 64 + (ul & 15) + 16 * ((rl & 15) >> 2), reg = rl & 3;
    // this wave's slice of Whh: rows u0 + 16 tile + l15 of gate g, k = k0 + 16 i + 4 lg + (0..3)
    f32x4 w[UT][8];
    {
        const float* wp = p.Whh + (long)g * H * H + (long)(u0 + l15) * H + k0 + 4 * lg;
#pragma unroll
        for (int tl = 0; tl < UT; ++tl)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[tl][i] = *(const f32x4*)(wp + (long)tl * 16 * H + 16 * i);
    }
    float bg[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f;
    if (pw) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bg[q] = p.bhh[q * H + punit];
    }
    // A wait that runs out (the row block's workgroups are not co-resident: another process's kernels hold CUs) marks the WHOLE
    // workgroup dead at the next barrier: it stores nothing from then on (no garbage reaches memory, its consumers' waits run out
    // in turn) and raises the launch's fault word, behind which pu_solo_kernel redoes the launch without cross-workgroup waits.
    bool dead = false, reported = false;
    // what does not depend on the state is requested one step ahead, before the wait
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, fnext = 0.f;
    auto fetch = [&](int t) {
        if (pw) {
            const float* gt = p.G + (long)t * p.g_step + (long)prc * 4 * H + punit;
#pragma unroll
            for (int q = 0; q < 4; ++q) gin[q] = gt[(long)q * H];
            fnext = t + 1 < p.J ? p.F[(long)(t + 1) * p.f_step + (long)prc * p.ldf + punit] : 0.f;
        }
    };
    fetch(0);
    for (int t = 0; t < p.J; ++t) {
        f32x4 acc[UT];
#pragma unroll
        for (int tl = 0; tl < UT; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {                             // hp_0 = 0: the recurrent product of the first step is the bias alone
            const float* hp = p.HP + (long)(t - 1) * p.hp_stride + (long)arow * H + k0 + 4 * lg;
            f32x4 a[8];
            // THE DATA IS THE FLAG: hp_t's buffer holds PU_SENTINEL words until its owners have stored it (write-through); each wave
            // re-reads its own fragments with agent-coherent loads (sc1: from memory / the Infinity Cache, never from a cached line)
            // until none of its 32 words is the sentinel.  No flag word, no drain + barrier + publish + poll round trips on the
            // critical path: one store and one load per hand-off.  hp = sigmoid * o * tanh is finite and a computed NaN is the
            // canonical 0x7fc00000, never the sentinel.
            for (int spins = 0;;) {
                asm volatile(
                    "global_load_dwordx4 %0, %8, off sc1\n\t"
                    "global_load_dwordx4 %1, %8, off offset:64 sc1\n\t"
                    "global_load_dwordx4 %2, %8, off offset:128 sc1\n\t"
                    "global_load_dwordx4 %3, %8, off offset:192 sc1\n\t"
                    "global_load_dwordx4 %4, %8, off offset:256 sc1\n\t"
                    "global_load_dwordx4 %5, %8, off offset:320 sc1\n\t"
                    "global_load_dwordx4 %6, %8, off offset:384 sc1\n\t"
                    "global_load_dwordx4 %7, %8, off offset:448 sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
                    : "v"(hp)
                    : "memory");
                bool missing = false;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u) missing |= __float_as_uint(a[i][u]) == PU_SENTINEL;
                if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
                if (dead || ++spins > (1 << 16)) { dead = true; break; }     // the row block is not co-resident: stop waiting
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tl = 0; tl < UT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w[tl][i][u], acc[tl], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tl = 0; tl < UT; ++tl) red[((kq4 * 4 + g) * UT + tl) * 64 + lane] = acc[tl];
        if (dead && lane == 0) *wg_dead = 1;
        __syncthreads();
        if (*wg_dead) {
            if (!reported && tid == 0) __hip_atomic_store(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = reported = true;
        }
        if (pw && prow < p.rows && !dead) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = red[(0 * 4 + q) * (UT * 64) + idx][reg];
#pragma unroll
                for (int wq = 1; wq < 4; ++wq) v += red[(wq * 4 + q) * (UT * 64) + idx][reg];
                pre[q] = v + bg[q];
            }
            const float pf = pre[0] + gin[0], pi = pre[1] + gin[1], pc = pre[2] + gin[2], po = pre[3] + gin[3];
            if (p.GP) {
                float* gt = p.GP + (long)t * p.g_step + (long)prow * 4 * H + punit;
                gt[0] = pf; gt[H] = pi; gt[2L * H] = pc; gt[3L * H] = po;
            }
            const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
            c = c * fg + ig * cg;
            const float hn = og * tanhf(c);
            if (p.C) p.C[(long)t * p.c_step + (long)prow * H + punit] = c;
            p.HS[(long)t * p.hs_step + (long)prow * H + punit] = hn;
            // written through to memory (agent scope): no L2 write-back needed before the other XCDs may read it
            if (t + 1 < p.J)
                __hip_atomic_store(p.HP + (long)t * p.hp_stride + (long)prow * H + punit, sigmoidf_(fnext) * hn, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t + 1 < p.J) {
            fetch(t + 1);
            __syncthreads();                     // every read of red is done before the next step's partial sums are written
        }
    }
}

// The same J-step recurrence WITHOUT cross-workgroup waits: one workgroup owns a row block (16 rows) and walks all H / 32 unit blocks
// of every step itself (Whh re-read from L2 each step: ~16 x slower than the chain).  Launched behind every pu_chain_kernel launch and
// exits at once unless that launch raised its fault word (a row block whose workgroups were not co-resident: the device is shared with
// another process, CU-masked, or being debugged) -- then it redoes the launch from the untouched inputs, so the caller still gets the
// right answer; the host learns about it through the mapped word and stops using the chain for the handle (egotap_abi.hip).
// Same k order, reduction order and pointwise expressions as pu_chain_kernel / pu_step_r16_kernel: bit-identical results.
static __global__ __launch_bounds__(1024) void pu_solo_kernel(const PuChain p) {
    if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    constexpr int UT = 2;
    __shared__ f32x4 red[4 * 4 * UT * 64];       // [quarter][gate][unit tile][lane]   (32 KiB)
    __shared__ float cst[16 * 512];              // cell state of the row block        (32 KiB; H = 512)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int g = wid & 3, kq4 = wid >> 2;
    const int H = p.H;
    const int r0 = blockIdx.x * 16;
    const int arow = min(r0 + l15, p.rows - 1);
    const int k0 = kq4 * 128;
    const int rl = tid / (16 * UT), ul = tid % (16 * UT);
    const int prow = r0 + rl, prc = min(prow, p.rows - 1);
    const bool pw = tid < 256 * UT;
    const int idx = (ul >> 4) * 64 + (ul & 15) + 16 * ((rl & 15) >> 2), reg = rl & 3;
    for (int i = tid; i < 16 * 512; i += 1024) cst[i] = 0.f;
    if (blockIdx.x == 0 && tid == 0 && p.fault_host) __hip_atomic_store(p.fault_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    for (int t = 0; t < p.J; ++t) {
        f32x4 a[8];
        if (t > 0) {
            const float* hp = p.HP + (long)(t - 1) * p.hp_stride + (long)arow * H + k0 + 4 * lg;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const f32x4*)(hp + 16 * i);
        }
        for (int ub = 0; ub < H / (16 * UT); ++ub) {
            const int u0 = ub * 16 * UT, punit = u0 + ul;
            f32x4 acc[UT];
#pragma unroll
            for (int tl = 0; tl < UT; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t > 0) {
                f32x4 w[UT][8];
                const float* wp = p.Whh + (long)g * H * H + (long)(u0 + l15) * H + k0 + 4 * lg;
#pragma unroll
                for (int tl = 0; tl < UT; ++tl)
#pragma unroll
                    for (int i = 0; i < 8; ++i) w[tl][i] = *(const f32x4*)(wp + (long)tl * 16 * H + 16 * i);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int tl = 0; tl < UT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w[tl][i][u], acc[tl], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int tl = 0; tl < UT; ++tl) red[((kq4 * 4 + g) * UT + tl) * 64 + lane] = acc[tl];
            __syncthreads();
            if (pw && prow < p.rows) {
                const float* gt = p.G + (long)t * p.g_step + (long)prc * 4 * H + punit;
                float pre[4], gin[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gin[q] = gt[(long)q * H];
                    float v = red[(0 * 4 + q) * (UT * 64) + idx][reg];
#pragma unroll
                    for (int wq = 1; wq < 4; ++wq) v += red[(wq * 4 + q) * (UT * 64) + idx][reg];
                    pre[q] = v + p.bhh[q * H + punit];
                }
                const float fnext = t + 1 < p.J ? p.F[(long)(t + 1) * p.f_step + (long)prc * p.ldf + punit] : 0.f;
                const float pf = pre[0] + gin[0], pi = pre[1] + gin[1], pc = pre[2] + gin[2], po = pre[3] + gin[3];
                if (p.GP) {
                    float* gp = p.GP + (long)t * p.g_step + (long)prow * 4 * H + punit;
                    gp[0] = pf; gp[H] = pi; gp[2L * H] = pc; gp[3L * H] = po;
                }
                const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
                const float c = cst[rl * 512 + punit] * fg + ig * cg;
                cst[rl * 512 + punit] = c;
                const float hn = og * tanhf(c);
                if (p.C) p.C[(long)t * p.c_step + (long)prow * H + punit] = c;
                p.HS[(long)t * p.hs_step + (long)prow * H + punit] = hn;
                if (t + 1 < p.J) p.HP[(long)t * p.hp_stride + (long)prow * H + punit] = sigmoidf_(fnext) * hn;
            }
            __syncthreads();                     // red is free again
        }
        // the gated state of this step was written by this workgroup's own waves: complete the stores, then everyone may read it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Workgroups of pu_chain_kernel<UT> the device keeps resident at once (0: the chain kernel cannot be used).
template <int UT>
static inline int pu_chain_resident() {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pu_chain_kernel<UT>, 1024, 0) != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}

// One PU layer over all J steps of B rows (p.HP: (J - 1) * hp_stride floats, armed here; p.fault: PU_FAULT_WORDS words, zeroed here, one per chunk).
// resident1/resident2: pu_chain_resident<1/2>().  Returns false when the chain kernel cannot run here (the caller then walks the steps
// with pu_step_launch).  Every chain launch is followed by its pu_solo_kernel, which does nothing unless the launch faulted.
static inline bool pu_chain_launch(hipStream_t s, int resident1, int resident2, PuChain p, int B, int debug_drop = 0) {
    const int nrb = (B + 15) / 16;
    const bool small = nrb * 32 <= resident1 && nrb <= 8;      // few rows: 16 units per workgroup, twice the workgroups
    const int nbg = small ? 32 : 16, resident = small ? resident1 : resident2;
    const int chunk_rb = resident / nbg;
    if (chunk_rb < 1) return false;
    {   // every gated-state word starts as "not stored yet" (a kernel, so a captured graph re-arms it on every replay)
        const long n = (long)(p.J - 1) * p.hp_stride;
        hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned*)p.HP, n, PU_SENTINEL, p.fault);
    }
    for (int rb0 = 0; rb0 < nrb; rb0 += chunk_rb) {
        const int nb = min(chunk_rb, nrb - rb0), row0 = rb0 * 16;
        PuChain q = p;
        // [r4] a fault word per chunk: a chunk whose wait ran out is redone by ITS solo kernel only; the chunks behind it, whose own chain
        // launches ran cleanly, no longer see a raised word and repeat their work at the ~16 x slower solo rate (B = 1024 is four chunks).
        // fault_host stays the sticky summary of the whole call.
        q.fault = p.fault + ((rb0 / chunk_rb) % PU_FAULT_WORDS);
        q.F = p.F + (long)row0 * p.ldf; q.G = p.G + (long)row0 * 4 * p.H; q.HS = p.HS + (long)row0 * p.H; q.HP = p.HP + (long)row0 * p.H;
        if (p.GP) q.GP = p.GP + (long)row0 * 4 * p.H;
        if (p.C) q.C = p.C + (long)row0 * p.H;
        q.rows = min(B - row0, nb * 16);
        const int wgs = nb * nbg - (rb0 + chunk_rb >= nrb ? debug_drop : 0);      // debug_drop: the last row block is starved (test hook)
        if (small) hipLaunchKernelGGL(pu_chain_kernel<1>, dim3(wgs), dim3(1024), 0, s, q);
        else hipLaunchKernelGGL(pu_chain_kernel<2>, dim3(wgs), dim3(1024), 0, s, q);
        hipLaunchKernelGGL(pu_solo_kernel, dim3(nb), dim3(1024), 0, s, q);
    }
    return true;
}

// Pose head.
//   pose_j = Wp . [left_j | right_j | skel_j] + bp  (+ global offset) ; UnrealEgo: head joint = global_mlp[3:6], output LAST.
// posz: [B*2J, hid] position embeddings (eye-major), hseq: [J, B, H] PU output (time-major).
static __global__ __launch_bounds__(256) void pose_head_kernel(const float* __restrict__ posz, const float* __restrict__ hseq,
                                                        const float* __restrict__ Wp, const float* __restrict__ bp,
                                                        const float* __restrict__ Wg, const float* __restrict__ bg,
                                                        float* __restrict__ pose, int B, int J, int hid, int H,
                                                        int estimate_head) {
    // grid (B, J + 1): block (b, j) = joint j of sample b, block (b, J) = the head joint (UnrealEgo).  Wave c < 3 owns coordinate c:
    // it computes the global output it needs itself (offset c for a joint, head coordinate c for the head block -- a 7680-long dot,
    // recomputed per joint: 16 x redundant and free) and then its joint's dot, so no wave waits for another and a B = 1 forward
    // spreads over 17 workgroups instead of running 54 dots on 4 waves (119 us).
    const int b = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
    if (c >= 3 || (j == J && !estimate_head)) return;
    float other = 0.f;
    if (estimate_head) {
        const int o = j == J ? 3 + c : c;
        float s = 0.f;
        if (H == 512 && (((size_t)Wg | (size_t)hseq) & 15) == 0) {      // (wave-uniform; a 4-byte-aligned view of either takes the scalar loop)
            // [r4] The 7680-long dot streams 61 KB per wave that nobody has touched yet: every (load, load, fma) round of the scalar loop below
            // was a full miss (48 us of a 1.4 ms B = 1 forward).  Five rows of h and of W are requested at once (20 x 16 bytes per lane),
            // three round trips in all.  Every batch size runs this same order, so rows stay bit-identical across batch sizes.
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < J; j0 += 5) {
                f32x4 a[5][2], w[5][2];
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int jj = min(j0 + r, J - 1);
                    const float* hrow = hseq + ((long)jj * B + b) * H + 4 * lane;
                    const float* wrow = Wg + (long)o * J * H + (long)jj * H + 4 * lane;
                    a[r][0] = *(const f32x4*)hrow; a[r][1] = *(const f32x4*)(hrow + 256);
                    w[r][0] = *(const f32x4*)wrow; w[r][1] = *(const f32x4*)(wrow + 256);
                }
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    if (j0 + r < J) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] += a[r][q][e] * w[r][q][e];
                    }
                }
            }
            s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        } else {
            for (int jj = 0; jj < J; ++jj) {
                const float* hrow = hseq + ((long)jj * B + b) * H;
                const float* wrow = Wg + (long)o * J * H + (long)jj * H;
                for (int u = lane; u < H; u += 64) s += hrow[u] * wrow[u];
            }
        }
        s = wave_sum(s);
        other = s + bg[o];
    }
    if (j == J) {
        if (lane == 0) pose[((long)b * (J + 1) + J) * 3 + c] = other;
        return;
    }
    const int kin = 2 * hid + H;
    const float* wrow = Wp + (long)c * kin;
    const float* left = posz + ((long)b * 2 * J + j) * hid;
    const float* right = posz + ((long)b * 2 * J + J + j) * hid;
    const float* hrow = hseq + ((long)j * B + b) * H;
    float s = 0.f;
    for (int u = lane; u < hid; u += 64) s += left[u] * wrow[u] + right[u] * wrow[hid + u];
    for (int u = lane; u < H; u += 64) s += hrow[u] * wrow[2 * hid + u];
    s = wave_sum(s);
    if (lane == 0) pose[((long)b * (J + (estimate_head ? 1 : 0)) + j) * 3 + c] = s + bp[c] + other;
}


// ----------------------------------------------------------------------------- backward of the propagation units
// One recurrent step, reverse time.  Pointwise part A (LSTM gates): from the total gradient on h_t and c_t to the
// gradient of the four gate pre-activations and of c_{t-1}.  The matrix part dhp = dG . Whh runs on the GEMM kernel
// (transposed weights); pointwise part B turns dhp into the gradient of h_{t-1} and of the x2f "forget" logits.
static __global__ __launch_bounds__(256) void pu_gates_bwd_kernel(const float* __restrict__ gpre, const float* __restrict__ c_prev,
                                                           const float* __restrict__ c_t, const float* __restrict__ dh_ext,
                                                           const float* __restrict__ dh_rec, const float* __restrict__ dc_next,
                                                           float* __restrict__ dG, float* __restrict__ dc_prev, int B, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H) return;
    const int row = (int)(i / H), u = (int)(i - (long)row * H);
    const float* gp = gpre + (long)row * 4 * H + u;
    const float f = sigmoidf_(gp[0]), ig = sigmoidf_(gp[H]), g = tanhf(gp[2 * H]), o = sigmoidf_(gp[3 * H]);
    const float tc = tanhf(c_t[i]);
    float dh = 0.f;
    if (dh_ext) dh += dh_ext[i];
    if (dh_rec) dh += dh_rec[i];
    float dc = dh * o * (1.0f - tc * tc);
    if (dc_next) dc += dc_next[i];
    float* dg = dG + (long)row * 4 * H + u;
    dg[0] = dc * c_prev[i] * f * (1.0f - f);
    dg[H] = dc * g * ig * (1.0f - ig);
    dg[2 * H] = dc * ig * (1.0f - g * g);
    dg[3 * H] = dh * tc * o * (1.0f - o);
    dc_prev[i] = dc * f;
}

// hp = sigmoid(f) * h_prev :  dh_prev = dhp * sigmoid(f) ;  df = dhp * h_prev * sigmoid'(f) ;  also emits hp (the
// operand of the h2h weight gradient)
static __global__ __launch_bounds__(256) void pu_hp_bwd_kernel(const float* __restrict__ dhp, const float* __restrict__ F_t, int ldf,
                                                        const float* __restrict__ h_prev, float* __restrict__ dh_rec_prev,
                                                        float* __restrict__ dF_t, int lddf, float* __restrict__ hp_out, int B, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H) return;
    const int row = (int)(i / H), u = (int)(i - (long)row * H);
    const float s = sigmoidf_(F_t[(long)row * ldf + u]);
    const float hprev = h_prev[i], d = dhp[i];
    dh_rec_prev[i] = d * s;
    dF_t[(long)row * lddf + u] = d * hprev * s * (1.0f - s);
    hp_out[i] = s * hprev;
}

// bridge gate of layer 0: b' = sigmoid(fb) * bridge ;  dfb = db' * bridge * sigmoid'(fb) (written into dF[:, H:]) ;
// dbridge = db' * sigmoid(fb) scattered back to the eye-major embedding layout [(b*2 + eye)*J + t, hid]
static __global__ __launch_bounds__(256) void pu_bridge_bwd_kernel(const float* __restrict__ dbp, const float* __restrict__ F, int ldf,
                                                            int fcol0, const float* __restrict__ rotz, float* __restrict__ dF,
                                                            float* __restrict__ drotz, int B, int J, int hid) {
    const int x = 2 * hid;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)J * B * x) return;
    const int m = (int)(i / x), k = (int)(i - (long)m * x);
    const int t = m / B, b = m - t * B, eye = k / hid, c = k - eye * hid;
    const long zi = ((long)(b * 2 + eye) * J + t) * hid + c;
    const float s = sigmoidf_(F[(long)m * ldf + fcol0 + k]);
    const float d = dbp[i], br = rotz[zi];
    dF[(long)m * ldf + fcol0 + k] = d * br * s * (1.0f - s);
    drotz[zi] = d * s;
}

// dposz[(b*2+eye)*J + t, c] = dxs[t*B + b, eye*hid + c] + dpose_feat contribution (already in dposz if accumulate)
static __global__ __launch_bounds__(256) void stereo_scatter_kernel(const float* __restrict__ dxs, float* __restrict__ dz, int B, int J,
                                                             int hid, int accumulate) {
    const int x = 2 * hid;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)J * B * x) return;
    const int m = (int)(i / x), k = (int)(i - (long)m * x);
    const int t = m / B, b = m - t * B, eye = k / hid, c = k - eye * hid;
    const long zi = ((long)(b * 2 + eye) * J + t) * hid + c;
    dz[zi] = (accumulate ? dz[zi] : 0.f) + dxs[i];
}

// Pose head backward.  Data gradients: one block per sample.
//   dpos[(b*2+eye)*J + j, c] = sum_k dpose'[b,j,k] Wp[k, eye*hid + c] ;  dskel[j,b,u] = sum_k dpose'[b,j,k] Wp[k, 2hid + u]
//                              + sum_o dother[b,o] Wg[o, j*H + u]   with dother[0:3] = sum_j dpose[b,j,:], dother[3:6] = dpose[b,J,:]
static __global__ __launch_bounds__(256) void pose_head_bwd_data_kernel(const float* __restrict__ dpose, const float* __restrict__ Wp,
                                                                 const float* __restrict__ Wg, float* __restrict__ dposz,
                                                                 float* __restrict__ dhseq, int B, int J, int hid, int H,
                                                                 int estimate_head) {
    __shared__ float dp[64 * 3], dother[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int JO = J + (estimate_head ? 1 : 0);
    for (int i = tid; i < JO * 3; i += 256) dp[i] = dpose[(long)b * JO * 3 + i];
    __syncthreads();
    if (tid < 6) {
        float s = 0.f;
        if (estimate_head) {
            if (tid < 3) for (int j = 0; j < J; ++j) s += dp[j * 3 + tid];
            else s = dp[J * 3 + tid - 3];
        }
        dother[tid] = s;
    }
    __syncthreads();
    const int kin = 2 * hid + H;
    for (int i = tid; i < J * kin; i += 256) {
        const int j = i / kin, k = i - j * kin;
        float s = dp[j * 3] * Wp[k] + dp[j * 3 + 1] * Wp[kin + k] + dp[j * 3 + 2] * Wp[2 * kin + k];
        if (k < 2 * hid) {
            const int eye = k / hid, c = k - eye * hid;
            dposz[((long)(b * 2 + eye) * J + j) * hid + c] = s;
        } else {
            const int u = k - 2 * hid;
            if (estimate_head)
#pragma unroll
                for (int o = 0; o < 6; ++o) s += dother[o] * Wg[(long)o * J * H + (long)j * H + u];
            dhseq[((long)j * B + b) * H + u] = s;
        }
    }
}

// Pose head weight gradients (tiny N): 32 lanes per weight column, lane l takes samples l, l + 32, ...; the lane partials meet in a
// fixed-order butterfly (deterministic).  One thread per column walked all B x J samples serially: 1.2 ms at B = 256, 5 ms at 1024.
static __global__ __launch_bounds__(256) void pose_head_bwd_weight_kernel(const float* __restrict__ dpose, const float* __restrict__ posz,
                                                                   const float* __restrict__ hseq, float* __restrict__ dWp,
                                                                   float* __restrict__ dbp, float* __restrict__ dWg,
                                                                   float* __restrict__ dbg, int B, int J, int hid, int H,
                                                                   int estimate_head, int accumulate) {
    const int kin = 2 * hid + H, JO = J + (estimate_head ? 1 : 0);
    const int col = blockIdx.x * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
    float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool is_w = col < kin, is_g = estimate_head && col >= kin && col < kin + J * H;
    const bool is_b = col == kin + (estimate_head ? J * H : 0);
    if (is_w) {                             // dWp[c][col] = sum_{b,j} dpose[b,j,c] * feat[b,j,col]
        for (int b = bl; b < B; b += 32)
            for (int j = 0; j < J; ++j) {
                float f;
                if (col < 2 * hid) {
                    const int eye = col / hid, c = col - eye * hid;
                    f = posz[((long)(b * 2 + eye) * J + j) * hid + c];
                } else {
                    f = hseq[((long)j * B + b) * H + col - 2 * hid];
                }
                const float* d = dpose + ((long)b * JO + j) * 3;
                s[0] += d[0] * f; s[1] += d[1] * f; s[2] += d[2] * f;
            }
    } else if (is_g) {                      // dWg[o][q] = sum_b dother[b,o] * skel_flat[b,q]
        const int q = col - kin, j = q / H, u = q - j * H;
        for (int b = bl; b < B; b += 32) {
            const float f = hseq[((long)j * B + b) * H + u];
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            for (int jj = 0; jj < J; ++jj) { const float* d = dpose + ((long)b * JO + jj) * 3; d0 += d[0]; d1 += d[1]; d2 += d[2]; }
            const float* dh = dpose + ((long)b * JO + J) * 3;
            s[0] += d0 * f; s[1] += d1 * f; s[2] += d2 * f; s[3] += dh[0] * f; s[4] += dh[1] * f; s[5] += dh[2] * f;
        }
    } else if (is_b) {                      // biases: s[0..2] pose bias, s[3..8] global_mlp bias
        for (int b = bl; b < B; b += 32) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            for (int j = 0; j < J; ++j) { const float* d = dpose + ((long)b * JO + j) * 3; d0 += d[0]; d1 += d[1]; d2 += d[2]; }
            s[0] += d0; s[1] += d1; s[2] += d2;
            if (estimate_head) {
                const float* dh = dpose + ((long)b * JO + J) * 3;
                s[3] += d0; s[4] += d1; s[5] += d2; s[6] += dh[0]; s[7] += dh[1]; s[8] += dh[2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s[i] += __shfl_xor(s[i], off, 64);
    if (bl != 0) return;
    if (is_w) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dWp[c * kin + col] = (accumulate ? dWp[c * kin + col] : 0.f) + s[c];
    } else if (is_g) {
        const int q = col - kin;
#pragma unroll
        for (int o = 0; o < 6; ++o) dWg[(long)o * J * H + q] = (accumulate ? dWg[(long)o * J * H + q] : 0.f) + s[o];
    } else if (is_b) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dbp[c] = (accumulate ? dbp[c] : 0.f) + s[c];
        if (estimate_head)
#pragma unroll
            for (int o = 0; o < 6; ++o) dbg[o] = (accumulate ? dbg[o] : 0.f) + s[3 + o];
    }
}
