// Microbenchmark [r5]: what does one LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) cost the wave that issues it, next to MFMAs,
// in its two address forms -- a 64-bit pointer per lane (`v[ptr], off`) and a wave-uniform base in scalar registers + a 32-bit lane offset
// (`v_off, s[base]`)?  Every wave runs  iters x [ NM x v_mfma_f32_16x16x32_bf16 (registers only) ; ND DMA instructions ; s_waitcnt vmcnt(ND) ],
// the sources are L2-resident (64 KB per workgroup, re-read) and shaped like a GEMM operand's K-tile -- one instruction = 8 rows x 128 bytes, rows 1 KB apart -- the
// LDS destinations rotate over 16 KB per wave.  Cycles per DMA =
// (t(ND) - t(0)) / ND at the clock the run held (reported by the ND = 0 line: 16 cycles per MFMA per wave when the pipe is saturated).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_issue_probe tools/dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NW, int NM, int ND, bool SBASE>
__global__ __launch_bounds__(64 * NW, 1) void probe(float* __restrict__ out, const char* __restrict__ src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + 0.001f * lane); b[e] = (__bf16)(0.5f); }
    const char* wg = src + (size_t)blockIdx.x * 65536;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wid * 16384;
    const unsigned voff = (lane >> 3) * 1024 + (lane & 7) * 16;      // row lane >> 3 (1 KB apart), 16-byte chunk lane & 7
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const unsigned piece = (unsigned)((it * ND + d) & 15);
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + piece * 1024);
            if constexpr (SBASE) {
                const unsigned long long v = (unsigned long long)(size_t)(wg + (wid & 7) * 8192 + (piece & 7u) * 128);
                unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(la) : "memory");
            } else {
                const char* g = wg + (wid & 7) * 8192 + (piece & 7u) * 128 + voff;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory");
            }
        }
        if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[blockIdx.x * 64 * NW + tid] = s + smem[tid];
}

template <int NW, int NM, int ND, bool SBASE>
double run(float* out, const char* src) {
    const int blocks = 256, iters = 20000;
    const size_t lds = (size_t)NW * 16384;
    auto k = probe<NW, NM, ND, SBASE>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, out, src, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, out, src, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)ms * 1e6 / iters;      // ns per iteration
}

template <int NW, int NM>
void sweep(float* out, const char* src) {
    const double t0 = run<NW, NM, 0, true>(out, src);
    const double ghz = (double)NM * 16.0 * (NW / 4) / t0;      // NW / 4 waves per SIMD share the pipe: NM x 16 cycles each
    printf("%d waves per workgroup (%d per SIMD), %d MFMAs per wave and iteration: %.1f ns per iteration without DMA (%.2f GHz if the pipe is full)\n", NW, NW / 4, NM, t0, ghz);
    const double f2 = run<NW, NM, 2, false>(out, src), s2 = run<NW, NM, 2, true>(out, src);
    const double f4 = run<NW, NM, 4, false>(out, src), s4 = run<NW, NM, 4, true>(out, src);
    const double f8 = run<NW, NM, 8, false>(out, src), s8 = run<NW, NM, 8, true>(out, src);
    const double c = ghz;      // cycles per ns
    printf("   DMA per iteration    pointers: ns (cycles per DMA and SIMD)     scalar base: ns (cycles per DMA and SIMD)\n");
    printf("   2                    %7.1f (%5.1f)                              %7.1f (%5.1f)\n", f2, (f2 - t0) * c / (2 * (NW / 4)), s2, (s2 - t0) * c / (2 * (NW / 4)));
    printf("   4                    %7.1f (%5.1f)                              %7.1f (%5.1f)\n", f4, (f4 - t0) * c / (4 * (NW / 4)), s4, (s4 - t0) * c / (4 * (NW / 4)));
    printf("   8                    %7.1f (%5.1f)                              %7.1f (%5.1f)\n", f8, (f8 - t0) * c / (8 * (NW / 4)), s8, (s8 - t0) * c / (8 * (NW / 4)));
}

int main() {
    float* out; char* src;
    CHECK(hipMalloc(&out, 4 << 20));
    CHECK(hipMalloc(&src, (size_t)256 * 65536));
    CHECK(hipMemset(src, 0, (size_t)256 * 65536));
    sweep<4, 32>(out, src);
    sweep<4, 64>(out, src);
    sweep<8, 32>(out, src);
    sweep<8, 64>(out, src);
    return 0;
}
