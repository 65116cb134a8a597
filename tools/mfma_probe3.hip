// Microbenchmark: does a DMA-only producer wave spare the MFMA waves the ~56 matrix-pipe cycles a vector-memory instruction costs?
//   4 MFMA waves per workgroup (the loop of mfma_probe.hip MODE 2: fragments from LDS, one barrier per 64 MFMAs) and, in PRODUCER
//   mode, a fifth wave that issues all 32 global_load_lds_dwordx4 of the iteration (8 per MFMA wave in mfma_probe MODE 7).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: no loads; 1: every MFMA wave issues its 8 DMAs (interleaved, as mfma_probe MODE 7); 2: a fifth wave issues all 32
template <int MODE>
__global__ __launch_bounds__(320, 2) void probe(const float* __restrict__ src, float* __restrict__ out, int iters, long stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int LDK = 36;
    for (int i = tid; i < 2 * 256 * LDK; i += blockDim.x) smem[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + 256 * LDK);
    const float* gp = src + (long)blockIdx.x * stride + (long)(lane >> 3) * 1024 + (lane & 7) * 4;
    if (MODE == 2 && wid == 4) {          // producer wave
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float* g = gp + (long)i * 8 * 1024 + (it & 31) * 32;
                asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds0 + i * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wid == 4) {                        // modes 0, 1: the fifth wave only keeps the barrier count
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 fa[2][2], fb[2][2];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) { fa[s][i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i + 1) * 1e-3f; fb[s][i] = f32x4{0.5f, 0.25f, 0.125f, 1.f} * (float)(lane + s + 1) * 1e-3f; }
    const float* Ab = smem + ((wid >> 1) * 64 + l31) * LDK + 4 * lh;
    const float* Bb = smem + 128 * LDK + ((wid & 1) * 64 + l31) * LDK + 4 * lh;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[(t + 1) & 1][i] = *(const f32x4*)(Ab + i * 32 * LDK + 8 * ((t + 1) & 3));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[(t + 1) & 1][j] = *(const f32x4*)(Bb + j * 32 * LDK + 8 * ((t + 1) & 3));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][i][u], fb[t & 1][j][u], acc[i][j], 0, 0, 0);
                if (MODE == 1 && (t == 0 || t == 1)) {
                    const int i = wid * 8 + t * 4 + u;
                    __builtin_amdgcn_sched_barrier(0);
                    const float* g = gp + (long)i * 8 * 1024 + (it & 31) * 32;
                    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds0 + i * 1024)) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, const float* src, float* out, long stride) {
    const size_t lds = 256 * 36 * 4 * 2 + 32 * 1024;
    CHECK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(320), lds, 0, src, out, iters, stride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(320), lds, 0, src, out, iters, stride);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    const size_t nsrc = (size_t)512 * 256 * 1024 + 1024 * 1024;
    CHECK(hipMalloc(&src, nsrc * 4)); CHECK(hipMemset(src, 0, nsrc * 4));
    CHECK(hipMalloc(&out, 4 << 20));
    run<0>("no loads (5th wave idle)", 512, 2000, src, out, 0);
    run<1>("each MFMA wave issues its 8 DMAs (L2)", 512, 2000, src, out, 0);
    run<2>("producer wave issues all 32 DMAs (L2)", 512, 2000, src, out, 0);
    run<1>("each MFMA wave issues its 8 DMAs (HBM)", 512, 2000, src, out, 256L * 1024);
    run<2>("producer wave issues all 32 DMAs (HBM)", 512, 2000, src, out, 256L * 1024);
    return 0;
}
