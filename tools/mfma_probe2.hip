// Does the per-VMEM-instruction bubble depend on the MFMA shape?  32x32x2 (64-cycle) vs 16x16x4 (32-cycle) fp32 MFMA,
// same FLOPs per wave, with 8 interleaved global_load_dwordx4 per 4096 MFMA-cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int SHAPE, int LOADS>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = (lane + i) * 1e-3f; b[i] = (lane * 3 + i) * 1e-3f; }
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    f32x4 g[8];
    for (int i = 0; i < 8; ++i) g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* gp = src + (long)(tid >> 3) * 1024 + (tid & 7) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {        // 8 segments of 512 MFMA-cycles each
            if (SHAPE == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc32[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + u) & 7], b[t], acc32[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s + t) & 7], b[t & 7], acc16[t], 0, 0, 0);
            }
            if (s < LOADS) {
                __builtin_amdgcn_sched_barrier(0);
                g[s] = *(const f32x4*)(gp + (long)s * 32 * 1024 + (it & 31) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LOADS > 0) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) sum += g[i][0];
            a[0] += sum * 1e-30f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc16[i][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int LOADS>
void run(const char* name, const float* src, float* out) {
    const int blocks = 512, iters = 2000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<SHAPE, LOADS>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<SHAPE, LOADS>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 * iters * 8 * (SHAPE == 0 ? 8 * 4096.0 : 16 * 2048.0);
    printf("%-40s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    CHECK(hipMalloc(&src, 64 << 20)); CHECK(hipMemset(src, 0, 64 << 20));
    CHECK(hipMalloc(&out, 4 << 20));
    run<0, 0>("32x32x2 bare", src, out);
    run<0, 8>("32x32x2 + 8 gload / 4096 cyc", src, out);
    run<0, 4>("32x32x2 + 4 gload / 4096 cyc", src, out);
    run<1, 0>("16x16x4 bare", src, out);
    run<1, 8>("16x16x4 + 8 gload / 4096 cyc", src, out);
    run<1, 4>("16x16x4 + 4 gload / 4096 cyc", src, out);
    return 0;
}
