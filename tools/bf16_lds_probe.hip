// Microbenchmark: how much of the bf16 matrix pipe survives the LDS fragment reads, per wave tiling?
//   v_mfma_f32_32x32x16_bf16, fragments = one ds_read_b128 per 32x16 operand block, static LDS image, no global traffic.
//   <TM, TN, NW>: each wave owns TM x TN 32x32 accumulator tiles, NW waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int TM, int TN, int NW, int WRITES, int GL = 0>
__global__ __launch_bounds__(64 * NW, 1) void probe(float* __restrict__ out, int iters, const float* __restrict__ src = nullptr, long stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = 80;                          // 32 bf16 of k (64 B) + 16 B pad per row
    constexpr int ROWS = 512;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < ROWS * ROWB / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3c003c00u + (i % 7);
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* Ab = smem + ((wid * 32 * TM) % 256 + l31) * ROWB + 16 * lh;
    const char* Bb = smem + (256 + (wid * 32 * TN) % 256 + l31) * ROWB + 16 * lh;
    f32x4 w = {1.f, 2.f, 3.f, 4.f};
    f32x4 g[GL > 0 ? GL : 1];
    for (int i = 0; i < (GL > 0 ? GL : 1); ++i) g[i] = w;
    const float* gp = src + (long)blockIdx.x * stride + (long)(tid >> 3) * 4096 + (tid & 7) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(Ab + (i * 32 % 256) * ROWB + ks * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(Bb + (j * 32 % 256) * ROWB + ks * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (WRITES) {      // the staging writes of one 256x256x32 slab: 32 KiB per workgroup
#pragma unroll
            for (int i = 0; i < 32768 / 16 / (64 * NW); ++i) *(f32x4*)(smem + ROWS * ROWB + (tid + i * 64 * NW) * 16) = GL > 0 ? g[i % (GL > 0 ? GL : 1)] : w;
        }
        if (GL > 0) {      // GL dwordx4 global loads per thread per slab, consumed by the next iteration's writes
#pragma unroll
            for (int i = 0; i < GL; ++i) g[i] = *(const f32x4*)(gp + (long)i * 64 * NW / 8 * 4096 + (it & 127) * 32);
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[blockIdx.x * 64 * NW + tid] = s;
}

template <int TM, int TN, int NW, int WRITES, int GL = 0>
void run(const char* name, float* out, const float* src = nullptr, long stride = 0) {
    const int blocks = 256, iters = 4000;
    const size_t lds = 512 * 80 + 32768;
    CHECK(hipFuncSetAttribute((const void*)probe<TM, TN, NW, WRITES, GL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<TM, TN, NW, WRITES, GL>), dim3(blocks), dim3(64 * NW), lds, 0, out, iters, src, stride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<TM, TN, NW, WRITES, GL>), dim3(blocks), dim3(64 * NW), lds, 0, out, iters, src, stride);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * NW * iters * 2.0 * TM * TN * (32.0 * 32 * 16 * 2);
    printf("%-44s %.3f ms  %.0f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4 << 20));
    run<2, 4, 8, 0>("8 waves x (64x128), reads only", out);
    run<2, 4, 8, 1>("8 waves x (64x128), + slab writes", out);
    run<4, 4, 4, 0>("4 waves x (128x128), reads only", out);
    run<4, 4, 4, 1>("4 waves x (128x128), + slab writes", out);
    float* src;
    const size_t nsrc = (size_t)256 * 512 * 4096 + (1 << 22);
    CHECK(hipMalloc(&src, nsrc * 4)); CHECK(hipMemset(src, 0, nsrc * 4));
    run<2, 4, 8, 1, 2>("8w 64x128 + writes + 2 gload/thr (L2)", out, src, 0);
    run<2, 4, 8, 1, 4>("8w 64x128 + writes + 4 gload/thr (L2)", out, src, 0);
    run<2, 4, 8, 1, 8>("8w 64x128 + writes + 8 gload/thr (L2)", out, src, 0);
    run<2, 4, 8, 1, 4>("8w 64x128 + writes + 4 gload/thr (HBM)", out, src, 512L * 4096);
    run<2, 4, 8, 1, 8>("8w 64x128 + writes + 8 gload/thr (HBM)", out, src, 512L * 4096);
    run<4, 4, 4, 1, 8>("4w 128x128 + writes + 8 gload/thr (L2)", out, src, 0);
    run<4, 4, 4, 1, 16>("4w 128x128 + writes + 16 gload/thr (L2)", out, src, 0);
    run<4, 2, 8, 0>("8 waves x (128x64), reads only", out);
    run<2, 2, 16, 0>("16 waves x (64x64), reads only", out);
    return 0;
}
