// Microbenchmark: where does an fp32-MFMA GEMM main loop lose the matrix pipe?
// Builds up the loop in stages; each variant does the same number of MFMAs per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: bare MFMA, operands in registers
// MODE 1: + ds_read_b128 fragments (double buffered in registers) from a static LDS image
// MODE 2: + one __syncthreads() per 64 MFMAs
// MODE 3: + 8 ds_write_b128 per 64 MFMAs (register data)
// MODE 4: + 8 global_load_dwordx4 per 64 MFMAs feeding the ds_writes (the full TileA-like loop)
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* __restrict__ out, int iters, long stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int LDK = 36;
    // fill LDS once
    for (int i = tid; i < 2 * 256 * LDK; i += 256) smem[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 fa[2][2], fb[2][2];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) { fa[s][i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i + 1) * 1e-3f; fb[s][i] = f32x4{0.5f, 0.25f, 0.125f, 1.f} * (float)(lane + s + 1) * 1e-3f; }
    const float* Ab = smem + ((wid >> 1) * 64 + l31) * LDK + 4 * lh;
    const float* Bb = smem + 128 * LDK + ((wid & 1) * 64 + l31) * LDK + 4 * lh;
    f32x4 g[8], g2[8];
    for (int i = 0; i < 8; ++i) { g[i] = f32x4{0.1f, 0.2f, 0.3f, 0.4f}; g2[i] = g[i]; }
    const int c4 = tid & 7, r0 = tid >> 3;
    const float* gp = src + (long)blockIdx.x * stride + (long)r0 * 1024 + c4 * 4;
    const float* bbase = src + (long)blockIdx.x * stride;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, 0x7fffffff, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = *(const f32x4*)(gp + (long)i * 32 * 1024 + (it & 31) * 32);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[(t + 1) & 1][i] = *(const f32x4*)(Ab + i * 32 * LDK + 8 * ((t + 1) & 3));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[(t + 1) & 1][j] = *(const f32x4*)(Bb + j * 32 * LDK + 8 * ((t + 1) & 3));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][i][u], fb[t & 1][j][u], acc[i][j], 0, 0, 0);
                if (MODE == 5 || MODE == 6 || MODE == 8) {       // one ds_write per 4 MFMAs during groups 1,2
                    if (t == 2 || t == 3) {
                        const int i = (t - 2) * 4 + u;
                        __builtin_amdgcn_sched_barrier(0);
                        *(f32x4*)(smem + 256 * LDK + (r0 + i * 32) * LDK + c4 * 4) = g[i];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (MODE == 6) {                    // one global load per 4 MFMAs during groups 3,0 (consumed next iteration)
                    if (t == 0 || t == 1) {
                        const int i = t * 4 + u;
                        __builtin_amdgcn_sched_barrier(0);
                        g2[i] = *(const f32x4*)(gp + (long)i * 32 * 1024 + (it & 31) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (MODE == 8) {                    // buffer loads: SGPR descriptor + one 32-bit VGPR offset
                    if (t == 0 || t == 1) {
                        const int i = t * 4 + u;
                        __builtin_amdgcn_sched_barrier(0);
                        const int off = (int)(((long)r0 * 1024 + c4 * 4 + (long)i * 32 * 1024 + (it & 31) * 32) * 4);
                        auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                        g2[i] = *(f32x4*)&v;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (MODE == 9) {                    // LDS-DMA through the buffer path: SGPR descriptor + one 32-bit VGPR offset
                    if (t == 0 || t == 1) {
                        const int i = t * 4 + u;
                        __builtin_amdgcn_sched_barrier(0);
                        const int off = (int)(((long)r0 * 1024 + c4 * 4 + (long)i * 32 * 1024 + (it & 31) * 32) * 4);
                        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + 256 * LDK + (wid * 8 + i) * 256);
                        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (MODE == 7) {                    // LDS-DMA: 8 x 1 KiB global_load_lds per 64 MFMAs, issued in groups 0,1
                    if (t == 0 || t == 1) {
                        const int i = t * 4 + u;
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + (long)i * 32 * 1024 + (it & 31) * 32),
                                                         (__attribute__((address_space(3))) void*)(smem + 256 * LDK + (wid * 8 + i) * 256), 16, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
            if ((MODE == 3 || MODE == 4) && t == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *(f32x4*)(smem + 256 * LDK + (r0 + (i & 7) * 32) * LDK + c4 * 4) = g[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 6 || MODE == 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = g2[i];
        }
        if (MODE == 7 || MODE == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, const float* src, float* out, long stride = 256L * 1024) {
    const size_t lds = 3 * 256 * 36 * 4;
    CHECK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, src, out, iters, stride);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-28s blocks %5d iters %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, iters, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    const size_t nsrc = (size_t)512 * 256 * 1024 + 1024 * 1024;
    CHECK(hipMalloc(&src, nsrc * 4)); CHECK(hipMemset(src, 0, nsrc * 4));
    CHECK(hipMalloc(&out, 4 << 20));
    for (int blocks : {512}) {
        run<0>("bare mfma", blocks, 2000, src, out);
        run<1>("+ds_read frags", blocks, 2000, src, out);
        run<2>("+barrier/64", blocks, 2000, src, out);
        run<3>("+8 ds_write/64", blocks, 2000, src, out);
        run<4>("+8 global_load/64", blocks, 2000, src, out);
        run<5>("interleaved ds_write", blocks, 2000, src, out);
        run<6>("interleaved dsw+gload", blocks, 2000, src, out);
        run<7>("LDS-DMA 8x1KiB/64", blocks, 2000, src, out);
        run<4>("burst gload  (L2-resident)", blocks, 2000, src, out, 0);
        run<6>("interleaved  (L2-resident)", blocks, 2000, src, out, 0);
        run<7>("LDS-DMA      (L2-resident)", blocks, 2000, src, out, 0);
        run<8>("buffer_load  (L2-resident)", blocks, 2000, src, out, 0);
        run<9>("buffer LDS-DMA (L2-resident)", blocks, 2000, src, out, 0);
        run<9>("buffer LDS-DMA 8x1KiB/64", blocks, 2000, src, out);
    }
    return 0;
}
