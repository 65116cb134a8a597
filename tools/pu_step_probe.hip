// Development probe (not part of the library): times ablations of the PU step kernel in a 15-step dependent chain.
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iegotap_amd/csrc -Iinclude tools/pu_step_probe.hip -o gpurun_out/pu_step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include "pu_chain.h"

// flags: 1 loads, 2 mfma, 4 pointwise math, 8 pointwise operand loads, 16 stores
template <int FL, int NT>
static __global__ __launch_bounds__(NT) void step_kernel(const float* __restrict__ hp_in, const float* __restrict__ Gin_t,
                                                           const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                           const float* c_prev, float* c_out, float* __restrict__ h_out,
                                                           const float* __restrict__ F_next, int ldf_next, float* __restrict__ hp_out,
                                                           float* __restrict__ gpre_out, int B, int H) {
    __shared__ f32x4 red[4 * 4 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int g = wid & 3, kq4 = wid >> 2;
    const int lin = blockIdx.x;
    const int r0 = (lin >> 4) * 16, u0 = (((lin & 7) << 1) | ((lin >> 3) & 1)) * 32;
    const int arow = min(r0 + l15, B - 1);
    const int k0 = kq4 * 128;
    const int prow = r0 + (tid >> 5), punit = u0 + (tid & 31);
    const int prc = min(prow, B - 1);
    const bool pw = tid < 512;
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f}, cpv = 0.f, fnext = 0.f;
    if ((FL & 8) && pw) {
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
    f32x4 a[8], w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (FL & 1) {
            a[i] = *(const f32x4*)(hp + 16 * i);
            w0[i] = *(const f32x4*)(wp + 16 * i);
            w1[i] = *(const f32x4*)(wp + (long)16 * H + 16 * i);
        } else {
            a[i] = f32x4{1.f * lane, 2.f, 3.f, 4.f}; w0[i] = f32x4{1.f, .5f * i, 3.f, 4.f}; w1[i] = f32x4{1.f, 2.f, .25f * lane, 4.f};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (FL & 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w0[i][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w1[i][u], acc[1], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[0] += a[i] * w0[i]; acc[1] += a[i] * w1[i]; }
    }
    red[((kq4 * 4 + g) * 2 + 0) * 64 + lane] = acc[0];
    red[((kq4 * 4 + g) * 2 + 1) * 64 + lane] = acc[1];
    __syncthreads();
    if (pw && prow < B) {
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
        float cn, hn, hpn;
        if (FL & 4) {
            const float fg = sigmoidf_(pf), ig = sigmoidf_(pi), cg = tanhf(pc), og = sigmoidf_(po);
            cn = cpv * fg + ig * cg;
            hn = og * tanhf(cn);
            hpn = sigmoidf_(fnext) * hn;
        } else { cn = cpv * pf + pi * pc; hn = po * cn; hpn = fnext * hn; }
        if (FL & 16) {
            if (gpre_out) {
                float* gp = gpre_out + (long)prow * 4 * H + punit;
                gp[0] = pf; gp[H] = pi; gp[2 * H] = pc; gp[3 * H] = po;
            }
            c_out[(long)prow * H + punit] = cn;
            h_out[(long)prow * H + punit] = hn;
            if (hp_out) hp_out[(long)prow * H + punit] = hpn;
        } else if (cn + hn + hpn == 12345.678f) c_out[0] = 1.f;
    }
}

static __global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 99999) p[0] = 1.f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = 512, J = 15, NF = 768;
    float *F, *G, *W, *b, *HS, *C, *Z, *HPA, *HPB;
    CK(hipMalloc(&F, (size_t)J * B * NF * 4)); CK(hipMalloc(&G, (size_t)J * B * 4 * H * 4)); CK(hipMalloc(&W, (size_t)4 * H * H * 4));
    CK(hipMalloc(&b, 4 * H * 4)); CK(hipMalloc(&HS, (size_t)J * B * H * 4)); CK(hipMalloc(&C, (size_t)B * H * 4)); CK(hipMalloc(&Z, (size_t)B * H * 4));
    CK(hipMalloc(&HPA, (size_t)B * H * 4)); CK(hipMalloc(&HPB, (size_t)B * H * 4));
    std::vector<float> hw(std::max((size_t)J * B * 4 * H, (size_t)4 * H * H));
    srand(1);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(G, hw.data(), (size_t)J * B * 4 * H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(F, hw.data(), (size_t)J * B * NF * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hw.data(), (size_t)4 * H * H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hw.data(), 4 * H * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Z, 0, (size_t)B * H * 4)); CK(hipMemset(C, 0, (size_t)B * H * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = ((B + 15) / 16) * 16;
    auto chain = [&](auto kern, int nt, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < 4; ++it)
                for (int t = 0; t < J; ++t) {
                    const float* hp_in = t == 0 ? Z : ((t & 1) ? HPA : HPB);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 0, s, hp_in, G + (size_t)t * B * 4 * H, W, b, C, C, HS + (size_t)t * B * H,
                                       t + 1 < J ? F + (size_t)(t + 1) * B * NF : nullptr, NF, (t & 1) ? HPB : HPA, (float*)nullptr, B, H);
                }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-34s %7.2f us/step\n", name, best * 1000.f / (4 * J));
    };
    {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < 60; ++it) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(1024), 0, s, C);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-34s %7.2f us/step\n", "empty kernel, same grid", best * 1000.f / 60);
    }
    chain(step_kernel<31, 1024>, 1024, "full");
    if (argc > 2) {
    chain(step_kernel<31 - 1, 1024>, 1024, "no GEMM operand loads");
    chain(step_kernel<31 - 2, 1024>, 1024, "no MFMA");
    chain(step_kernel<31 - 4, 1024>, 1024, "no transcendental");
    chain(step_kernel<31 - 8, 1024>, 1024, "no pointwise operand loads");
    chain(step_kernel<31 - 16, 1024>, 1024, "no stores");
    chain(step_kernel<0, 1024>, 1024, "nothing but LDS reduce");
    }
    // the one-launch chain against the step kernels: same bits, time per step
    {
        float *G2, *HS2, *HP2;
        CK(hipMalloc(&G2, (size_t)J * B * 4 * H * 4)); CK(hipMalloc(&HS2, (size_t)J * B * H * 4)); CK(hipMalloc(&HP2, (size_t)J * B * H * 4));
        CK(hipMemcpy(G2, hw.data(), (size_t)J * B * 4 * H * 4, hipMemcpyHostToDevice));
        // reference: step kernels (library launcher)
        CK(hipMemset(C, 0, (size_t)B * H * 4));
        for (int t = 0; t < J; ++t) {
            const float* hp_in = t == 0 ? Z : ((t & 1) ? HPA : HPB);
            pu_step_launch(s, B, H, hp_in, G + (size_t)t * B * 4 * H, W, b, C, C, HS + (size_t)t * B * H,
                           t + 1 < J ? F + (size_t)(t + 1) * B * NF : nullptr, NF, (t & 1) ? HPB : HPA, nullptr);
        }
        CK(hipStreamSynchronize(s));
        const int r1 = pu_chain_resident<1>(), r2 = pu_chain_resident<2>();
        printf("resident workgroups: UT=1 %d, UT=2 %d\n", r1, r2);
        PuChain pc{F, (long)B * NF, NF, G2, (long)B * 4 * H, W, b, nullptr, 0, HS2, (long)B * H, HP2, (long)B * H, B, H, J, 0};
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < 4; ++it)
                if (!pu_chain_launch(s, r1, r2, pc, B)) { printf("chain kernel unavailable\n"); return 1; }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-34s %7.2f us/step\n", "one-launch chain", best * 1000.f / (4 * J));
        std::vector<float> h1((size_t)J * B * H), h2((size_t)J * B * H);
        CK(hipMemcpy(h1.data(), HS, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), HS2, h2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; double mx = 0;
        for (size_t i = 0; i < h1.size(); ++i) { if (memcmp(&h1[i], &h2[i], 4)) { ++bad; mx = std::max(mx, (double)fabsf(h1[i] - h2[i])); } }
        printf("chain vs steps: %zu of %zu differ (max abs %.3g); h[last] sample %.6f %.6f\n", bad, h1.size(), mx, h1[h1.size() - 1], h2[h2.size() - 1]);
    }
    return 0;
}
