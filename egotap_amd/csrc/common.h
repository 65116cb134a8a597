// Shared host/device helpers for the egotap_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "egotap.h"
#include "egotap_debug.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define EGOTAP_WAVE 64

// thread-local error text returned by egotap_last_error()
void egotap_set_error(const char* fmt, ...);

#define EGO_CHECK(cond, ...)                  \
    do {                                      \
        if (!(cond)) {                        \
            egotap_set_error(__VA_ARGS__);    \
            return EGOTAP_ERR_INVALID;        \
        }                                     \
    } while (0)

#define EGO_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            egotap_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                             __FILE__, __LINE__);                                      \
            return EGOTAP_ERR_HIP;                                                     \
        }                                                                              \
    } while (0)


// erf for the GELU epilogues (modeling_vit.py:320-327 ACT2FN["gelu"], exact-erf GELU): Abramowitz-Stegun 7.1.26,
// erf|x| = 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1/(1 + p|x|): 13 VALU instructions (one v_rcp_f32, one v_exp_f32)
// against ~45 for ocml's erff.  In fp32 arithmetic: max |error| 5.5e-7 on erf; GELU(x) = x/2 (1 + erf(x/sqrt 2)) is
// within 4.7e-7 of float64 over [-8, 8], the same as the fp32 rounding of the final product with an exact erf (4.5e-7).
// The epilogue runs with the matrix pipe idle (all waves of a block reach it together): on the bf16x3 GEMM the erff
// GELU was a quarter of the MLP-up launch.  exp(-x^2) is returned too: the GELU derivative needs exp(-z^2/2) = exp(-(z/sqrt 2)^2).
__device__ __forceinline__ float fast_erf_exp(float x, float& e) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    e = __builtin_amdgcn_exp2f(-(ax * ax) * 1.4426950408889634f);
    return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + fast_erf_exp(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float dgelu_erf(float z) {      // d/dz GELU(z) = Phi(z) + z phi(z)
    float e;
    const float er = fast_erf_exp(z * 0.70710678118654752440f, e);
    return fmaf(z * e, 0.39894228040143267794f, 0.5f * (1.0f + er));
}

// A vector of N floats that may be split over up to three equally long
// segments (the ViT keeps query/key/value as three nn.Linear; we run them as
// one GEMM without packing, so weights stay the caller's live tensors).
struct SegVec {
    const float* p[3];
    int seg;  // floats per segment
    __device__ __forceinline__ float at(int n) const { return p[n / seg][n % seg]; }
    // four consecutive values n0 .. n0 + 3 (n0 and seg multiples of 4: they lie in one segment).  The segment pointer is SELECTED, not
    // indexed (a lane-dependent index into the kernel-argument struct is a gather from memory and a dependent wait), and the four values
    // come as one 16-byte load where the pointer allows it -- [r4] the epilogues that built them from four at() calls kept 16 loads, the
    // pointer loads in front of them and their waits in the half-block loop of gemm_f32_dma.h.
    __device__ __forceinline__ f32x4 at4(int n0) const {
        const int k = n0 >= 2 * seg ? 2 : (n0 >= seg ? 1 : 0);
        const float* q = (k == 2 ? p[2] : (k == 1 ? p[1] : p[0])) + (n0 - k * seg);
        if (((size_t)q & 15) == 0) return *(const f32x4*)q;
        return f32x4{q[0], q[1], q[2], q[3]};
    }
};

// [N,K] row-major weight (nn.Linear layout) split the same way along N.
struct SegMat {
    const float* p[3];
    int seg;  // rows per segment
    long ld;  // row stride (floats)
    __device__ __forceinline__ const float* row(int n) const { return p[n / seg] + (long)(n % seg) * ld; }
};

// the same weight matrix already rounded to bf16 (plain-bf16 mode: a scratch copy made right before the launch, so the W operand
// costs half the vector-memory bytes -- the 64 B/clk/CU L1 path, not the matrix pipe, bounds that mode)
struct SegMatB {
    const __bf16* p[3];
    int seg;
    long ld;  // row stride (bf16 elements)
    __device__ __forceinline__ const __bf16* row(int n) const { return p[n / seg] + (long)(n % seg) * ld; }
};

static inline SegVec segvec1(const float* p, int n) {
    SegVec v;
    v.p[0] = p; v.p[1] = p; v.p[2] = p; v.seg = n > 0 ? n : 1;
    return v;
}
static inline SegMat segmat1(const float* p, int n, long ld) {
    SegMat v;
    v.p[0] = p; v.p[1] = p; v.p[2] = p; v.seg = n > 0 ? n : 1; v.ld = ld;
    return v;
}

// blocks are dealt round-robin over the 8 XCDs: give every XCD a contiguous chunk of the linear work order (bijective for any grid;
// placement affects speed only)
__device__ __forceinline__ int xcd_lin(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = bid & 7;
    return (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (bid >> 3);
}

// XCD-aware block -> tile map.  Blocks are dealt round-robin over the 8 XCDs
// (blockIdx % 8 labels the group that shares an L2), so give every group a
// contiguous chunk of a grouped tile order: inside a group of GM tile-rows
// walk M fastest, then N.  Bijective for any grid size; placement affects
// speed only (cdna_hip_programming.md T1).
__device__ __forceinline__ void xcd_tile(int bid, int nblk, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    const int per_group = GM * tiles_n;
    const int g = lin / per_group;
    const int first = g * GM;
    const int gsz = min(tiles_m - first, GM);
    const int in = lin - g * per_group;
    tm = first + in % gsz;
    tn = in / gsz;
}
