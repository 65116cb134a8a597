// Ground-truth heatmap synthesis on the device: joints -> the lifting head's input tensor [B, 6J, S, S], replacing the
// per-frame CPU work of the reference's data loader (dataloader/data_loader.py:76-215) when training with --use_gt_heatmap:
//   position maps   utils/projection.py:263-279 coord2d_to_heatmap: unit impulse at the truncated pixel of joint j in an
//                   (S+8)^2 image, scipy gaussian_filter(sigma 1, truncate 4 -> 9 taps, float32 image), crop, / 0.15915589
//   limb maps       utils/data.py:175-252 get_limb_data('line'): anti-aliased segment parent -> joint between the ROUNDED
//                   pixel coordinates (skimage.draw.line_aa, assignment not accumulation: the last write of a pixel wins),
//                   gaussian_filter(sigma 1, mode 'constant'), x sigma, then x 2 and x cos / sin of the limb's elevation
//                   theta = atan(dz / |dxy|) taken from the LEFT view for both eyes (data.py:255-262,
//                   data_loader.py:127, 193-199)
//   gt_pixel_length |p - c| + 1 in heatmap pixels, per eye.
// One block per (frame, eye, joint): the impulse map is closed form (the filter is separable and the reflect boundary of
// the padded image never reaches the cropped region), the segment is drawn by one thread into an LDS tile (it is a
// sequential error-diffusion walk of <= 2S steps) and blurred by the whole block in two passes that round to float32
// after each axis exactly like scipy's float32 output arrays do.
// skimage is not installed in the build container, so the line walk restates skimage 0.2x's _line_aa from its published
// source (Zingl's anti-aliased Bresenham): parity of that step is unpinned; everything else is pinned (see oracle).
#pragma once
#include "common.h"

struct GaussTaps { double w[9]; };

__device__ __forceinline__ int line_aa_draw(float* tile, int S, int r0, int c0, int r1, int c1) {
    // rr = first coordinate (x), cc = second (y); limb_heatmap[cc, rr] = 1 - val   (utils/data.py:176-186)
    const int dc = abs(c0 - c1), dr = abs(r0 - r1);
    float err = (float)(dc - dr);
    const int sign_c = c0 < c1 ? 1 : -1, sign_r = r0 < r1 ? 1 : -1;
    const float ed = dc + dr == 0 ? 1.0f : sqrtf((float)(dc * dc + dr * dr));
    int c = c0, r = r0, n = 0;
    auto put = [&](int rr, int cc, float v) {
        if (rr >= 0 && rr <= S - 1 && cc >= 0 && cc <= S - 1) tile[cc * S + rr] = (float)(1.0 - (double)v);
        ++n;
    };
    for (int guard = 0; guard < 8 * S + 16; ++guard) {      // every wave reaches the exit: at most dc + dr + 1 iterations
        put(r, c, fabsf(err - dc + dr) / ed);
        const float err_prime = err;
        const int c_prime = c;
        if (2.0f * err_prime >= (float)-dc) {
            if (c == c1) break;
            if (err_prime + dr < ed) put(r + sign_r, c, fabsf(err_prime + dr) / ed);
            err -= dr;
            c += sign_c;
        }
        if (2.0f * err_prime <= (float)dr) {
            if (r == r1) break;
            if (dc - err_prime < ed) put(r, c_prime + sign_c, fabsf(dc - err_prime) / ed);
            err += dc;
            r += sign_r;
        }
    }
    return n;
}

// grid: B * 2 * J blocks; pts2d_l / pts2d_r [B, J+1, 2] (x, y in the 1024-pixel frame), pose3d [B, J+1, 3], parents [J+1]
static __global__ __launch_bounds__(256) void heatmap_synth_kernel(const float* __restrict__ p2l, const float* __restrict__ p2r,
                                                            const float* __restrict__ p3, const int* __restrict__ parents, int B, int J,
                                                            int S, GaussTaps g, float* __restrict__ hm, float* __restrict__ plength,
                                                            float* __restrict__ theta_out) {
    extern __shared__ float tl[];           // [2][S*S]
    float* t0 = tl;
    float* t1 = tl + S * S;
    const int j = blockIdx.x % J + 1, eye = (blockIdx.x / J) & 1, b = blockIdx.x / (2 * J);
    const int J1 = J + 1, tid = threadIdx.x, HW = S * S;
    const float* p2 = (eye ? p2r : p2l) + (long)b * J1 * 2;
    float* out = hm + (long)b * 6 * J * HW;
    const float scale = (float)S / 1024.0f;      // coord / 1024.0 * res  ==  coord / (1024.0 / res)

    // ---- position map of joint j
    {
        const float x = p2[2 * j] / 1024.0f * S, y = p2[2 * j + 1] / 1024.0f * S;
        const bool on = -4.0f <= y && y < (float)(S + 4) && -4.0f <= x && x < (float)S;
        const int iy = (int)y, ix = (int)x;               // int(): truncation toward zero; margin 4 cancels against the crop
        float* dst = out + (long)(eye * J + (j - 1)) * HW;
        for (int i = tid; i < HW; i += 256) {
            const int r = i / S, c = i - r * S;
            const int dy = r - iy, dx = c - ix;
            float v = 0.f;
            if (on && dy >= -4 && dy <= 4 && dx >= -4 && dx <= 4) {
                const float a0 = (float)g.w[dy + 4];                  // axis 0 pass, rounded to the float32 array
                v = (float)(g.w[dx + 4] * (double)a0);                // axis 1 pass
                v = v / 0.15915589174187972f;
            }
            dst[i] = v;
        }
    }
    // ---- limb parent(j) -> j
    const int par = parents[j];
    const float pxf = p2[2 * par] * scale, pyf = p2[2 * par + 1] * scale, cxf = p2[2 * j] * scale, cyf = p2[2 * j + 1] * scale;
    for (int i = tid; i < HW; i += 256) t0[i] = 0.f;
    __syncthreads();
    if (tid == 0) {
        line_aa_draw(t0, S, (int)rintf(pxf), (int)rintf(pyf), (int)rintf(cxf), (int)rintf(cyf));
        if (plength) plength[((long)b * 2 + eye) * J + (j - 1)] = sqrtf((pxf - cxf) * (pxf - cxf) + (pyf - cyf) * (pyf - cyf)) + 1.0f;
    }
    __syncthreads();
    // elevation angle from the 3D joints (the pelvis offset cancels in the difference); the LEFT view's value serves both eyes
    const float* q = p3 + (long)b * J1 * 3;
    const float lx = q[3 * par] - q[3 * j], ly = q[3 * par + 1] - q[3 * j + 1], lz = q[3 * par + 2] - q[3 * j + 2];
    const float th = atanf(lz / sqrtf(lx * lx + ly * ly));
    if (tid == 0 && eye == 0 && theta_out) theta_out[(long)b * J + (j - 1)] = th;
    // gaussian_filter(mode='constant'): axis 0 then axis 1, float32 after each
    for (int i = tid; i < HW; i += 256) {
        const int r = i / S, c = i - r * S;
        double acc = 0.0;
#pragma unroll
        for (int k = -4; k <= 4; ++k) {
            const int rr = r + k;
            if (rr >= 0 && rr < S) acc += g.w[k + 4] * (double)t0[rr * S + c];
        }
        t1[i] = (float)acc;
    }
    __syncthreads();
    const float cs = cosf(th), sn = sinf(th);
    float* dcos = out + (long)(2 * J + eye * 2 * J + (j - 1)) * HW;
    float* dsin = dcos + (long)J * HW;
    for (int i = tid; i < HW; i += 256) {
        const int r = i / S, c = i - r * S;
        double acc = 0.0;
#pragma unroll
        for (int k = -4; k <= 4; ++k) {
            const int cc = c + k;
            if (cc >= 0 && cc < S) acc += g.w[k + 4] * (double)t1[r * S + cc];
        }
        const float raw = (float)acc * 2.0f;                 // x sigma (= 1), then the loader's x 2
        dcos[i] = raw * cs;
        dsin[i] = raw * sn;
    }
}
