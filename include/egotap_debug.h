/* egotap_debug.h -- test and measurement hooks of libegotap_hip.so.
 *
 * Not part of the drop-in boundary (include/egotap.h): nothing here is needed to run, train or evaluate the hot path.  The entry
 * points exist for this repo's parity tests (tests/), its bench (bench.py) and its profiling scripts; they are exported by the same
 * library and follow the same conventions (return codes, egotap_last_error, caller's stream).
 */
#ifndef EGOTAP_DEBUG_H
#define EGOTAP_DEBUG_H
#include "egotap.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fault injection ---- */
/* Test hook: launch the one-launch recurrence with its last `n` workgroups missing (0 = off), which starves a row block exactly as a
 * shared device does. */
int egotap_debug_pu_drop_workgroups(egotap_handle h, int n);

/* ---- partial forward (parity tests against the reference's per-layer hidden states, tests/golden/lift_fwd_*.npz) ---- */
/* debugging aid for parity tests: 0 = full forward (default); 1 = return after the embeddings;
 * 2+i = return after ViT layer i.  The state is then readable as intermediate "x". */
int egotap_lift_debug_stop(egotap_handle h, int stage);

/* ---- measurement / test switch (process wide, one definition in the library): which K-tile depth the bf16-storage NT GEMM with plain
 * operands uses: 0 (default) = 64-deep kernel (csrc/gemm_bf16s64.h) where the shape allows, else the 32-deep one (csrc/gemm_bf16s.h);
 * 32 / 64 = always that one (64 fails on shapes it does not take).  Both run the same MFMAs in the same k order: bit-identical results. */
int egotap_debug_gemm_bk(int bk);

/* ---- measurement / test switch (process wide): how the 64-deep GEMM addresses a convolution operand (csrc/gemm_bf16s64.h): 0 (default) = one
 * wave-uniform origin + 32-bit lane offsets where map and zero page lie within 4 GB of each other, 1 = a 64-bit pointer per lane always.  The same
 * bytes are fetched either way: bit-identical results. */
int egotap_debug_conv_addressing(int mode);

/* ---- host-side planning, exposed for unit tests ---- */
/* (test aid, host only: no device call) the number of partial slabs a weight-gradient launch splits its contraction into, and the slabs per split:
 * workgroups in a row on the busiest CU x slabs each + a fixed part per workgroup + the traffic of the slab reduction, within slab_bytes of
 * workspace.  0 when not even one slab of n_floats fits. */
int egotap_debug_wgrad_splits(int tiles, int64_t slabs, int64_t n_floats, size_t slab_bytes, int num_cu, int lds_bytes, double slab_us, double fixed_us,
                              int* per);

/* ---- measurement hooks (bench.py roofline) ---- */
/* when enabled, every GEMM launch of the handle is bracketed by HIP events on the caller's stream */
int egotap_timing_enable(egotap_handle h, int enable);
/* synchronises the recorded events; returns launches, summed milliseconds and summed algorithmic FLOPs
 * of the fp32 GEMM kernel since the last reset, then resets */
int egotap_timing_read(egotap_handle h, int* launches, double* total_ms, double* total_flops);
/* JSON array written by the last egotap_timing_read: one object per GEMM role
 * {"role","kernel","launches","ms","flops"}; the pointer stays valid until the next read */
const char* egotap_timing_detail(egotap_handle h);

#ifdef __cplusplus
}
#endif
#endif /* EGOTAP_DEBUG_H */
