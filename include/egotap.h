/* egotap.h -- C ABI of libegotap_hip.so: EgoTAP's heatmap -> 3D lifting hot path on MI355X (gfx950).
 *
 * The reference (tho-kn/EgoTAP) is pure Python/PyTorch and has no FFI of its own; the boundary it
 * offers is the nn.Module call.  Each entry point below names the reference interface it replaces
 * (paths under the reference repo).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Contract
 *   - plain pointers and sizes only; every pointer marked "device" is a HIP device pointer owned
 *     by the caller (PyTorch allocates params, activations, workspace); the library borrows them.
 *   - no device allocation, no synchronisation, no internal streams: all work is enqueued on the
 *     caller's stream (pass torch.cuda.current_stream().cuda_stream as a void*).  Graph-capture safe
 *     after one eager call per handle (the first forward asks the device for its occupancy figures).
 *   - every function returns 0 on success, an EGOTAP_ERR_* code otherwise; the message is in
 *     egotap_last_error() (thread local).  No C++ exception crosses the ABI.
 *   - a handle is not thread-safe; data parallelism = one process + one handle per GPU.
 *   - the library reads no environment variable.  Test and measurement hooks (fault injection, partial forward, GEMM event timing)
 *     are exported too but declared separately, in egotap_debug.h: nothing a deployment calls.
 */
#ifndef EGOTAP_H
#define EGOTAP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): egotap_config grew (hm_blocks, round 3); test / measurement hooks moved to egotap_debug.h; egotap_pose_metrics_batch_axes added */
#define EGOTAP_ABI_VERSION 2

enum { EGOTAP_OK = 0, EGOTAP_ERR_INVALID = 1, EGOTAP_ERR_HIP = 2, EGOTAP_ERR_UNBOUND = 3, EGOTAP_ERR_WORKSPACE = 4 };

/* which of the wrapper's three networks a parameter belongs to
 * (model/egotap_autoencoder_model.py:109-111: net_AutoEncoder, net_HeatMap, net_RotHeatMap) */
enum { EGOTAP_NET_LIFT = 0, EGOTAP_NET_HM_POS = 1, EGOTAP_NET_HM_ROT = 2, EGOTAP_NET_COUNT = 3 };

enum { EGOTAP_F32 = 0, EGOTAP_I64 = 1 };

/* Options that shape the networks; mirrors the reference flags
 * --joint_preset/--num_heatmap/--ae_hidden_size/--load_size_heatmap (options/base_options.py:52-66,
 * options/dataset_options.py:29-41) and the fixed ViT/PU sizes of model/net_architecture.py:340-348, 650. */
typedef struct egotap_config {
    int32_t struct_bytes;   /* sizeof(egotap_config), for ABI checking */
    int32_t n_joints_hm;    /* heatmaps per eye: 15 UnrealEgo, 17 EgoCap */
    int32_t estimate_head;  /* 1: UnrealEgo (head joint + global offset from global_mlp), 0: EgoCap */
    int32_t hm_size;        /* heatmap side (64; 128 for 512x512 RGB) */
    int32_t hidden;         /* --ae_hidden_size (128) */
    int32_t vit_dim;        /* 1024 */
    int32_t vit_heads;      /* 8 */
    int32_t vit_layers;     /* 3 */
    int32_t patch;          /* 16 */
    int32_t pu_hidden;      /* 512 */
    int32_t hm_blocks[4];   /* BasicBlocks per ResNet stage of the heatmap estimators (--model_name, net_architecture.py:57-64):
                             * {2,2,2,2} resnet18 (all zeros mean this), {3,4,6,3} resnet34; at most 6 per stage */
} egotap_config;

typedef struct egotap_handle_s* egotap_handle;

/* library */
int egotap_abi_version(void);
const char* egotap_last_error(void);

/* replaces network.define_AutoEncoder / define_HeatMap (model/network.py:11-33): host-side state only */
int egotap_create(const egotap_config* cfg, egotap_handle* out);
void egotap_destroy(egotap_handle h);

/* replaces nn.Module parameter ownership: bind one state_dict entry (the reference's own key,
 * SURVEY.md Appendix B, e.g. "pos_heatmap_encoder.vit.encoder.layer.0.attention.attention.query.weight")
 * to a device pointer.  numel and dtype are checked against the expected shape. */
int egotap_bind_param(egotap_handle h, int net, const char* state_dict_key, void* dev_ptr, int64_t numel, int dtype);
/* number of state_dict entries the forward of `net` needs that are not bound yet (0 = ready) */
int egotap_unbound_count(egotap_handle h, int net, int* count);

/* EgoTAPAutoEncoder.forward / predict_pose (model/net_architecture.py:679-758), eval mode:
 *   hm    device f32 [B, 6*n_joints_hm, hm_size, hm_size]  (L pos, R pos, L cos, L sin, R cos, R sin)
 *   pose  device f32 [B, n_joints_hm + estimate_head, 3]   (head joint last)
 *   ws    device scratch of at least egotap_lift_workspace_bytes(B), 256-byte aligned
 * The reference's three all-zero outputs (rot, indep_pos, reconstructed heatmaps; :718-719, 756) are
 * not computed here: the host mirror returns cached zeros. */
int egotap_lift_workspace_bytes(egotap_handle h, int B, size_t* bytes);
int egotap_lift_forward(egotap_handle h, const float* hm, int B, float* pose, void* ws, size_t ws_bytes, void* stream);

/* where an intermediate lives inside ws after egotap_lift_forward (for parity tests):
 * name in {"tokens","pos_embed","rot_embed","skel_embed"}; offset in bytes, numel in floats */
int egotap_lift_intermediate(egotap_handle h, int B, const char* name, size_t* offset, int64_t* numel);

/* HeatMap_UnrealEgo_Shared.forward(left, right) (model/net_architecture.py:25-173; resnet18 backbone), eval mode:
 *   net    EGOTAP_NET_HM_POS (2*n_joints_hm output channels) or EGOTAP_NET_HM_ROT (4*n_joints_hm)
 *   left, right  device f32 [B, 3, 4*hm_size, 4*hm_size]
 *   out    device f32, channel 0 of this net's output; image b starts at out + b*out_image_stride (floats), so the
 *          result can be written straight into a channel slice of the lifting head's input
 *          (torch.cat of egotap_autoencoder_model.py:195-216 is never materialised)
 *   ws     device scratch of at least egotap_hm_workspace_bytes(B), 256-byte aligned (shared by both nets) */
int egotap_hm_workspace_bytes(egotap_handle h, int B, size_t* bytes);
int egotap_hm_forward(egotap_handle h, int net, const float* left, const float* right, int B, float* out,
                      int64_t out_image_stride, void* ws, size_t ws_bytes, void* stream);
/* name in {"layer0".."layer4" (backbone pyramid, images interleaved n = 2b + eye), "conv_up3","conv_up2","conv_up1"} */
int egotap_hm_intermediate(egotap_handle h, int B, const char* name, size_t* offset, int64_t* numel);
/* [r5] The same forward with BATCH-statistics BatchNorm2d and no graph: what the reference's FROZEN estimators compute while the lifting
 * head trains -- train.py:91 model.train() leaves their BatchNorm2d in training mode (model/egotap_autoencoder_model.py:127-129 freezes
 * parameters only, :177-216 calls them under autocast); the shared backbone runs once per eye (model/net_architecture.py:45-50), so each
 * BatchNorm normalises every eye's batch with that eye's statistics and updates its bound buffers twice per call, left then right:
 * running_mean / running_var (momentum 0.1, unbiased variance) and, where bound as EGOTAP_I64, num_batches_tracked += 2.
 * EGOTAP_PREC_BF16 only (bf16 channels-last kernels; the fp32 form is composed from egotap_hmtrain_conv_fwd / egotap_hmtrain_bn2d_fwd).
 * The backbone runs over the whole batch (its statistics couple the frames); the decoder, which has no BatchNorm, runs in pieces of
 * `chunk` frames (0 = the whole batch) so that its scratch stays at the chunk's size.  B >= 2.  Arguments as egotap_hm_forward;
 * ws at least egotap_hm_forward_bnbatch_workspace_bytes(B, chunk). */
int egotap_hm_forward_bnbatch_workspace_bytes(egotap_handle h, int B, int chunk, size_t* bytes);
int egotap_hm_forward_bnbatch(egotap_handle h, int net, const float* left, const float* right, int B, float* out, int64_t out_image_stride,
                              int chunk, void* ws, size_t ws_bytes, void* stream);
/* where a backbone map lives inside ws after egotap_hm_forward_bnbatch (parity tests): name in {"pool0" (stem + max-pool), "layer1" ..
 * "layer4"}; bf16 [B * s * s, 2 C], pixel (b, y, x) = [left C | right C]; offset in bytes, numel in bf16 elements */
int egotap_hm_forward_bnbatch_intermediate(egotap_handle h, int B, int chunk, const char* name, size_t* offset, int64_t* numel);

/* Arithmetic of the large GEMMs of the lifting head (nn.Linear layers of the ViT and fc1; everything else is always fp32).
 *   EGOTAP_PREC_F32     v_mfma_f32_32x32x2_f32: exact fp32 products (default; what the headline benchmark measures)
 *   EGOTAP_PREC_BF16X3  each fp32 operand split in registers into hi + lo bf16 (16 significant bits), a*b taken as
 *                       a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; operands and
 *                       results stay fp32 in HBM.  Error ~2^-16 per product against 2^-24: opt-in fast mode, the reference
 *                       offers the analogous knob as --use_amp (egotap_autoencoder_model.py:177-183).
 *   EGOTAP_PREC_BF16    operands rounded to bf16 in registers (round to nearest even), one MFMA per product, fp32 accumulate,
 *                       fp32 master weights / activations / gradients in HBM: the reduced-precision training configurations
 *                       (the reference trains under fp16 autocast, egotap_autoencoder_model.py:299-323 + --use_amp).
 * The mode also selects the kernels of egotap_train_gemm_nt / egotap_train_gemm_tn (forward, input-gradient and
 * weight-gradient GEMMs of the training step) for shapes the bf16 kernels cover (N, K multiples of 256, M >= 1024). */
enum { EGOTAP_PREC_F32 = 0, EGOTAP_PREC_BF16X3 = 1, EGOTAP_PREC_BF16 = 2 };
int egotap_set_precision(egotap_handle h, int mode);
/* The propagation units' recurrence (custom_cells.py:149-197) runs as ONE launch per layer whose workgroups hand the state to each other
 * inside the launch: they must all be resident together, which holds when the calling process has the device to itself while a
 * forward runs (one process per GPU, the deployment this library is written for).  Where the device is shared -- another process, or
 * another stream of this one running long kernels -- pass enable = 0: the recurrence then runs as one kernel per step (same bits,
 * ~2x the latency of that part).  With the chain enabled on a shared device the waits inside it are bounded and run out: the
 * workgroups concerned store nothing and raise a fault word in the workspace, and a second kernel queued behind every chain launch
 * (idle otherwise) then redoes that launch without cross-workgroup waits -- the call's results are RIGHT (same bits), it only took
 * ~0.1-0.2 s longer.  The library notices at its next call on the handle (a host-mapped word, no synchronisation) and from then on
 * uses the per-step kernels for that handle by itself. */
int egotap_set_pu_chain(egotap_handle h, int enable);
/* *enabled: whether the handle still uses the one-launch recurrence; *faults: chain launches that had to be redone so far (see above).
 * Exact for calls whose stream the caller has synchronised. */
int egotap_pu_chain_status(egotap_handle h, int* enabled, int* faults);
/* EGOTAP_PREC_BF16 only: caller-owned device scratch (16-byte aligned) into which a GEMM's weight matrix is rounded to bf16 right
 * before the launch (stream ordered; nothing is cached, the live fp32 parameters stay the source of truth).  Halves the W operand's
 * vector-memory bytes, which bound that mode.  bytes >= 2 * the largest N*K (67 MB for fc1 of the position encoder); NULL = off. */
int egotap_set_weight_scratch(egotap_handle h, void* buf, size_t bytes);
/* EGOTAP_PREC_BF16 only: caller-owned device scratch (16-byte aligned) into which a GEMM's plain row-major activation operand [M, K]
 * is rounded to bf16 right before the launch; with both operands in bf16 the product runs on the LDS-DMA kernel (gemm_bf16_dma.h,
 * same rounding and summation order as without it -- bit-identical results, about twice the GEMM rate).  bytes >= 2 * the largest
 * M*K (tokens x 4096 for the ViT MLP: 1.2 GB at B = 256); a GEMM whose operand does not fit falls back to the register-staged
 * kernel.  NULL = off. */
int egotap_set_act_scratch(egotap_handle h, void* buf, size_t bytes);


/* ---- single operators (same kernels the forward uses; exported for unit tests and reuse) ---- */
/* y = epi(x W^T + b): nn.Linear (+ residual / exact GELU / BatchNorm1d-eval + LeakyReLU 0.2).
 * epi: 0 bias, 1 bias + residual r[M,N], 2 bias + GELU(erf), 3 bias + BN(eval) + LeakyReLU (bn = gamma,beta,mean,var; eps 1e-5)
 * tile: 0 default, else a tile-shape id (see egotap_gemm_tile_name) */
int egotap_linear_f32(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int epi,
                      const float* r, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                      const float* bn_var, int tile, void* stream);
const char* egotap_gemm_tile_name(int tile);
/* y = x w^T + b with both operands already bf16 (caller-owned copies x [M,K], w [N,K], 16-byte aligned; N % 256 == 0, K % 32 == 0):
 * the LDS-DMA kernel of EGOTAP_PREC_BF16 (gemm_bf16_dma.h), fp32 accumulate, fp32 result */
int egotap_linear_bf16_dma(const void* x_bf16, const void* w_bf16, const float* b, float* y, int M, int N, int K, void* stream);
/* nn.LayerNorm over the last dim (1024), modeling_vit.py:357-358 */
int egotap_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int rows, int dim, float eps,
                         void* stream);
/* ViTSelfAttention core (modeling_vit.py:233-252) on a fused [B*N, 3*heads*128] q|k|v buffer -> ctx [B*N, heads*128].
 * N >= 32, a multiple of 4: every heatmap side the reference allows (a multiple of 16, net_architecture.py:327: N = 36 (side / 16)^2 for
 * UnrealEgo) -- a ragged last 32-key tile is masked, the last 32-query block overlaps its predecessor. */
int egotap_attention_f32(const float* qkv, float* ctx, int B, int N, int heads, void* stream);
/* the same operator with the arithmetic of egotap_set_precision (EGOTAP_PREC_F32 / _BF16X3 / _BF16).  The bf16 / bf16x3 kernels need
 * N % 32 == 0 (heatmap sides 64, 128: every shipped configuration); for other N the call -- and egotap_lift_forward in those modes -- runs the
 * exact-fp32 kernel above instead (a fallback by name: the result is MORE exact than asked for).  The attention BACKWARD
 * (egotap_train_attention_bwd, egotap_lift_backward) needs N % 32 == 0 and says so. */
int egotap_attention(const float* qkv, float* ctx, int B, int N, int heads, int precision, void* stream);

/* Evaluation metrics of EgoTAPAutoEncoderModel.evaluate (model/egotap_autoencoder_model.py:329-350): per-sample MPJPE and
 * Procrustes-aligned MPJPE (utils/util.py:328-379 batch_compute_similarity_transform_torch: 3x3 SVD, reflection fix,
 * scale, translation), one launch for the batch instead of the reference's two Python loops.
 *   pred, gt  device f32 [B, J, 3];  mpjpe, pa_mpjpe  device f32 [B] (input units);  aligned  device f32 [B, J, 3] or NULL */
int egotap_pose_metrics(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pa_mpjpe, float* aligned, void* stream);
/* The same metrics AS THE REFERENCE COMPUTES THEM FOR A BATCH OF 2 OR 3 FRAMES.  utils/util.py:337 decides whether to transpose its
 * [B, J, 3] input by testing shape[0] against 3 and 2 (meant for unbatched 3 x N / 2 x N point sets), so for B = 2 or 3 the similarity
 * transform is solved over the wrong axes (J "coordinates", 3 "points"; a J x J SVD of rank <= 2) and PA-MPJPE of a frame depends on
 * the size of the batch it arrives in.  test.py / utils/evaluate.py:149-168 print exactly these numbers for a ragged last batch, so the
 * wrapper's evaluate() calls this entry for B in {2, 3} by default (opt.pa_mpjpe_reference_batch_axes).
 * B must be 2 or 3; same arguments as egotap_pose_metrics; `aligned` = the reference's S1_hat (not transposed back, as there). */
int egotap_pose_metrics_batch_axes(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pa_mpjpe, float* aligned, void* stream);

/* Ground-truth heatmaps from joints, written in the lifting head's input layout (the data loader's per-frame CPU work when
 * training with --use_gt_heatmap: dataloader/data_loader.py:76-215, utils/projection.py:263-279 coord2d_to_heatmap,
 * utils/data.py:175-262 get_limb_data / overwrite_limb_data).
 *   pts2d_left/right  device f32 [B, J+1, 2]  joints in the 1024-pixel image frame (gt_camera_2d_*), joint 0 = root
 *   pose3d            device f32 [B, J+1, 3]  gt_local_pose (the pelvis offset cancels in the limb direction)
 *   parents           device i32 [J+1]        kinematic parents (utils/util.py:51-52)
 *   hm                device f32 [B, 6J, res, res]: L pos, R pos, L cos, L sin, R cos, R sin
 *   plength           device f32 [B, 2, J] or NULL (gt_pixel_length_left/right);  theta  device f32 [B, J] or NULL */
int egotap_synth_heatmaps(const float* pts2d_left, const float* pts2d_right, const float* pose3d, const int* parents, int B, int J,
                          int res, float* hm, float* plength, float* theta, void* stream);

/* ---- training-step operators (fp32), called by the autograd glue (egotap_amd/training.py) ------------------------------
 * They implement the backward of the modules above plus loss / optimizer (egotap_autoencoder_model.py:284-323,
 * utils/loss.py:54-85, network.py:72-78 AdamW).  All buffers are caller-owned device memory; reductions have a fixed order.
 * loader: 0 plain, 1 ViT patch gather, 2 per-heatmap token regroup, 3 stereo cos/sin gather, 4 stereo joint features,
 *         5 stereo joint features x sigmoid gate (aux).  epi: 0 none, 1 bias, 2 bias + residual r, 3 bias + GELU (stores
 *         the pre-activation to z), 4 accumulate onto r, 5 multiply by GELU'(r). */
int egotap_train_gemm_nt(egotap_handle h, int loader, const float* x, int64_t lda, const float* aux, const float* w, const float* b,
                         float* y, int M, int N, int K, int epi, const float* r, float* z, int Bsz, void* stream);
int egotap_train_gemm_tn(egotap_handle h, int loader, const float* dy, int64_t ldy, const float* x, const float* aux, float* dw, int M,
                         int N, int K, int accumulate, int Bsz, void* ws, size_t ws_bytes, void* stream);
int egotap_train_colsum(const float* y, int64_t ldy, float* out, int M, int N, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* [r3] weight and bias gradient of one nn.Linear with a plain input in one call: dw[N,K] (+)= dy^T x, db[N] (+)= column sums of dy (autograd of
 * the ViT layers' Linear modules, model/modeling_vit.py:226-230, 271, 319-344).  fp32 with M % 32 == 0: the workgroups that stage dy for the
 * product also sum its columns (one pass over dy); otherwise egotap_train_gemm_tn followed by egotap_train_colsum. */
int egotap_train_gemm_tn_bias(egotap_handle h, const float* dy, int64_t ldy, const float* x, float* dw, float* db, int M, int N, int K,
                              int accumulate, void* ws, size_t ws_bytes, void* stream);
int egotap_train_transpose(const float* in, float* out, int R, int C, int64_t ldo, void* stream);
int egotap_train_add_inplace(float* out, const float* in, int64_t n, void* stream);
int egotap_train_patch_fwd(egotap_handle h, const float* hm, int B, const float* w, const float* b, const float* mask_tok,
                           const float* pos, float* x, void* stream);
int egotap_train_patch_split(egotap_handle h, const float* dpos, float* dbias, float* dmask, int accumulate, void* stream);
int egotap_train_tokens_scatter(egotap_handle h, const float* dA, float* dtok, int B, void* stream);
int egotap_train_layernorm_fwd(const float* x, float* y, const float* g, const float* b, float* mean, float* rstd, int rows,
                               float eps, void* stream);
int egotap_train_layernorm_bwd(const float* x, const float* dy, const float* g, const float* mean, const float* rstd,
                               const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int accumulate,
                               void* ws, size_t ws_bytes, void* stream);
int egotap_train_bn_lrelu_fwd(const float* z, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                              float* run_mean, float* run_var, int R, int C, float eps, float momentum, void* ws,
                              size_t ws_bytes, void* stream);
int egotap_train_bn_lrelu_bwd(const float* z, const float* y, const float* dy, const float* gamma, const float* mean,
                              const float* rstd, float* dz, float* dgamma, float* dbeta, int R, int C, int accumulate,
                              void* ws, size_t ws_bytes, void* stream);
int egotap_train_qkv_fwd(egotap_handle h, const float* y, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                         const float* bv, float* qkv, int M, int D, void* stream);
int egotap_train_attention_fwd(const float* qkv, float* ctx, float* lse, int B, int N, int heads, int precision, void* stream);
int egotap_train_attention_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* delta,
                               float* dqkv, int B, int N, int heads, int precision, void* stream);
int egotap_train_pu_saved_bytes(egotap_handle h, int B, size_t* bytes, size_t* hs1_offset);
int egotap_train_pu_fwd(egotap_handle h, const float* posz, const float* rotz, int B, void* saved, size_t saved_bytes, void* stream);
int egotap_train_pu_bwd_ws_bytes(egotap_handle h, int B, size_t* bytes);
int egotap_train_pu_bwd(egotap_handle h, const float* posz, const float* rotz, int B, const void* saved, const float* dhs1,
                        float* dposz, float* drotz, float* const* grads, int accumulate, void* ws, size_t ws_bytes, void* stream);
int egotap_train_pose_head_fwd(egotap_handle h, const float* posz, const float* hs1, int B, float* pose, void* stream);
int egotap_train_pose_head_bwd(egotap_handle h, const float* posz, const float* hs1, const float* dpose, int B, float* dposz,
                               float* dhs1, float* dWp, float* dbp, float* dWg, float* dbg, int accumulate, void* stream);
/* out[2] = (loss_pose, loss_cos_sim) as backward_AutoEncoder weighs them; dpred [2, B, J, 3]: plane 0 = d loss_pose / d pred,
 * plane 1 = d loss_cos_sim / d pred; partial [B, 2] scratch */
int egotap_train_pose_loss(egotap_handle h, const float* pred, const float* gt, float* dpred, float* out, float* partial,
                           int B, float lambda_mpjpe, float lambda_cos_sim, void* stream);
/* torch.optim.AdamW update of one tensor; hyper-parameters are doubles (python floats): the bias corrections 1 - beta^step are
 * computed in double on the host, as torch does */
int egotap_train_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                       double weight_decay, int step, void* stream);
/* the same update for every tensor of a network in ONE launch (the optimizer step of egotap_autoencoder_model.py:313-314): gradients
 * and the two moment buffers are flat arenas of `span` floats with one layout, table = device int64 [nseg][3] = {arena offset, numel,
 * parameter pointer} sorted by offset (segments may be separated by padding) */
int egotap_train_adamw_multi(const void* table, int nseg, const float* g, float* m, float* v, int64_t span, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int step, void* stream);

/* ---- the lifting head's training step as one call per direction (reference: the autograd graph behind
 * egotap_autoencoder_model.py:284-311 forward() + loss.backward()) -----------------------------------------------------------
 * egotap_lift_forward_train = egotap_lift_forward in train mode (BatchNorm1d on batch statistics, running statistics updated in
 * the bound buffers; num_batches_tracked is the caller's bookkeeping) keeping every activation the backward needs in `saved`;
 * egotap_lift_backward walks the layers in reverse and OVERWRITES the gradient buffers bound with egotap_bind_grad (same keys as
 * egotap_bind_param; every trained tensor must have one).  Both compose the granular operators above, in a fixed order, on the
 * caller's stream.  fp32 tensors; the large GEMMs follow egotap_set_precision (f32 / bf16x3 / bf16 operand copies).
 *   hm        device f32 [B, 6J, S, S]          pose    device f32 [B, out_joints, 3]        dpose  device f32 [B, out_joints, 3]
 *   saved     device, egotap_lift_train_bytes   ws      device scratch, egotap_lift_train_bytes (shared by both calls)
 *   bucket_events  NULL / n_events = 0, or vit_layers + 2 hipEvent_t: event k is recorded on `stream` when every gradient of
 *             bucket k is final -- bucket 0: pose head, propagation units, both FC encoders, final LayerNorm and the last ViT layer's
 *             output.dense.bias; bucket 1 + j: ViT layer L-1-j plus the output.dense.bias of the layer below; last: embeddings --
 *             so that a data-parallel caller starts each bucket's all-reduce behind its event while the backward goes on. */
int egotap_bind_grad(egotap_handle h, const char* key, void* dev_ptr, int64_t numel);
int egotap_lift_train_bytes(egotap_handle h, int B, size_t* saved_bytes, size_t* ws_bytes);
int egotap_lift_forward_train(egotap_handle h, const float* hm, int B, float* pose, void* saved, size_t saved_bytes, void* ws,
                              size_t ws_bytes, void* stream);
int egotap_lift_backward(egotap_handle h, const float* hm, const float* dpose, int B, const void* saved, size_t saved_bytes, void* ws,
                         size_t ws_bytes, void* const* bucket_events, int n_events, void* stream);

/* ---- heatmap-estimator training operators (fp32), called by the autograd glue (egotap_amd/hm_training.py) ----------------
 * One optimisation step of the stage-1 model (model/heatmap_shared_model.py:98-172): HeatMap_UnrealEgo_Shared in train mode
 * (model/net_architecture.py:25-173: BatchNorm2d on batch statistics), MSE / limb-length-normalised MSE losses, Adam.
 * All tensors NCHW fp32 with explicit image strides (floats), so concat slices are read and written in place.
 *   conv_fwd     convolution + bias (+ residual) (+ ReLU) on the forward kernels; also the INPUT gradient: dX = conv(dY, conv_wt(W)),
 *                stride-2 layers after zero_upsample(dY)
 *   conv_wgrad   dW[Cout][Cin][ks][ks] (+)= sum_{n,y,x} dY * shifted X  (ks 1 / 3 / 7, stride 1 / 2; split over images,
 *                fixed-order reduction: bitwise reproducible)
 *   bn2d_fwd     batch statistics over N*H*W, running-stat update (momentum, unbiased variance), y = [relu](bn(z) [+ res])
 *   bn2d_bwd     dz, dgamma, dbeta (and the residual branch's gradient dres = dy * [y > 0]) */
int egotap_hmtrain_conv_fwd(egotap_handle h, const float* x, const float* w, const float* bias, const float* res, float* y, int Nimg, int Cin,
                            int Cout, int wout, int taps, int stride, int relu, int64_t in_istride, int64_t out_istride, int64_t res_istride,
                            void* stream);
/* [r3] Eval-mode building blocks for the backbones egotap_hm_forward does not cover -- the Bottleneck ResNets behind --model_name resnet50 /
 * resnet101 (reference: model/net_architecture.py:61-64 torchvision resnet50 / resnet101, :108-111 feature_scale 4).  The host side composes
 * the forward from them (egotap_amd/networks.py): y = [relu](BatchNorm_eval(conv(x, w)) [+ res]) with the BatchNorm folded in the epilogue as
 * gamma / sqrt(var + 1e-5) (what egotap_hm_forward does for the BasicBlock nets), and the 7x7 / 2 stem with its BatchNorm + ReLU
 * (y [2B, 64, S0/2, S0/2], image n = 2b + eye).  fp32 only. */
int egotap_hm_conv_bn_fwd(egotap_handle h, const float* x, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                          const float* res, float* y, int Nimg, int Cin, int Cout, int wout, int taps, int stride, int relu, int64_t in_istride,
                          int64_t out_istride, int64_t res_istride, void* stream);
int egotap_hm_stem_bn_fwd(const float* left, const float* right, const float* w, const float* gamma, const float* beta, const float* mean,
                          const float* var, float* y, int B, int S0, void* stream);
/* bf16 precision modes: the 3x3 stride-1 convolutions (forward and input gradient) of the training step run on conv_bf16 once a
 * scratch buffer for their repacked weights is set (egotap_hmtrain_pack_bytes() bytes, caller-owned, 16-byte aligned) */
int egotap_hmtrain_set_pack_buffer(egotap_handle h, void* buf, size_t bytes);
size_t egotap_hmtrain_pack_bytes(void);
int egotap_hmtrain_stem_fwd(const float* left, const float* right, const float* w, float* z, int B, int S0, void* stream);
int egotap_hmtrain_bn2d_fwd(const float* z, float* y, const float* res, const float* gamma, const float* beta, float* mean, float* rstd,
                            float* run_mean, float* run_var, int N, int C, int HW, int64_t z_istride, int64_t y_istride, int64_t res_istride,
                            int relu, float eps, float momentum, void* ws, size_t ws_bytes, void* stream);
int egotap_hmtrain_bn2d_bwd(const float* z, const float* y, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dz,
                            float* dres, float* dgamma, float* dbeta, int N, int C, int HW, int64_t z_istride, int64_t dy_istride, int relu,
                            int accumulate, int dres_accumulate, void* ws, size_t ws_bytes, void* stream);
int egotap_hmtrain_chansum(const float* dy, float* out, int N, int C, int HW, int64_t istride, int accumulate, void* ws, size_t ws_bytes, void* stream);
int egotap_hmtrain_conv_wt(const float* w, float* wt, int Cout, int Cin, int taps, void* stream);
int egotap_hmtrain_zero_upsample(const float* in, float* out, int N, int C, int H, int64_t in_istride, int64_t out_istride, void* stream);
int egotap_hmtrain_conv_wgrad(const float* dy, const float* x, float* dw, int Nimg, int Cin, int Cout, int wout, int ks, int stride,
                              int64_t dy_istride, int64_t x_istride, int accumulate, int precision, void* ws, size_t ws_bytes, void* stream);
int egotap_hmtrain_relu_bwd(const float* y, const float* dy, float* dz, int N, int C, int HW, int64_t y_istride, int64_t dy_istride,
                            int64_t dz_istride, void* stream);
int egotap_hmtrain_maxpool_bwd(const float* x, const float* dy, float* dx, int64_t planes, int HIN, void* stream);
int egotap_hmtrain_upsample_bwd(const float* dy, float* dx, int N, int C, int HIN, int64_t dy_istride, int64_t dx_istride, void* stream);
int egotap_hmtrain_maxpool_fwd(const float* x, float* y, int64_t planes, int HIN, void* stream);
int egotap_hmtrain_upsample_fwd(const float* x, float* y, int N, int C, int HIN, int64_t in_istride, int64_t out_istride, void* stream);
int egotap_hmtrain_mse(const float* pred, const float* gt, const float* plen, float* dpred, float* loss, int B, int Cn, int HW, float lambda,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- bf16-storage operators (EGOTAP_PREC_BF16 with bf16 tensors in HBM; gemm_bf16s.h) -----------------------------------------
 * nn.Linear forward / input gradient on bf16 operands:  OUT = epi(x[M,K] w[N,K]^T), x row stride ldx, outputs row stride ldo
 * (elements).  N % 256 == 0, K % 32 == 0, 16-byte aligned pointers.  epi:
 *   0  out0 bf16 = acc (+ bias if not NULL)                    1  out0 f32 = acc + bias + aux (aux: f32 residual, may alias out0)
 *   2  out0 bf16 = z = acc + bias (may be NULL), out1 bf16 = GELU(z)      3  out0 bf16 = acc * GELU'(aux), aux: bf16 z;
 *   4  out0 f32 = acc + bias                                        with epi 3, out1 (optional) f32 [2 ceil(M / 256)][N]: per
 *      128-row block partial column sums of the stored out0 (finish with egotap_train_colsum over those rows: the bias gradient of
 *      the layer whose output gradient out0 is, without another pass over out0) */
int egotap_bf16_gemm_nt(const void* x, int64_t ldx, const void* w, const float* bias, int M, int N, int K, int epi, const void* aux,
                        void* out0, void* out1, int64_t ldo, void* stream);

/* nn.Linear weight gradient on bf16 operands: dw[N,K] (+)= dy[M,N]^T x[M,K] (fp32 result).  N, K % 256 == 0.  zeros: >= 512 bytes of
 * zeros (rows past M are fetched from it); ws: scratch for the split-M partial slabs (>= 4*N*K bytes per split, up to 256 splits are
 * used when it allows; fixed-order reduction: bitwise reproducible) */
int egotap_bf16_gemm_tn(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, int M, int N, int K, int accumulate,
                        const void* zeros, void* ws, size_t ws_bytes, void* stream);

/* LayerNorm(1024) with a bf16 output (the next GEMM's operand) and its backward: dy bf16, dx fp32 (+ dxb: a bf16 copy when not NULL),
 * dgamma / dbeta, and dcolsum (not NULL): column sums of dx = the bias gradient of the Linear layer whose output gradient dx is.
 * ws >= (3 * ceil(rows / 64) + 3 + 3 * ceil(ceil(rows / 64) / 64)) * 4096 bytes */
int egotap_bf16_layernorm_fwd(const float* x, void* y, const float* g, const float* b, float* mean, float* rstd, int rows, float eps, void* stream);
int egotap_bf16_layernorm_bwd(const float* x, const void* dy, const float* g, const float* mean, const float* rstd, const float* dres, float* dx,
                              void* dxb, float* dgamma, float* dbeta, float* dcolsum, int rows, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* out[N] (+)= column sums of a bf16 matrix y[M, N] (row stride ldy): bias gradients */
int egotap_bf16_colsum(const void* y, int64_t ldy, float* out, int M, int N, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* per-step weight preparation: w fp32 [N, K] (the live master weights) -> wb bf16 [N, K] and, when wt != NULL, wt bf16 [K, N] with row
 * stride ldt (so three projections can share one transposed [K, 3N] matrix) */
int egotap_bf16_prep_weight(const float* w, void* wb, void* wt, int N, int K, int64_t ldt, void* stream);
int egotap_bf16_from_f32(const float* src, void* dst, int64_t n, void* stream);
/* ViTSelfAttention core on bf16 tensors (modeling_vit.py:233-252): qkv bf16 [B*N, 3*heads*128] -> ctx bf16 [B*N, heads*128], lse fp32
 * [B*heads*N]; backward: dqkv bf16 (same layout as qkv), delta fp32 [B*heads*N] scratch.  Two backward kernels (dQ; dK + dV). */
int egotap_bf16_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int N, int heads, void* stream);
int egotap_bf16_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, int B, int N, int heads,
                              void* stream);
/* The same, and the q | k | v bias gradients (column sums of dqkv) as well: the kernels' epilogues leave per-block partial sums in ws
 * (needs (B N / 32)(3 heads 128) floats + 64 MB), three small fp32 column sums finish them -- no pass over dqkv.  Shapes without that
 * epilogue (N % 64 != 0) fall back to the column-sum pass. */
int egotap_bf16_attention_bwd_bias(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, float* dq_bias,
                                   float* dk_bias, float* dv_bias, int B, int N, int heads, void* ws, size_t ws_bytes, void* stream);
/* fc1 of the two heatmap encoders on bf16 operands (which 0: position encoder, src = final-LayerNorm tokens bf16 [B*seq, D];
 * 1: rotation encoder, src = the head's input heatmaps as bf16 [B, 6J, S, S]); z fp32 [B*T, 2048] = x w^T + bias.
 * wgrad: dw fp32 [2048, K1] = dz^T x;  dgrad_tokens (position encoder): dtok bf16 [B*seq, D] = scatter(dz wt^T), wt bf16 [K1, 2048] */
int egotap_bf16_fc1_fwd(egotap_handle h, int which, const void* src, const void* w, const float* bias, float* z, int B, void* stream);
/* Patch embedding of the position heatmaps on bf16 operands (ViTPatchEmbeddings + mask token + position embeddings over the tiled heatmap
 * image: net_architecture.py:326-336, modeling_vit.py:137-153): hmb = the head's input heatmaps as bf16 [B, 6J, S, S], w = bf16 copy of
 * projection.weight [D, 256], zeros >= 16 bytes of zeros (dummy cells); x fp32 [B*seq, D]. */
int egotap_bf16_patch_fwd(egotap_handle h, const void* hmb, const void* w, const float* bias, const float* mask_tok, const float* pos,
                          const void* zeros, float* x, int B, void* stream);
int egotap_bf16_fc1_wgrad(egotap_handle h, int which, const void* dz, const void* src, float* dw, int B, const void* zeros, void* ws, size_t ws_bytes,
                          void* stream);
int egotap_bf16_fc1_dgrad_tokens(egotap_handle h, const void* dz, const void* wt, void* dtok, int B, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* EGOTAP_H */
