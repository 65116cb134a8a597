// C ABI of libegotap_hip.so (include/egotap.h): handle, parameter binding, the lifting-head forward
// (EgoTAPAutoEncoder.forward, model/net_architecture.py:682-758) as a sequence of HIP launches on the
// caller's stream, single-operator entry points and the GEMM timing hook.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "attention_f32.h"
#include "common.h"
#include "conv_f32.h"
#include "conv_bf16.h"
#include "gemm_f32.h"
#include "gemm_bf16.h"
#include "gemm_bf16_dma.h"
#include "gemm_f32_dma.h"
#include "attention_bf16.h"
#include "layernorm.h"
#include "metrics.h"
#include "heatmap_synth.h"
#include "pu_chain.h"
#include "gemm_tn_bf16.h"
#include "train_ops.h"
#include "gemm_bf16s.h"
#include "gemm_bf16s64.h"
#include "gemm_tn_bf16s.h"
#include "bf16s_ops.h"
#include "attention_bf16s.h"
#include "attention_bf16s2.h"
#include "conv_bf16s.h"
#include "stem_bf16s.h"
#include "conv64_bf16s.h"
#include "bn_bf16s.h"

// The library is ONE source compiled as four translation units in parallel (egotap_amd/build.py: -DEGOTAP_PART=0 core and
// inference, 1 lifting-head training operators, 2 heatmap-estimator training operators, 3 bf16-storage operators); every exported function belongs to one
// part, the static helpers and kernel templates are visible to all.  Without EGOTAP_PART the file is a single translation unit.
#ifndef EGOTAP_PART
#define EGOTAP_PART -1
#endif
#define EGOTAP_IN(n) (EGOTAP_PART < 0 || EGOTAP_PART == (n))

// ------------------------------------------------------------------------------------------------ errors
#if EGOTAP_IN(0)
static thread_local char g_err[1024] = "";
void egotap_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#endif
#if EGOTAP_IN(0)
extern "C" const char* egotap_last_error(void) { return g_err; }
#endif
#if EGOTAP_IN(0)
extern "C" int egotap_abi_version(void) { return EGOTAP_ABI_VERSION; }
#endif

// ------------------------------------------------------------------------------------------------ tiles
using TileA = GemmCfg<128, 128, 32, 2, 2, 2>;   // 4 waves, 64x64 per wave, 2 blocks/CU
using TileB = GemmCfg<256, 128, 32, 4, 2, 2>;   // 8 waves, 64x64 per wave, 1 block/CU
using TileC = GemmCfg<128, 256, 32, 2, 4, 2>;   // 8 waves, 64x64 per wave, 1 block/CU
using TileD = GemmCfg<256, 256, 16, 4, 2, 2>;   // 8 waves, 64x128 per wave, 1 block/CU
using TileE = GemmCfg<256, 128, 16, 2, 2, 2>;   // 4 waves, 128x64 per wave, 2 blocks/CU
using TileF = GemmCfg<64, 64, 32, 2, 2, 2>;     // 4 waves, 32x32 per wave (small problems)
using TileG = GemmCfg<256, 128, 32, 2, 2, 1>;   // 4 waves, 128x64 per wave, 1 block/CU
using TileS = GemmCfg<64, 128, 32, 2, 4, 1>;    // [r4] 8 waves, 32x32 per wave (two waves per SIMD when the workgroup is alone on its CU: -1.8 % on a B = 1 forward against 4 waves): GEMMs of at most 64 rows (the encoders' fc layers and the PU projections at B <= 2-4) --
                                                // on the 128-row tile a 30-row product is paced by its MFMAs on 98 rows of zeros (fc1 at B = 1: 1.7 TB/s of weights)
using PipeA = PipeCfg<128, 128, 16, 2, 2, 2>;   // pipelined: 3 x 20 KiB slabs, 2 blocks/CU
using PipeB = PipeCfg<128, 128, 32, 2, 2, 1>;   // pipelined: 3 x 36 KiB slabs, 1 block/CU
using PipeC = PipeCfg<256, 128, 16, 4, 2, 2>;   // pipelined, 8 waves: 3 x 30 KiB slabs, 1 block/CU
using PipeD = PipeCfg<256, 256, 16, 4, 2, 2>;   // pipelined, 8 waves 64x128 per wave: 3 x 40 KiB slabs

#if EGOTAP_IN(0)
extern "C" const char* egotap_gemm_tile_name(int tile) {
    switch (tile) {
        case 0: case 1: return "128x128x32/4w";
        case 2: return "256x128x32/8w";
        case 3: return "128x256x32/8w";
        case 4: return "256x256x16/8w";
        case 5: return "256x128x16/4w";
        case 6: return "64x64x32/4w";
        case 7: return "256x128x32/4w";
        case 8: return "pipe128x128x16/4w";
        case 9: return "pipe128x128x32/4w";
        case 10: return "pipe256x128x16/8w";
        case 11: return "pipe256x256x16/8w";
        case 12: return "persist256x256x16/8w";
        case 13: return "persist256x256x16/8w/bf16x3";
        case 14: return "persist256x256x32/8w/bf16";
        case 15: return "persist256x256x16/8w/bf16x3/interleaved";
        case 16: return "persist256x256x32/8w/bf16/interleaved";
        case 19: return "dma256x256x16/8w (exact fp32, LDS-DMA staging)";
        default: return nullptr;
    }
}
#endif

// ------------------------------------------------------------------------------------------------ handle
struct Param {
    void* ptr = nullptr;
    int64_t numel = 0;
    int dtype = EGOTAP_F32;
};

struct LiftParams {   // resolved raw pointers of net_AutoEncoder
    const float *mask_tok, *pos_emb, *patch_w, *patch_b;
    struct Layer {
        const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b, *up_w, *up_b, *dn_w, *dn_b;
        const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    } layer[8];
    const float *lnf_g, *lnf_b;
    struct Fc { const float *w, *b, *g, *beta, *mean, *var; } pos_fc[3], rot_fc[3];
    const float *x2f0_w, *x2f0_b, *x2h0_w, *x2h0_b, *b2h0_w, *b2h0_b, *h2h0_w, *h2h0_b;
    const float *x2f1_w, *x2f1_b, *x2h1_w, *x2h1_b, *h2h1_w, *h2h1_b;
    const float *pose_w, *pose_b, *glob_w, *glob_b;
};

struct HmParams {   // resolved raw pointers of one heatmap estimator (net_HeatMap / net_RotHeatMap)
    struct Bn { const float *g, *b, *m, *v; long long* nbt; };      // nbt: num_batches_tracked where bound (EGOTAP_I64), else null; m / v are written by the batch-statistics forward
    const float* stem_w; Bn stem_bn;
    struct Block { const float *w1, *w2, *wd; Bn bn1, bn2, bnd; } blk[4][6];
    int nblk[4];        // BasicBlocks per stage (resnet18: 2,2,2,2; resnet34: 3,4,6,3)
    struct Cv { const float *w, *b; } l1x1[4], up[3], head;   // layerK_1x1 (K=1..4), conv_up1..3, conv_heatmap
    int n_out;   // output channels of conv_heatmap (2 * heatmaps per eye)
};

struct egotap_handle_s {
    egotap_config cfg;
    // derived (spec.py LiftPreset)
    int J, T, grid, ppd, side, seq, C, D, H, hid, out_joints;
    std::unordered_map<std::string, int64_t> expect[EGOTAP_NET_COUNT];   // key -> numel (forward-needed keys)
    std::unordered_map<std::string, Param> bound[EGOTAP_NET_COUNT];
    bool lift_resolved = false;
    LiftParams lp;
    std::unordered_map<std::string, Param> bound_grad;                   // egotap_bind_grad: where egotap_lift_backward writes
    bool grad_resolved = false;
    LiftParams lg;                                                       // the same tensors' gradient buffers (running stats unused)
    bool hm_resolved[EGOTAP_NET_COUNT] = {false, false, false};
    HmParams hp[EGOTAP_NET_COUNT];
    int debug_stop = 0;
    int pu_resident[2] = {-1, -1};     // workgroups of pu_chain_kernel<1> / <2> the device keeps resident (-1: not asked yet)
    bool pu_chain = true;              // egotap_set_pu_chain: one-launch recurrence (needs the device to itself) or per-step kernels
    unsigned* pu_fault_host = nullptr; // pinned, device-mapped word: set by pu_solo_kernel when a chain launch had to be redone (pu_chain.h)
    unsigned* pu_fault_dev = nullptr;  // the same word as the device sees it
    int pu_faults = 0;                 // chain launches found faulted so far (the chain is switched off for the handle at the first)
    int pu_debug_drop = 0;             // egotap_debug_pu_drop_workgroups
    int precision = EGOTAP_PREC_F32;   // arithmetic of the large GEMMs (egotap_set_precision)
    __bf16* wscratch = nullptr;        // scratch for the bf16 copy of a GEMM's weight matrix (plain-bf16 mode), caller-owned
    size_t wscratch_bytes = 0;
    __bf16* ascratch = nullptr;        // scratch for the bf16 copy of a GEMM's activation operand (plain-bf16 mode, gemm_bf16_dma_kernel), caller-owned
    size_t ascratch_bytes = 0;
    __bf16* conv_pack = nullptr;       // scratch for repacked conv weights (set per egotap_hm_forward call from the workspace)
    float* conv_part = nullptr;        // [r4] fp32 egotap_hm_forward only (null outside it): scratch for the input-channel-split partials of conv_f32.h
    size_t conv_part_floats = 0;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;   // start/stop pairs
    struct Rec { const char* role; const char* kernel; double flops; };
    std::vector<Rec> rec;         // one per recorded pair
    size_t ev_used = 0;
    std::string detail;           // JSON written by egotap_timing_read
};
typedef egotap_handle_s Handle;

static int isqrt_floor(int v) {
    int r = 0;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

static void lift_expect(Handle* h) {
    auto& e = h->expect[EGOTAP_NET_LIFT];
    const int64_t D = h->D, H = h->H, hid = h->hid, x = 2 * hid;
    const std::string v = "pos_heatmap_encoder.vit.";
    e[v + "embeddings.mask_token"] = D;
    e[v + "embeddings.position_embeddings"] = (int64_t)h->seq * D;
    e[v + "embeddings.patch_embeddings.projection.weight"] = D * h->cfg.patch * h->cfg.patch;
    e[v + "embeddings.patch_embeddings.projection.bias"] = D;
    auto lin = [&](const std::string& p, int64_t n_out, int64_t n_in) {
        e[p + ".weight"] = n_out * n_in;
        e[p + ".bias"] = n_out;
    };
    for (int i = 0; i < h->cfg.vit_layers; ++i) {
        const std::string l = v + "encoder.layer." + std::to_string(i) + ".";
        lin(l + "attention.attention.query", D, D);
        lin(l + "attention.attention.key", D, D);
        lin(l + "attention.attention.value", D, D);
        lin(l + "attention.output.dense", D, D);
        lin(l + "intermediate.dense", 4 * D, D);
        lin(l + "output.dense", D, 4 * D);
        e[l + "layernorm_before.weight"] = D; e[l + "layernorm_before.bias"] = D;
        e[l + "layernorm_after.weight"] = D;  e[l + "layernorm_after.bias"] = D;
    }
    e[v + "layernorm.weight"] = D; e[v + "layernorm.bias"] = D;
    auto fcb = [&](const std::string& p, int64_t n_in, int64_t n_out) {
        lin(p + ".fc", n_out, n_in);
        e[p + ".bn.weight"] = n_out; e[p + ".bn.bias"] = n_out;
        e[p + ".bn.running_mean"] = n_out; e[p + ".bn.running_var"] = n_out;
    };
    const int64_t k_pos = (int64_t)h->ppd * h->ppd * D, k_rot = 2LL * h->cfg.hm_size * h->cfg.hm_size;
    fcb("pos_heatmap_encoder.fc1", k_pos, 2048); fcb("pos_heatmap_encoder.fc2", 2048, 512); fcb("pos_heatmap_encoder.fc3", 512, hid);
    fcb("rot_heatmap_encoder.fc1", k_rot, 2048); fcb("rot_heatmap_encoder.fc2", 2048, 512); fcb("rot_heatmap_encoder.fc3", 512, hid);
    const std::string c = "skel_sequential_layer.lstm_custom.layers.";
    lin(c + "0.x2f", H + x, x); lin(c + "0.x2h", 4 * H, x); lin(c + "0.b2h", 4 * H, x); lin(c + "0.h2h", 4 * H, H);
    lin(c + "1.x2f", H, H); lin(c + "1.x2h", 4 * H, H); lin(c + "1.h2h", 4 * H, H);
    lin("pose_mlp.pose_fcs.0", 3, x + H);
    if (h->cfg.estimate_head) lin("global_mlp.pose_fcs.0", 6, (int64_t)h->J * H);
}

static const int HM_CH[4] = {64, 128, 256, 512};
// timing-hook role names of the backbone convolutions (static strings: the hook keeps the pointers)
#define HM_ROLE_ROW(L, S) {"hm.l" #L ".0." S, "hm.l" #L ".1." S, "hm.l" #L ".2." S, "hm.l" #L ".3." S, "hm.l" #L ".4." S, "hm.l" #L ".5." S}
static const char* const HM_R1[4][6] = {HM_ROLE_ROW(1, "conv1"), HM_ROLE_ROW(2, "conv1"), HM_ROLE_ROW(3, "conv1"), HM_ROLE_ROW(4, "conv1")};
static const char* const HM_R2[4][6] = {HM_ROLE_ROW(1, "conv2"), HM_ROLE_ROW(2, "conv2"), HM_ROLE_ROW(3, "conv2"), HM_ROLE_ROW(4, "conv2")};
static const char* const HM_RD[4] = {"", "hm.l2.down", "hm.l3.down", "hm.l4.down"};
static inline int hm_nblk(const Handle* h, int i) { return h->cfg.hm_blocks[i] > 0 ? h->cfg.hm_blocks[i] : 2; }

static void hm_expect(Handle* h, int net, int n_out) {
    auto& e = h->expect[net];
    const std::string bb = "backbone.backbone.backbone.";
    auto bn = [&](const std::string& p, int64_t c) {
        e[p + ".weight"] = c; e[p + ".bias"] = c; e[p + ".running_mean"] = c; e[p + ".running_var"] = c;
    };
    e[bb + "conv1.weight"] = 64 * 3 * 49;
    bn(bb + "bn1", 64);
    int cin = 64;
    for (int i = 0; i < 4; ++i) {
        const int c = HM_CH[i];
        for (int b = 0; b < hm_nblk(h, i); ++b) {
            const std::string p = bb + "layer" + std::to_string(i + 1) + "." + std::to_string(b);
            const int bc = b == 0 ? cin : c;
            e[p + ".conv1.weight"] = (int64_t)c * bc * 9; bn(p + ".bn1", c);
            e[p + ".conv2.weight"] = (int64_t)c * c * 9;  bn(p + ".bn2", c);
            if (b == 0 && i > 0) { e[p + ".downsample.0.weight"] = (int64_t)c * bc; bn(p + ".downsample.1", c); }
        }
        cin = c;
    }
    const std::string a = "after_backbone.";
    auto cv = [&](const std::string& n, int64_t co, int64_t ci, int k) { e[a + n + ".weight"] = co * ci * k * k; e[a + n + ".bias"] = co; };
    cv("layer1_1x1.0", 128, 128, 1); cv("layer2_1x1.0", 256, 256, 1); cv("layer3_1x1.0", 516, 512, 1); cv("layer4_1x1.0", 1024, 1024, 1);
    cv("conv_up3.0", 1024, 1540, 3); cv("conv_up2.0", 512, 1280, 3); cv("conv_up1.0", 512, 640, 3);
    cv("conv_heatmap", n_out, 512, 1);
}

#if EGOTAP_IN(0)
extern "C" int egotap_create(const egotap_config* cfg, egotap_handle* out) {
    EGO_CHECK(cfg && out, "egotap_create: null argument");
    EGO_CHECK(cfg->struct_bytes == (int32_t)sizeof(egotap_config), "egotap_create: egotap_config is %d bytes, library expects %d",
              cfg->struct_bytes, (int)sizeof(egotap_config));
    EGO_CHECK(cfg->n_joints_hm >= 1 && cfg->n_joints_hm <= 64, "n_joints_hm out of range");
    EGO_CHECK(cfg->patch == 16, "patch size must be 16 (net_architecture.py:326)");
    EGO_CHECK(cfg->hm_size > 0 && cfg->hm_size % 16 == 0, "hm_size must be a positive multiple of 16 (net_architecture.py:327)");
    EGO_CHECK(cfg->vit_dim == 1024 && cfg->vit_heads == 8, "ViT hidden size / heads are fixed at 1024 / 8 (net_architecture.py:340-348)");
    EGO_CHECK(cfg->vit_layers >= 1 && cfg->vit_layers <= 8, "vit_layers out of range");
    EGO_CHECK(cfg->pu_hidden == 512 && cfg->hidden == 128, "hidden sizes are fixed at ae_hidden_size 128 / PU 512 in this build");
    for (int i = 0; i < 4; ++i) EGO_CHECK(cfg->hm_blocks[i] >= 0 && cfg->hm_blocks[i] <= 6, "hm_blocks: at most 6 BasicBlocks per ResNet stage (resnet18 / resnet34)");
    Handle* h = new Handle();
    h->cfg = *cfg;
    h->J = cfg->n_joints_hm;
    h->T = 2 * h->J;
    h->grid = isqrt_floor(h->T - 1) + 1;
    h->ppd = cfg->hm_size / cfg->patch;
    h->side = h->grid * h->ppd;
    h->seq = h->side * h->side;
    h->C = 6 * h->J;
    h->D = cfg->vit_dim;
    h->H = cfg->pu_hidden;
    h->hid = cfg->hidden;
    h->out_joints = h->J + (cfg->estimate_head ? 1 : 0);
    lift_expect(h);
    hm_expect(h, EGOTAP_NET_HM_POS, 2 * h->J);        // num_heatmap per eye, stereo
    hm_expect(h, EGOTAP_NET_HM_ROT, 4 * h->J);        // cos + sin per limb per eye
    h->hp[EGOTAP_NET_HM_POS].n_out = 2 * h->J;
    h->hp[EGOTAP_NET_HM_ROT].n_out = 4 * h->J;
    *out = h;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" void egotap_destroy(egotap_handle h) {
    if (!h) return;
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->pu_fault_host) (void)hipHostFree(h->pu_fault_host);
    delete h;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_bind_param(egotap_handle h, int net, const char* key, void* dev_ptr, int64_t numel, int dtype) {
    EGO_CHECK(h && key, "egotap_bind_param: null argument");
    EGO_CHECK(net >= 0 && net < EGOTAP_NET_COUNT, "egotap_bind_param: bad net id %d", net);
    EGO_CHECK(dev_ptr != nullptr || numel == 0, "egotap_bind_param(%s): null device pointer", key);
    auto it = h->expect[net].find(key);
    if (it != h->expect[net].end()) {
        EGO_CHECK(dtype == EGOTAP_F32, "egotap_bind_param(%s): expected f32", key);
        EGO_CHECK(it->second == numel, "egotap_bind_param(%s): %lld elements, expected %lld", key, (long long)numel,
                  (long long)it->second);
        EGO_CHECK(((uintptr_t)dev_ptr & 15) == 0, "egotap_bind_param(%s): pointer must be 16-byte aligned", key);
    }
    Param p;
    p.ptr = dev_ptr; p.numel = numel; p.dtype = dtype;
    h->bound[net][key] = p;   // keys the forward never reads (pooler, cls_token, num_batches_tracked) are kept but unused
    if (net == EGOTAP_NET_LIFT) h->lift_resolved = false;
    else h->hm_resolved[net] = false;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_unbound_count(egotap_handle h, int net, int* count) {
    EGO_CHECK(h && count, "egotap_unbound_count: null argument");
    EGO_CHECK(net >= 0 && net < EGOTAP_NET_COUNT, "bad net id %d", net);
    int n = 0;
    for (auto& kv : h->expect[net])
        if (!h->bound[net].count(kv.first)) ++n;
    *count = n;
    return EGOTAP_OK;
}
#endif

typedef std::unordered_map<std::string, Param> ParamMap;
static const float* P(Handle*, const ParamMap& m, const std::string& key, bool& ok) {
    auto it = m.find(key);
    if (it == m.end()) {
        if (ok) egotap_set_error("'%s' is not bound", key.c_str());
        ok = false;
        return nullptr;
    }
    return (const float*)it->second.ptr;
}
static const float* P(Handle* h, int net, const std::string& key, bool& ok) { return P(h, h->bound[net], key, ok); }

// fills a LiftParams from a key -> pointer map: the parameters (with_stats: and the BatchNorm running statistics) or their gradients
static int lift_resolve_into(Handle* h, const ParamMap& N, LiftParams& p, bool with_stats) {
    bool ok = true;
    const std::string v = "pos_heatmap_encoder.vit.";
    p.mask_tok = P(h, N, v + "embeddings.mask_token", ok);
    p.pos_emb = P(h, N, v + "embeddings.position_embeddings", ok);
    p.patch_w = P(h, N, v + "embeddings.patch_embeddings.projection.weight", ok);
    p.patch_b = P(h, N, v + "embeddings.patch_embeddings.projection.bias", ok);
    for (int i = 0; i < h->cfg.vit_layers; ++i) {
        const std::string l = v + "encoder.layer." + std::to_string(i) + ".";
        auto& L = p.layer[i];
        L.q_w = P(h, N, l + "attention.attention.query.weight", ok); L.q_b = P(h, N, l + "attention.attention.query.bias", ok);
        L.k_w = P(h, N, l + "attention.attention.key.weight", ok);   L.k_b = P(h, N, l + "attention.attention.key.bias", ok);
        L.v_w = P(h, N, l + "attention.attention.value.weight", ok); L.v_b = P(h, N, l + "attention.attention.value.bias", ok);
        L.o_w = P(h, N, l + "attention.output.dense.weight", ok);    L.o_b = P(h, N, l + "attention.output.dense.bias", ok);
        L.up_w = P(h, N, l + "intermediate.dense.weight", ok);       L.up_b = P(h, N, l + "intermediate.dense.bias", ok);
        L.dn_w = P(h, N, l + "output.dense.weight", ok);             L.dn_b = P(h, N, l + "output.dense.bias", ok);
        L.ln1_g = P(h, N, l + "layernorm_before.weight", ok);        L.ln1_b = P(h, N, l + "layernorm_before.bias", ok);
        L.ln2_g = P(h, N, l + "layernorm_after.weight", ok);         L.ln2_b = P(h, N, l + "layernorm_after.bias", ok);
    }
    p.lnf_g = P(h, N, v + "layernorm.weight", ok);
    p.lnf_b = P(h, N, v + "layernorm.bias", ok);
    const char* encs[2] = {"pos_heatmap_encoder", "rot_heatmap_encoder"};
    for (int e = 0; e < 2; ++e)
        for (int i = 0; i < 3; ++i) {
            const std::string f = std::string(encs[e]) + ".fc" + std::to_string(i + 1);
            LiftParams::Fc& fc = e == 0 ? p.pos_fc[i] : p.rot_fc[i];
            fc.w = P(h, N, f + ".fc.weight", ok);        fc.b = P(h, N, f + ".fc.bias", ok);
            fc.g = P(h, N, f + ".bn.weight", ok);        fc.beta = P(h, N, f + ".bn.bias", ok);
            fc.mean = fc.var = nullptr;
            if (with_stats) { fc.mean = P(h, N, f + ".bn.running_mean", ok); fc.var = P(h, N, f + ".bn.running_var", ok); }
        }
    const std::string c = "skel_sequential_layer.lstm_custom.layers.";
    p.x2f0_w = P(h, N, c + "0.x2f.weight", ok); p.x2f0_b = P(h, N, c + "0.x2f.bias", ok);
    p.x2h0_w = P(h, N, c + "0.x2h.weight", ok); p.x2h0_b = P(h, N, c + "0.x2h.bias", ok);
    p.b2h0_w = P(h, N, c + "0.b2h.weight", ok); p.b2h0_b = P(h, N, c + "0.b2h.bias", ok);
    p.h2h0_w = P(h, N, c + "0.h2h.weight", ok); p.h2h0_b = P(h, N, c + "0.h2h.bias", ok);
    p.x2f1_w = P(h, N, c + "1.x2f.weight", ok); p.x2f1_b = P(h, N, c + "1.x2f.bias", ok);
    p.x2h1_w = P(h, N, c + "1.x2h.weight", ok); p.x2h1_b = P(h, N, c + "1.x2h.bias", ok);
    p.h2h1_w = P(h, N, c + "1.h2h.weight", ok); p.h2h1_b = P(h, N, c + "1.h2h.bias", ok);
    p.pose_w = P(h, N, "pose_mlp.pose_fcs.0.weight", ok); p.pose_b = P(h, N, "pose_mlp.pose_fcs.0.bias", ok);
    p.glob_w = p.glob_b = nullptr;
    if (h->cfg.estimate_head) {
        p.glob_w = P(h, N, "global_mlp.pose_fcs.0.weight", ok);
        p.glob_b = P(h, N, "global_mlp.pose_fcs.0.bias", ok);
    }
    return ok ? EGOTAP_OK : EGOTAP_ERR_UNBOUND;
}

static int lift_resolve(Handle* h) {
    if (h->lift_resolved) return EGOTAP_OK;
    const int rc = lift_resolve_into(h, h->bound[EGOTAP_NET_LIFT], h->lp, true);
    if (rc != EGOTAP_OK) return rc;
    h->lift_resolved = true;
    return EGOTAP_OK;
}

// ------------------------------------------------------------------------------------------------ timing
#if EGOTAP_IN(0)
extern "C" int egotap_timing_enable(egotap_handle h, int enable) {
    EGO_CHECK(h, "null handle");
    h->timing = enable != 0;
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(0)
extern "C" int egotap_timing_read(egotap_handle h, int* launches, double* total_ms, double* total_flops) {
    EGO_CHECK(h && launches && total_ms && total_flops, "null argument");
    struct Agg { std::string role, kernel; int n = 0; double ms = 0, flops = 0; };
    std::vector<Agg> aggs;
    double ms = 0.0, fl = 0.0;
    const size_t pairs = h->ev_used / 2;
    for (size_t i = 0; i < pairs; ++i) {
        EGO_HIP(hipEventSynchronize(h->ev[2 * i + 1]));
        float t = 0.f;
        EGO_HIP(hipEventElapsedTime(&t, h->ev[2 * i], h->ev[2 * i + 1]));
        ms += t;
        fl += h->rec[i].flops;
        Agg* a = nullptr;
        for (auto& x : aggs)
            if (x.role == h->rec[i].role) { a = &x; break; }
        if (!a) { aggs.emplace_back(); a = &aggs.back(); a->role = h->rec[i].role; a->kernel = h->rec[i].kernel; }
        a->n += 1; a->ms += t; a->flops += h->rec[i].flops;
    }
    std::string js = "[";
    for (size_t i = 0; i < aggs.size(); ++i) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s{\"role\": \"%s\", \"kernel\": \"%s\", \"launches\": %d, \"ms\": %.6f, \"flops\": %.6e}",
                 i ? ", " : "", aggs[i].role.c_str(), aggs[i].kernel.c_str(), aggs[i].n, aggs[i].ms, aggs[i].flops);
        js += buf;
    }
    js += "]";
    h->detail = js;
    *launches = (int)pairs;
    *total_ms = ms;
    *total_flops = fl;
    h->ev_used = 0;
    h->rec.clear();
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(0)
extern "C" const char* egotap_timing_detail(egotap_handle h) { return h ? h->detail.c_str() : ""; }
#endif

struct GemmTimer {   // brackets one GEMM launch when the handle's timing hook is on
    Handle* h;
    hipStream_t s;
    bool on;
    GemmTimer(Handle* h_, hipStream_t s_, const char* role, const char* kernel, double flops)
        : h(h_), s(s_), on(h_ && h_->timing) {
        if (!on) return;
        if (h->ev_used + 2 > h->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            h->ev.push_back(a);
            h->ev.push_back(b);
        }
        (void)hipEventRecord(h->ev[h->ev_used], s);
        h->rec.push_back({role, kernel, flops});
    }
    ~GemmTimer() {
        if (!on) return;
        (void)hipEventRecord(h->ev[h->ev_used + 1], s);
        h->ev_used += 2;
    }
};

template <class AL> struct AlName;
template <> struct AlName<ALoadPlain> { static constexpr const char* v = "ALoadPlain"; };
template <> struct AlName<ALoadPatch> { static constexpr const char* v = "ALoadPatch"; };
template <> struct AlName<ALoadTokens> { static constexpr const char* v = "ALoadTokens"; };
template <> struct AlName<ALoadRot> { static constexpr const char* v = "ALoadRot"; };
template <> struct AlName<ALoadStereo> { static constexpr const char* v = "ALoadStereo"; };
template <> struct AlName<ALoadStereoGated> { static constexpr const char* v = "ALoadStereoGated"; };
template <class E> struct EpiName;
template <> struct EpiName<EpiBias> { static constexpr const char* v = "EpiBias"; };
template <> struct EpiName<EpiBiasRes> { static constexpr const char* v = "EpiBiasRes"; };
template <> struct EpiName<EpiBiasGelu> { static constexpr const char* v = "EpiBiasGelu"; };
template <> struct EpiName<EpiBnLrelu> { static constexpr const char* v = "EpiBnLrelu"; };
template <> struct EpiName<EpiPatch> { static constexpr const char* v = "EpiPatch"; };

template <class Cfg, class AL, class Epi>
static hipError_t gemm(Handle* h, const char* role, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc,
                       int M, int N, int K, hipStream_t s) {
    static const std::string kname = std::string("gemm_f32_kernel<") + std::to_string(Cfg::BM) + "x" +
                                     std::to_string(Cfg::BN) + "x" + std::to_string(Cfg::BK) + "," + AlName<AL>::v + "," +
                                     EpiName<Epi>::v + ">";
    GemmTimer t(h, s, role, kname.c_str(), 2.0 * M * N * K);
    return gemm_f32_launch<Cfg, AL, Epi>(al, W, epi, C, ldc, M, N, K, s);
}

// large GEMMs (ViT projections, fc1): 256x256 tile, 3-stage pipelined kernel -- fewest global-load
// instructions per MFMA (each costs the matrix pipe ~56 cycles, tools/mfma_probe.hip), DESIGN.md section 4
static int device_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// stream-ordered zero fill as a KERNEL: a hipMemsetAsync captured into a HIP graph did not re-execute on replay (ROCm 7.2: the
// propagation units' cell state then carried over from one replay to the next), a kernel node does
static __global__ __launch_bounds__(256) void zero_fill_kernel(f32x4* __restrict__ p, long n16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
static hipError_t zero_fill(void* p, size_t bytes, hipStream_t s) {       // p 16-byte aligned, bytes a multiple of 16
    if (bytes == 0) return hipSuccess;
    if ((((uintptr_t)p) | bytes) & 15) return hipErrorInvalidValue;
    const long n16 = (long)(bytes / 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (f32x4*)p, n16);
    return hipGetLastError();
}

// out[0:n | n:2n | 2n:3n] = a | b | c  (the fused q|k|v bias of the bf16-storage forward)
static __global__ __launch_bounds__(256) void concat3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                             float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = a[i]; out[n + i] = b[i]; out[2 * n + i] = c[i]; }
}

// rows of an fp32 matrix (row stride lda) -> dense bf16 [M, K]
static __global__ __launch_bounds__(256) void f32_to_bf16_rows_kernel(const float* __restrict__ src, long lda, __bf16* __restrict__ dst, int k8, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const long row = i / k8;
    const float* p = src + row * lda + (i - row * k8) * 8;
    *(bf16x8*)(dst + i * 8) = bf16_round8(*(const f32x4*)p, *(const f32x4*)(p + 4));
}
// plain-bf16 GEMM: W rounded to bf16 into the handle's scratch right before the launch (stream ordered), so the W operand costs
// half the vector-memory bytes; without a scratch (or if it is too small) the kernel converts fp32 weights on the fly.
// A plain row-major activation operand is rounded to bf16 the same way (egotap_set_act_scratch; skipped when the scratch is
// absent or too small) and the product runs on the
// LDS-DMA kernel (gemm_bf16_dma.h): same rounding of both operands, same fp32 accumulation order per output element.
template <class AL, class Epi>
static hipError_t gemm_bf16_plain(Handle* h, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N, int K, hipStream_t s) {
    const int nseg = N / W.seg;
    if (h && h->wscratch && (size_t)N * K * 2 <= h->wscratch_bytes && K % 8 == 0 && W.ld == K && nseg >= 1 && nseg <= 3 && nseg * W.seg == N) {
        SegMatB B;
        for (int i = 0; i < 3; ++i) B.p[i] = h->wscratch + (size_t)(i < nseg ? i : 0) * W.seg * K;
        B.seg = W.seg; B.ld = K;
        const long n8 = (long)W.seg * K / 8;
        for (int i = 0; i < nseg; ++i)
            hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, W.p[i], h->wscratch + (size_t)i * W.seg * K, n8);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if constexpr (std::is_same<AL, ALoadPlain>::value) {
            if (K % DmaCfg::BK == 0 && N % DmaCfg::BN == 0 && W.seg % DmaCfg::BN == 0 && al.lda % 4 == 0 && M >= 1024) {
                __bf16* a = (size_t)M * K * 2 <= h->ascratch_bytes ? h->ascratch : nullptr;
                if (a) {
                    const long a8 = (long)M * K / 8;
                    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((unsigned)((a8 + 255) / 256)), dim3(256), 0, s, al.A, al.lda, a, K / 8, a8);
                    e = hipGetLastError();
                    if (e != hipSuccess) return e;
                    return gemm_bf16_dma_launch(a, (long)K, B, epi, C, ldc, M, N, K, device_cu_count(), s);
                }
            }
        }
        return gemm_bf16_persist_launch<BfCfg<1, 1>, AL, Epi, SegMatB>(al, B, epi, C, ldc, M, N, K, device_cu_count(), s);
    }
    return gemm_bf16_persist_launch<BfCfg<1, 1>, AL, Epi>(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
}

template <class AL, class Epi>
static hipError_t gemm_big(Handle* h, const char* role, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc,
                           int M, int N, int K, hipStream_t s) {
    using Cfg = PipeD;
    if (h && h->precision == EGOTAP_PREC_BF16X3) {
        static const std::string kname3 = std::string("gemm_bf16_persist_kernel<256x256x16,bf16x3,") + AlName<AL>::v + "," + EpiName<Epi>::v + ">";
        GemmTimer t(h, s, role, kname3.c_str(), 2.0 * M * N * K);
        return gemm_bf16_persist_launch<BfCfg<3, 1>, AL, Epi>(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
    }
    if (h && h->precision == EGOTAP_PREC_BF16) {
        static const std::string kname1 = std::string("gemm_bf16_persist_kernel<256x256x32,bf16,") + AlName<AL>::v + "," + EpiName<Epi>::v + ">";
        GemmTimer t(h, s, role, kname1.c_str(), 2.0 * M * N * K);
        return gemm_bf16_plain(h, al, W, epi, C, ldc, M, N, K, s);
    }
    if constexpr (AL::HAS_PTR) {
        // loaders that are pure address math: same tile, same k order (bit-identical), slabs staged global -> LDS by DMA (gemm_f32_dma.h)
        if (N % DmaF32Cfg::BN == 0 && K % DmaF32Cfg::BK == 0 && W.seg % DmaF32Cfg::BN == 0 && W.ld % 4 == 0 && al.dma_ok()) {
            static const std::string kdma = std::string("gemm_f32_dma_kernel<256x256x16,") + AlName<AL>::v + "," + EpiName<Epi>::v + ">";
            GemmTimer t(h, s, role, kdma.c_str(), 2.0 * M * N * K);
            return gemm_f32_dma_launch(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
        }
    }
    static const std::string kname = std::string("gemm_f32_persist_kernel<256x256x16,") + AlName<AL>::v + "," + EpiName<Epi>::v + ">";
    GemmTimer t(h, s, role, kname.c_str(), 2.0 * M * N * K);
    return gemm_f32_persist_launch<Cfg, AL, Epi>(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
}

// Small batches (serving): the 256x256 persistent tiling leaves most CUs idle (B = 1: 12 tiles for attn_out / mlp_down, one
// tile row for fc1).  Below `fill` tiles the GEMM runs as exact-fp32 128x128 tiles with K split over blockIdx.y
// (gemm_f32_splitk_launch) in every precision mode; at and above it gemm_big as before, so large-batch results do not change.
static constexpr size_t SPLITK_FLOATS = (size_t)1 << 24;
// ONE routing decision for an fp32-tensor NT product, used by gemm_small (which launches it) and by gemm_res_ln (which fuses the reduce of a split
// product with the LayerNorm behind it and therefore must see exactly the split layout gemm_small would produce).
// The fitted constants below (0.2319 us per 16-deep K step of a 256 x 256 tile = 0.92 x 157.3 TF over 256 CUs, 5 us per round; the per-slab times inside
// gemm_f32_splitk_plan / gemm_f32_direct_estimate_us) were measured on the 256-CU MI355X; `cus` only scales the round counts.  On another part they
// would still pick a CORRECT kernel, not necessarily the faster one.
enum { GS_BIG = 0, GS_DIRECT_A, GS_DIRECT_S, GS_SPLIT_S, GS_SPLIT_A, GS_HYBRID };
struct SmallRoute { int kind, splits, head_rows; };
// epilogues that do not index their operands by the output ROW: a product may be cut into row ranges without touching the functor (GS_HYBRID)
template <class E> struct epi_row_free : std::false_type {};
template <> struct epi_row_free<EpiBias> : std::true_type {};
template <> struct epi_row_free<EpiBiasGelu> : std::true_type {};
template <> struct epi_row_free<EpiBnLrelu> : std::true_type {};
template <> struct epi_row_free<EpiNone> : std::true_type {};
static SmallRoute gemm_small_route(const Handle* h, bool plain_loader, int M, int N, int K, int wseg, int cus, bool row_free = false) {
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long fill = (h && h->precision != EGOTAP_PREC_F32) ? 64 : 160;
    if (N % 128 != 0 || K % 32 != 0 || wseg % 128 != 0) return SmallRoute{GS_BIG, 1, 0};
    if (tiles256 >= fill) {
        // [r4] fp32, plain operands, fewer than four rounds of 256 x 256 tiles: the 256 x 256 kernel runs WHOLE rounds at 0.92 of the matrix
        // pipe (B = 8, N = 4096: 288 tiles = two rounds for 1.1 rounds of work, 484 us) -- the 128 x 128 kernel with its epilogue inside takes
        // 415 us there.  Both estimates are fitted to tools/gemm_small_vs_big_probe.py and name the faster kernel at 46 of its 48 points.
        if (plain_loader && h && h->precision == EGOTAP_PREC_F32 && tiles256 < 4L * cus) {
            const double t_round = K * 0.2319 + 5.0;
            const double t_big = (double)((tiles256 + cus - 1) / cus) * t_round;
            const double t_direct = gemm_f32_direct_estimate_us<TileA>(M, N, K, cus);
            // [r5] GS_HYBRID: the tile rows that fill WHOLE rounds of the 256 x 256 kernel run there, the rows past them as a product of their own
            // (a few tiles: K split over the idle CUs) -- B = 8, N = 4096: one round of 256 tiles + 32 tiles split two ways, ~290 us for the 375 the
            // 128 x 128 kernel took.  Only for epilogues that do not index by row (the functor is passed on unchanged).
            if (row_free && N % 256 == 0 && tiles256 > cus) {
                const int tn = N / 256, mh = (int)((tiles256 / cus) * cus / tn);      // whole tile rows inside the whole rounds
                const int Mt = M - mh * 256;
                if (mh >= 1 && Mt > 0) {
                    const long tt = (long)((Mt + 255) / 256) * tn;
                    const double t_head = (double)(((long)mh * tn + cus - 1) / cus) * t_round;
                    const double t_tail = 4.0 + (tt >= fill ? std::min((double)((tt + cus - 1) / cus) * t_round, gemm_f32_direct_estimate_us<TileA>(Mt, N, K, cus))
                                                            : gemm_f32_splitk_plan<TileA>(Mt, N, K, SPLITK_FLOATS, cus).us);
                    if (t_head + t_tail < std::min(t_big, t_direct)) return SmallRoute{GS_HYBRID, 1, mh * 256};
                }
            }
            if (t_direct < t_big) return SmallRoute{GS_DIRECT_A, 1, 0};
        }
        return SmallRoute{GS_BIG, 1, 0};
    }
    const SplitPlan pa = gemm_f32_splitk_plan<TileA>(M, N, K, SPLITK_FLOATS, cus);
    if (M <= 640) {   // [r4] one frame (576 rows = 4.5 tiles of 128 rows): the 64-row tile when its plan is estimated faster -- nine row tiles exactly, up
        // to one workgroup per CU without a split (qkv: 216 tiles, epilogue in the kernel: no partials, no reduce launch).  Measured: B = 1 forward
        // -33 us; from two frames on the 64-row tile loses what it gains (its slab is shorter than a load's round trip), so the rule stops here.
        const SplitPlan ps = gemm_f32_splitk_plan<TileS>(M, N, K, SPLITK_FLOATS, cus);
        if (ps.us < pa.us) return SmallRoute{ps.splits == 1 ? GS_DIRECT_S : GS_SPLIT_S, ps.splits, 0};
    }
    return SmallRoute{GS_SPLIT_A, pa.splits, 0};
}
template <class AL, class Epi>
static hipError_t gemm_small(Handle* h, const char* role, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc,
                             int M, int N, int K, float* P, hipStream_t s) {
    const int cus = device_cu_count();
    const SmallRoute r = gemm_small_route(h, std::is_same<AL, ALoadPlain>::value, M, N, K, W.seg, cus, epi_row_free<Epi>::value);
    if constexpr (std::is_same<AL, ALoadPlain>::value && epi_row_free<Epi>::value) {
        if (r.kind == GS_HYBRID) {      // whole rounds of 256 x 256 tiles, then the remaining rows routed as a product of their own (never GS_HYBRID again: fewer tiles than CUs)
            hipError_t e = gemm_big(h, role, al, W, epi, C, ldc, r.head_rows, N, K, s);
            if (e != hipSuccess) return e;
            return gemm_small(h, role, ALoadPlain{al.A + (long)r.head_rows * al.lda, al.lda}, W, epi, C + (long)r.head_rows * ldc, ldc, M - r.head_rows, N, K, P, s);
        }
    }
    switch (r.kind) {
        case GS_BIG: case GS_HYBRID: return gemm_big(h, role, al, W, epi, C, ldc, M, N, K, s);
        case GS_DIRECT_A: return gemm<TileA>(h, role, al, W, epi, C, ldc, M, N, K, s);
        case GS_DIRECT_S: return gemm<TileS>(h, role, al, W, epi, C, ldc, M, N, K, s);
        case GS_SPLIT_S: {
            static const std::string kname64 = std::string("gemm_f32_splitk_kernel<64x128x32,") + AlName<AL>::v + ">+splitk_reduce_kernel<" + EpiName<Epi>::v + ">";
            GemmTimer t(h, s, role, kname64.c_str(), 2.0 * M * N * K);
            return gemm_f32_splitk_launch<TileS>(al, W, epi, C, ldc, P, SPLITK_FLOATS, M, N, K, s, cus, r.splits);
        }
        default: {
            static const std::string kname = std::string("gemm_f32_splitk_kernel<128x128x32,") + AlName<AL>::v + ">+splitk_reduce_kernel<" + EpiName<Epi>::v + ">";
            GemmTimer t(h, s, role, kname.c_str(), 2.0 * M * N * K);
            return gemm_f32_splitk_launch<TileA>(al, W, epi, C, ldc, P, SPLITK_FLOATS, M, N, K, s, cus, r.splits);
        }
    }
}

// fc2 / fc3 of the two encoders: one or two 128-row tiles at small batch -> split K as fc1 does (same row threshold)
static constexpr int SKINNY_ROWS = 1024;       // encoder rows (30 or 34 per frame) below which the fc GEMMs run split-K: B <= 34 / 30
template <class AL, class Epi>
static hipError_t fc_gemm(Handle* h, const char* role, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N, int K,
                          float* P, hipStream_t s) {
    // [r3] bf16 mode only (the fp32 routing, and with it the fp32 bits per batch size, stays as it was): also above the row threshold
    // while the 128 x 128 tiles fill less than half the chip (EgoCap at 128 x 128 heatmaps, B = 32 / 64: 36 / 68 tiles of fc2)
    const bool few_tiles = h && h->precision == EGOTAP_PREC_BF16 && 2L * ((M + 127) / 128) * (N / 128) <= device_cu_count() &&
                           (size_t)M * N * 8 <= SPLITK_FLOATS;
    if ((M >= SKINNY_ROWS && !few_tiles) || N % 128 != 0 || K % 32 != 0) return gemm<TileA>(h, role, al, W, epi, C, ldc, M, N, K, s);
    if (M <= 64) {
        static const std::string kname64 = std::string("gemm_f32_splitk_kernel<64x128x32,") + AlName<AL>::v + ">+splitk_reduce_kernel<" + EpiName<Epi>::v + ">";
        GemmTimer t(h, s, role, kname64.c_str(), 2.0 * M * N * K);
        return gemm_f32_splitk_launch<TileS>(al, W, epi, C, ldc, P, SPLITK_FLOATS, M, N, K, s, device_cu_count());
    }
    static const std::string kname = std::string("gemm_f32_splitk_kernel<128x128x32,") + AlName<AL>::v + ">+splitk_reduce_kernel<" + EpiName<Epi>::v + ">";
    GemmTimer t(h, s, role, kname.c_str(), 2.0 * M * N * K);
    return gemm_f32_splitk_launch<TileA>(al, W, epi, C, ldc, P, SPLITK_FLOATS, M, N, K, s);
}

// ------------------------------------------------------------------------------------------------ workspace
// asks the device once how many workgroups of the one-launch PU chain it keeps resident (pu_chain.h)
static void pu_chain_probe(Handle* h) {
    // A chain launch of an EARLIER call whose row blocks were not co-resident (shared device) was redone on the device by
    // pu_solo_kernel -- the results were right, but it cost a ~0.1 s stall: from now on this handle walks the steps with the per-step
    // kernels (same bits).  Read without synchronising: the word is host memory the device writes through.
    if (h->pu_fault_host && __atomic_load_n(h->pu_fault_host, __ATOMIC_RELAXED) != 0u) {
        __atomic_store_n(h->pu_fault_host, 0u, __ATOMIC_RELAXED);
        h->pu_faults++;
        h->pu_chain = false;
    }
    if (!h->pu_chain) { h->pu_resident[0] = h->pu_resident[1] = 0; return; }      // 0 resident workgroups: pu_chain_launch declines
    if (!h->pu_fault_host) {           // once per handle, at its first forward (never inside a stream capture: wrappers run eagerly first)
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h->pu_fault_host = (unsigned*)hp; h->pu_fault_dev = (unsigned*)dp;
            *h->pu_fault_host = 0u;
        } else {
            if (hp) (void)hipHostFree(hp);
            (void)hipGetLastError();
        }
    }
    if (h->pu_resident[0] <= 0) {
        h->pu_resident[0] = pu_chain_resident<1>();
        h->pu_resident[1] = pu_chain_resident<2>();
    }
}

struct LiftWs {
    size_t X, Y, QKV, CTX, HID, Z1, Z2, POSZ, ROTZ, F0, G0, HS0, F1, G1, HS1, C0, C1, ZERO, HPA, HPB, FAULT, SPLITK, total;
};
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static LiftWs lift_ws(const Handle* h, int B) {
    LiftWs w;
    const size_t M = (size_t)B * h->seq, D = h->D, BT = (size_t)B * h->T, JB = (size_t)h->J * B, H = h->H;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = al256(o + floats * 4); return r; };
    w.X = take(M * D); w.Y = take(M * D); w.QKV = take(M * 3 * D); w.CTX = take(M * D); w.HID = take(M * 4 * D);
    w.Z1 = take(BT * 2048); w.Z2 = take(BT * 512); w.POSZ = take(BT * h->hid); w.ROTZ = take(BT * h->hid);
    w.F0 = take(JB * (H + 2 * h->hid)); w.G0 = take(JB * 4 * H); w.HS0 = take(JB * H);
    w.F1 = take(JB * H); w.G1 = take(JB * 4 * H); w.HS1 = take(JB * H);
    w.C0 = take((size_t)B * H); w.C1 = take((size_t)B * H); w.ZERO = take((size_t)B * H);
    w.HPA = take((size_t)h->J * B * H);                            // the propagation units' gated state: one [B, H] buffer per step
    w.HPB = w.HPA + al256((size_t)B * H * 4);                      // (the per-step fallback kernels ping-pong between the first two)
    w.FAULT = take(64);                                            // fault word of the one-launch recurrence (pu_chain.h)
    w.SPLITK = take(SPLITK_FLOATS);       // split-K partial sums of the small-batch GEMMs (gemm_small)
    w.total = o;
    return w;
}

#if EGOTAP_IN(0)
extern "C" int egotap_lift_workspace_bytes(egotap_handle h, int B, size_t* bytes) {
    EGO_CHECK(h && bytes, "null argument");
    EGO_CHECK(B >= 0, "negative batch");
    *bytes = lift_ws(h, B > 0 ? B : 1).total;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_lift_intermediate(egotap_handle h, int B, const char* name, size_t* offset, int64_t* numel) {
    EGO_CHECK(h && name && offset && numel, "null argument");
    const LiftWs w = lift_ws(h, B > 0 ? B : 1);
    const int64_t M = (int64_t)B * h->seq;
    if (!strcmp(name, "tokens")) { *offset = w.Y; *numel = M * h->D; }
    else if (!strcmp(name, "x")) { *offset = w.X; *numel = M * h->D; }
    else if (!strcmp(name, "pos_embed")) { *offset = w.POSZ; *numel = (int64_t)B * h->T * h->hid; }
    else if (!strcmp(name, "rot_embed")) { *offset = w.ROTZ; *numel = (int64_t)B * h->T * h->hid; }
    else if (!strcmp(name, "skel_embed")) { *offset = w.HS1; *numel = (int64_t)h->J * B * h->H; }
    else { egotap_set_error("unknown intermediate '%s'", name); return EGOTAP_ERR_INVALID; }
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_set_pu_chain(egotap_handle h, int enable) {
    EGO_CHECK(h, "null handle");
    h->pu_chain = enable != 0;
    h->pu_resident[0] = h->pu_resident[1] = -1;
    return EGOTAP_OK;
}
extern "C" int egotap_debug_pu_drop_workgroups(egotap_handle h, int n) {
    EGO_CHECK(h && n >= 0 && n < 16, "egotap_debug_pu_drop_workgroups: n in [0, 16)");
    h->pu_debug_drop = n;
    return EGOTAP_OK;
}
extern "C" int egotap_pu_chain_status(egotap_handle h, int* enabled, int* faults) {
    EGO_CHECK(h, "null handle");
    // meaningful after the caller has synchronised the stream of the call it asks about (the word is written by the device)
    if (h->pu_fault_host && __atomic_load_n(h->pu_fault_host, __ATOMIC_RELAXED) != 0u) {
        __atomic_store_n(h->pu_fault_host, 0u, __ATOMIC_RELAXED);
        h->pu_faults++;
        h->pu_chain = false;
        h->pu_resident[0] = h->pu_resident[1] = -1;
    }
    if (enabled) *enabled = h->pu_chain ? 1 : 0;
    if (faults) *faults = h->pu_faults;
    return EGOTAP_OK;
}
extern "C" int egotap_set_precision(egotap_handle h, int mode) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(mode == EGOTAP_PREC_F32 || mode == EGOTAP_PREC_BF16X3 || mode == EGOTAP_PREC_BF16, "egotap_set_precision: unknown mode %d", mode);
    h->precision = mode;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_set_weight_scratch(egotap_handle h, void* buf, size_t bytes) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(((uintptr_t)buf & 15) == 0, "egotap_set_weight_scratch: 16-byte alignment");
    h->wscratch = (__bf16*)buf;
    h->wscratch_bytes = buf ? bytes : 0;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_set_act_scratch(egotap_handle h, void* buf, size_t bytes) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(((uintptr_t)buf & 15) == 0, "egotap_set_act_scratch: 16-byte alignment");
    h->ascratch = (__bf16*)buf;
    h->ascratch_bytes = buf ? bytes : 0;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_lift_debug_stop(egotap_handle h, int stage) {
    EGO_CHECK(h, "null handle");
    h->debug_stop = stage;   // 0: full forward; 1: after embeddings; 2+i: after ViT layer i  ("x" holds the state)
    return EGOTAP_OK;
}
#endif

// fc1 of an encoder (rows gathered from the ViT tokens / the limb heatmaps) as an NT product: [r5] on the 64-deep kernel where its shape rules hold
// (D, HW multiples of 64: a K-tile inside one token / map), the 32-deep kernel otherwise or when egotap_debug_gemm_bk pins it; same k order per
// output element: same bits
template <class Epi>
static hipError_t fc1_nt(Handle* h, int which, const __bf16* src, const __bf16* w, long K, const Epi& ep, int BT, int cus, hipStream_t s) {
    const int HW = h->cfg.hm_size * h->cfg.hm_size;
    const bool deep = g_gemm_bf16s_bk != 32 && K % 64 == 0 && K >= 128 && (which == 0 ? h->D % 64 == 0 : HW % 64 == 0) && (long)256 * K * 2 < (1L << 31);
    if (which == 0) {
        // [r5] tensors under 4 GB (every shipped batch): scalar origin + 32-bit lane offsets (X64TokensS / X64RotS); the pointer loaders otherwise
        const size_t tok_bytes = (size_t)(BT / h->T) * h->seq * h->D * 2;
        if (deep && g_conv_addressing == 0 && tok_bytes < ((size_t)1 << 32) - 4096)
            return gemm_bf16s64_launch_x(X64TokensS{src, 0u, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, w, K, ep, BT, 2048, (int)K, cus, s);
        if (deep) return gemm_bf16s64_launch_x(X64Tokens{src, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, w, K, ep, BT, 2048, (int)K, cus, s);
        return gemm_bf16s_launch(XTokens{src, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, w, K, ep, BT, 2048, (int)K, cus, s);
    }
    const size_t hm_bytes = (size_t)(BT / (2 * h->J)) * h->C * HW * 2;
    if (deep && g_conv_addressing == 0 && hm_bytes < ((size_t)1 << 32) - 4096) return gemm_bf16s64_launch_x(X64RotS{src, 0u, h->C, h->J, HW}, w, K, ep, BT, 2048, (int)K, cus, s);
    if (deep) return gemm_bf16s64_launch_x(X64Rot{src, h->C, h->J, HW}, w, K, ep, BT, 2048, (int)K, cus, s);
    return gemm_bf16s_launch(XRot{src, h->C, h->J, HW}, w, K, ep, BT, 2048, (int)K, cus, s);
}

// ------------------------------------------------------------------------------------------------ forward
static hipError_t launch_ln(const float* x, float* y, const float* g, const float* b, int rows, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int blocks = min((rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(layernorm_f32_kernel<1024>, dim3(blocks), dim3(256), 0, s, x, y, g, b, rows, eps);
    return hipGetLastError();
}

#if EGOTAP_IN(0)
// [r4] x = x + ctx W^T + bias, y = LayerNorm(x): at serving batches (the product is split over K) the reduce of the partials and the LayerNorm
// that follows are ONE launch (splitk_reduce_res_ln_kernel: same bits as the two); otherwise gemm_small + launch_ln as before.
static hipError_t gemm_res_ln(Handle* h, const char* role, const float* A, long lda, const float* Wp, const float* bias, float* X, int M, int D, int K,
                              const float* ln_g, const float* ln_b, float* Y, float* P, hipStream_t s) {
    if (D == 1024) {
        const SmallRoute r = gemm_small_route(h, true, M, D, K, D, device_cu_count());      // gemm_small's own decision, not a copy of its rule
        if ((r.kind == GS_SPLIT_S || r.kind == GS_SPLIT_A) && r.splits > 1) {
            static const std::string kname = "gemm_f32_splitk_kernel<ALoadPlain>+splitk_reduce_res_ln_kernel";
            GemmTimer t(h, s, role, kname.c_str(), 2.0 * M * D * K);
            hipError_t e = r.kind == GS_SPLIT_S ? gemm_f32_splitk_partials<TileS>(ALoadPlain{A, lda}, segmat1(Wp, D, K), P, SPLITK_FLOATS, M, D, K, s, r.splits)
                                                : gemm_f32_splitk_partials<TileA>(ALoadPlain{A, lda}, segmat1(Wp, D, K), P, SPLITK_FLOATS, M, D, K, s, r.splits);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(splitk_reduce_res_ln_kernel<1024>, dim3(min((M + 3) / 4, 256 * 8)), dim3(256), 0, s, (const float*)P, bias, (const float*)X, X, ln_g,
                               ln_b, Y, M, r.splits, 1e-12f);
            return hipGetLastError();
        }
    }
    hipError_t e = gemm_small(h, role, ALoadPlain{A, lda}, segmat1(Wp, D, K), EpiBiasRes{segvec1(bias, D), X, D}, X, D, M, D, K, P, s);
    if (e != hipSuccess) return e;
    return launch_ln(X, Y, ln_g, ln_b, M, 1e-12f, s);
}

extern "C" int egotap_lift_forward(egotap_handle h, const float* hm, int B, float* pose, void* ws, size_t ws_bytes,
                                   void* stream) {
    EGO_CHECK(h, "null handle");
    if (B == 0) return EGOTAP_OK;
    EGO_CHECK(B > 0 && hm && pose && ws, "egotap_lift_forward: null argument or negative batch");
    EGO_CHECK(((uintptr_t)hm & 15) == 0 && ((uintptr_t)ws & 255) == 0, "hm must be 16-byte and ws 256-byte aligned");
    int rc = lift_resolve(h);
    if (rc != EGOTAP_OK) return rc;
    const LiftWs w = lift_ws(h, B);
    if (ws_bytes < w.total) {
        egotap_set_error("workspace too small: %zu bytes given, %zu needed for B=%d", ws_bytes, w.total, B);
        return EGOTAP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const LiftParams& p = h->lp;
    char* base = (char*)ws;
    auto F = [&](size_t off) { return (float*)(base + off); };
    float *X = F(w.X), *Y = F(w.Y), *QKV = F(w.QKV), *CTX = F(w.CTX), *HID = F(w.HID);
    float* SPK = F(w.SPLITK);
    float *Z1 = F(w.Z1), *Z2 = F(w.Z2), *POSZ = F(w.POSZ), *ROTZ = F(w.ROTZ);
    float *F0 = F(w.F0), *G0 = F(w.G0), *HS0 = F(w.HS0), *F1 = F(w.F1), *G1 = F(w.G1), *HS1 = F(w.HS1);
    float *C0 = F(w.C0), *C1 = F(w.C1), *ZERO = F(w.ZERO), *HPA = F(w.HPA), *HPB = F(w.HPB);
    const int D = h->D, M = B * h->seq, BT = B * h->T, J = h->J, H = h->H, hid = h->hid, JB = J * B;
    const int S = h->cfg.hm_size, HW = S * S;
    using Tile = TileA;

    // EGOTAP_PREC_BF16 at a batch that fills the chip: bf16 ACTIVATION STORAGE (the same workspace slices hold bf16): LayerNorm, the
    // GEMM epilogues and attention write bf16, every GEMM reads bf16 operands through the LDS DMA (gemm_bf16s.h); weights are rounded
    // into the caller's weight scratch right before each launch (nothing cached: the fp32 parameters stay the source of truth)
#ifndef EGOTAP_BF16S_MIN_ROWS
#define EGOTAP_BF16S_MIN_ROWS 512       // [r4] 4096 before: a B = 4 forward (2304 rows) took 2.1 ms on the fp32-tensor path, 1.5 ms on this one (B = 1: 1.46 -> 1.38)
#endif
    constexpr int g_bf16s_min_rows = EGOTAP_BF16S_MIN_ROWS;
    // (sequence lengths that are not a multiple of 32 -- heatmap sides 32, 48, 96 ...: the bf16-storage attention tiles whole 32-key blocks -- stay on
    // fp32 tensors with bf16 products in the GEMMs and the exact-fp32 attention kernel, which masks a ragged last key tile)
    const bool bf16s = h->precision == EGOTAP_PREC_BF16 && D == 1024 && M >= g_bf16s_min_rows && h->wscratch != nullptr && h->seq % 32 == 0 &&
                       h->wscratch_bytes >= (size_t)2 * 2048 * (size_t)(h->ppd * h->ppd * D);
    // H1+H2: tile -> patch embed -> mask token -> + position embeddings
    if (bf16s) {
        // [r3] on the bf16-storage GEMM: the heatmaps' bf16 copy in the (still free) MLP buffer, by LDS DMA; zeros for the dummy cells in SPK
        __bf16* hmb0 = (__bf16*)HID;
        const long n8 = (long)B * h->C * HW / 8;
        hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, hm, hmb0, n8);
        hipLaunchKernelGGL(prep_weight_kernel, dim3((256 + 63) / 64, (D + 63) / 64), dim3(256), 0, s, p.patch_w, h->wscratch, (__bf16*)nullptr, D, 256, (long)D);
        EGO_HIP(zero_fill(SPK, 256, s));
        const XPatch xl{hmb0, (const __bf16*)SPK, h->C, S, h->seq, h->side, h->ppd, h->grid, h->T};
        const SEpiPatchF32 ep{p.patch_b, p.mask_tok, p.pos_emb, X, D, h->seq, h->side, h->ppd, h->grid, h->T};
        EGO_HIP(gemm_bf16s_launch(xl, h->wscratch, 256L, ep, M, D, 256, device_cu_count(), s));
    } else {
        ALoadPatch al{hm, h->C, S, h->seq, h->side, h->ppd, h->grid, h->T};
        // [r3] fp32 at a batch that fills the chip: a zero page for the dummy cells (the head of the PU chain's ZERO slice, which is cleared as a
        // whole further down and not read before) makes the loader pure address math -> the LDS-DMA kernel instead of the register-staged one
        if (h->precision == EGOTAP_PREC_F32 && (long)((M + 255) / 256) * (D / 256) >= 160 && (size_t)B * H * 4 >= 1024) {
            EGO_HIP(zero_fill(ZERO, 1024, s));
            al.zeros = ZERO;
        }
        EpiPatch ep{p.patch_b, p.mask_tok, p.pos_emb, D, h->seq, h->side, h->ppd, h->grid, h->T};
        EGO_HIP((gemm_small(h, "patch_embed", al, segmat1(p.patch_w, D, 256), ep, X, D, M, D, 256, SPK, s)));
    }
    if (h->debug_stop == 1) return EGOTAP_OK;
    if (bf16s) {
        __bf16 *Yb = (__bf16*)Y, *QKVb = (__bf16*)QKV, *CTXb = (__bf16*)CTX, *HIDb = (__bf16*)HID, *Wb = h->wscratch;
        float* bias3 = SPK;
        const int cu = device_cu_count();
        auto prep = [&](const float* wsrc, __bf16* dst, int n, int k) {
            hipLaunchKernelGGL(prep_weight_kernel, dim3((k + 63) / 64, (n + 63) / 64), dim3(256), 0, s, wsrc, dst, (__bf16*)nullptr, n, k, (long)n);
        };
        auto ln = [&](const float* gg, const float* bb) {
            hipLaunchKernelGGL(ln_fwd_bf16_kernel<1024>, dim3((M + 3) / 4), dim3(256), 0, s, (const float*)X, Yb, gg, bb, (float*)nullptr, (float*)nullptr, M, 1e-12f);
        };
        for (int i = 0; i < h->cfg.vit_layers; ++i) {
            const auto& L = p.layer[i];
            ln(L.ln1_g, L.ln1_b);
            prep(L.q_w, Wb, D, D); prep(L.k_w, Wb + (size_t)D * D, D, D); prep(L.v_w, Wb + (size_t)2 * D * D, D, D);
            hipLaunchKernelGGL(concat3_kernel, dim3((D + 255) / 256), dim3(256), 0, s, L.q_b, L.k_b, L.v_b, bias3, D);
            EGO_HIP(hipGetLastError());
            EGO_HIP(gemm_bf16s_plain_launch(XPlain{Yb, (long)D}, Wb, (long)D, SEpiBf16{bias3, QKVb, 3L * D}, M, 3 * D, D, cu, s));
            EGO_HIP(attention_bf16s_fwd_launch(QKVb, CTXb, nullptr, B, h->seq, h->cfg.vit_heads, s));
            prep(L.o_w, Wb, D, D);
            EGO_HIP(gemm_bf16s_plain_launch(XPlain{CTXb, (long)D}, Wb, (long)D, SEpiResF32{L.o_b, X, X, (long)D}, M, D, D, cu, s));
            ln(L.ln2_g, L.ln2_b);
            prep(L.up_w, Wb, 4 * D, D);
            EGO_HIP(gemm_bf16s_plain_launch(XPlain{Yb, (long)D}, Wb, (long)D, SEpiGelu{L.up_b, HIDb, 4L * D}, M, 4 * D, D, cu, s));
            prep(L.dn_w, Wb, D, 4 * D);
            EGO_HIP(gemm_bf16s_plain_launch(XPlain{HIDb, 4L * D}, Wb, 4L * D, SEpiResF32{L.dn_b, X, X, (long)D}, M, D, 4 * D, cu, s));
            if (h->debug_stop == 2 + i) return EGOTAP_OK;
        }
        ln(p.lnf_g, p.lnf_b);                               // tokens (bf16) -> fc1's gathering loader
        auto bnf = [](const LiftParams::Fc& f, float* out) { return SEpiBnLreluF32{f.b, f.g, f.beta, f.mean, f.var, 1e-5f, 0.2f, out, 2048L}; };
        auto bn = [](const LiftParams::Fc& f) { return EpiBnLrelu{f.b, f.g, f.beta, f.mean, f.var, 1e-5f, 0.2f}; };
        {
            const int K1 = h->ppd * h->ppd * D;
            prep(p.pos_fc[0].w, Wb, 2048, K1);
            // [r3] few encoder rows (EgoCap at 128 x 128 heatmaps, B = 32: 1088 rows = 40 tiles for 256 CUs, K = 65536): split K over the chip
            const XTokens xt{Yb, h->T, D, h->seq, h->side, h->ppd, h->grid};
            const int sp = gemm_bf16s_ksplit(BT, 2048, K1, cu, SPLITK_FLOATS);
            if (sp > 1) EGO_HIP(gemm_bf16s_splitk_launch(xt, Wb, (long)K1, bnf(p.pos_fc[0], Z1), SPK, sp, BT, 2048, K1, cu, s));
            else EGO_HIP(fc1_nt(h, 0, Yb, Wb, (long)K1, bnf(p.pos_fc[0], Z1), BT, cu, s));
            EGO_HIP((fc_gemm(h, "pos_fc2", ALoadPlain{Z1, 2048}, segmat1(p.pos_fc[1].w, 512, 2048), bn(p.pos_fc[1]), Z2, 512, BT, 512, 2048, SPK, s)));
            EGO_HIP((fc_gemm(h, "pos_fc3", ALoadPlain{Z2, 512}, segmat1(p.pos_fc[2].w, hid, 512), bn(p.pos_fc[2]), POSZ, hid, BT, hid, 512, SPK, s)));
        }
        {
            __bf16* hmb = HIDb;                             // the MLP's hidden buffer is free now: bf16 copy of the input heatmaps
            const long n8 = (long)B * h->C * HW / 8;
            hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, hm, hmb, n8);
            prep(p.rot_fc[0].w, Wb, 2048, 2 * HW);
            EGO_HIP(hipGetLastError());
            const XRot xr{hmb, h->C, J, HW};
            const int sp = gemm_bf16s_ksplit(BT, 2048, 2 * HW, cu, SPLITK_FLOATS);
            if (sp > 1) EGO_HIP(gemm_bf16s_splitk_launch(xr, Wb, 2L * HW, bnf(p.rot_fc[0], Z1), SPK, sp, BT, 2048, 2 * HW, cu, s));
            else EGO_HIP(fc1_nt(h, 1, hmb, Wb, 2L * HW, bnf(p.rot_fc[0], Z1), BT, cu, s));
            EGO_HIP((fc_gemm(h, "rot_fc2", ALoadPlain{Z1, 2048}, segmat1(p.rot_fc[1].w, 512, 2048), bn(p.rot_fc[1]), Z2, 512, BT, 512, 2048, SPK, s)));
            EGO_HIP((fc_gemm(h, "rot_fc3", ALoadPlain{Z2, 512}, segmat1(p.rot_fc[2].w, hid, 512), bn(p.rot_fc[2]), ROTZ, hid, BT, hid, 512, SPK, s)));
        }
    } else {
    // H3-H8: pre-LN transformer layers.  [r4] Every LayerNorm but the first follows a residual projection: gemm_res_ln runs the two as one launch
    // where the projection is split over K (serving batches); ln1 of layer i + 1 (or the final LayerNorm) therefore rides with layer i's MLP.
    const int NL = h->cfg.vit_layers;
    EGO_HIP(launch_ln(X, Y, NL > 0 ? p.layer[0].ln1_g : p.lnf_g, NL > 0 ? p.layer[0].ln1_b : p.lnf_b, M, 1e-12f, s));
    for (int i = 0; i < NL; ++i) {
        const auto& L = p.layer[i];
        {
            SegMat Wqkv; Wqkv.p[0] = L.q_w; Wqkv.p[1] = L.k_w; Wqkv.p[2] = L.v_w; Wqkv.seg = D; Wqkv.ld = D;
            SegVec bqkv; bqkv.p[0] = L.q_b; bqkv.p[1] = L.k_b; bqkv.p[2] = L.v_b; bqkv.seg = D;
            EGO_HIP((gemm_small(h, "qkv", ALoadPlain{Y, D}, Wqkv, EpiBias{bqkv}, QKV, 3L * D, M, 3 * D, D, SPK, s)));
        }
        if (h->precision == EGOTAP_PREC_BF16X3 && h->seq % 32 == 0) EGO_HIP(attention_bf16_launch<3>(QKV, CTX, B, h->seq, h->cfg.vit_heads, s));
        else if (h->precision == EGOTAP_PREC_BF16 && h->seq % 32 == 0) EGO_HIP(attention_bf16_launch<1>(QKV, CTX, B, h->seq, h->cfg.vit_heads, s));
        else EGO_HIP(attention_f32_launch(QKV, CTX, B, h->seq, h->cfg.vit_heads, s, nullptr, SPK, SPLITK_FLOATS, device_cu_count()));   // (SPK: free between the GEMMs; key-split partials at B <= 2)
        EGO_HIP(gemm_res_ln(h, "attn_out", CTX, D, L.o_w, L.o_b, X, M, D, D, L.ln2_g, L.ln2_b, Y, SPK, s));
        EGO_HIP((gemm_small(h, "mlp_up", ALoadPlain{Y, D}, segmat1(L.up_w, 4 * D, D), EpiBiasGelu{segvec1(L.up_b, 4 * D)}, HID, 4L * D, M, 4 * D, D, SPK, s)));
        const bool last = i + 1 == NL;
        EGO_HIP(gemm_res_ln(h, "mlp_down", HID, 4L * D, L.dn_w, L.dn_b, X, M, D, 4 * D, last ? p.lnf_g : p.layer[i + 1].ln1_g, last ? p.lnf_b : p.layer[i + 1].ln1_b, Y,
                            SPK, s));
        if (h->debug_stop == 2 + i) return EGOTAP_OK;
    }
    // H9-H10: per-heatmap regroup folded into fc1's A loader; fc blocks with folded BatchNorm + LeakyReLU
    auto bn = [](const LiftParams::Fc& f) { return EpiBnLrelu{f.b, f.g, f.beta, f.mean, f.var, 1e-5f, 0.2f}; };
    // fc1 of the two encoders has few rows (30 / 34 per frame) and a huge K (16384 ... 65536): below 100 tiles of 256 x 256 (UnrealEgo:
    // B < 107; EgoCap with 128 x 128 heatmaps: B < 95) most CUs would idle, so K is split over them (exact fp32 in every mode)
    // (the bf16 modes keep their own kernels down to 1024 rows: at 16x the MFMA rate a quarter-filled chip still beats fp32 split-K)
    const bool skinny_fc1 = BT < SKINNY_ROWS || (h->precision == EGOTAP_PREC_F32 && (long)((BT + 255) / 256) * (2048 / 256) < 100);
    {
        const int K1 = h->ppd * h->ppd * D;
        ALoadTokens al{Y, h->T, D, h->seq, h->side, h->ppd, h->grid};
        if (skinny_fc1 && BT <= 64)     // [r4] at most 64 rows: the 64-row tile (TileS)
            EGO_HIP((gemm_f32_splitk_launch<TileS>(al, segmat1(p.pos_fc[0].w, 2048, K1), bn(p.pos_fc[0]), Z1, 2048, SPK, SPLITK_FLOATS, BT, 2048, K1, s, device_cu_count())));
        else if (skinny_fc1)       // few row tiles: split K over the CUs
            EGO_HIP((gemm_f32_splitk_launch<TileA>(al, segmat1(p.pos_fc[0].w, 2048, K1), bn(p.pos_fc[0]), Z1, 2048, SPK, SPLITK_FLOATS, BT, 2048, K1, s)));
        else
            EGO_HIP((gemm_big(h, "pos_fc1", al, segmat1(p.pos_fc[0].w, 2048, K1), bn(p.pos_fc[0]), Z1, 2048, BT, 2048, K1, s)));
        EGO_HIP((fc_gemm(h, "pos_fc2", ALoadPlain{Z1, 2048}, segmat1(p.pos_fc[1].w, 512, 2048), bn(p.pos_fc[1]), Z2, 512, BT, 512, 2048, SPK, s)));
        EGO_HIP((fc_gemm(h, "pos_fc3", ALoadPlain{Z2, 512}, segmat1(p.pos_fc[2].w, hid, 512), bn(p.pos_fc[2]), POSZ, hid, BT, hid, 512, SPK, s)));
    }
    // H11-H12: rotation (cos/sin) heatmaps straight from the input tensor
    {
        ALoadRot al{hm, h->C, J, HW};
        if (skinny_fc1 && BT <= 64)
            EGO_HIP((gemm_f32_splitk_launch<TileS>(al, segmat1(p.rot_fc[0].w, 2048, 2L * HW), bn(p.rot_fc[0]), Z1, 2048, SPK, SPLITK_FLOATS, BT, 2048, 2 * HW, s, device_cu_count())));
        else if (skinny_fc1)
            EGO_HIP((gemm_f32_splitk_launch<TileA>(al, segmat1(p.rot_fc[0].w, 2048, 2L * HW), bn(p.rot_fc[0]), Z1, 2048, SPK, SPLITK_FLOATS, BT, 2048, 2 * HW, s)));
        else
            EGO_HIP((gemm_big(h, "rot_fc1", al, segmat1(p.rot_fc[0].w, 2048, 2L * HW), bn(p.rot_fc[0]), Z1, 2048, BT, 2048, 2 * HW, s)));
        EGO_HIP((fc_gemm(h, "rot_fc2", ALoadPlain{Z1, 2048}, segmat1(p.rot_fc[1].w, 512, 2048), bn(p.rot_fc[1]), Z2, 512, BT, 512, 2048, SPK, s)));
        EGO_HIP((fc_gemm(h, "rot_fc3", ALoadPlain{Z2, 512}, segmat1(p.rot_fc[2].w, hid, 512), bn(p.rot_fc[2]), ROTZ, hid, BT, hid, 512, SPK, s)));
    }
    }       // fp32 tensors in HBM
    // H13-H14: propagation units.  State-independent projections of all J steps as GEMMs (rows time-major t*B+b) ...
    // ([r4] through fc_gemm: below 1024 rows -- serving batches: J rows per frame -- a 128-row tiling has one row tile and 4-16 column tiles,
    // 24-41 us per launch at B = 1 for work the chip does in 2; K is split like the encoders' fc layers)
    const int x = 2 * hid, NF0 = H + x;
    ALoadStereo xs{POSZ, B, J, hid};
    EGO_HIP((fc_gemm(h, "pu0_x2f", xs, segmat1(p.x2f0_w, NF0, x), EpiBias{segvec1(p.x2f0_b, NF0)}, F0, NF0, JB, NF0, x, SPK, s)));
    EGO_HIP((fc_gemm(h, "pu0_x2h", xs, segmat1(p.x2h0_w, 4 * H, x), EpiBias{segvec1(p.x2h0_b, 4 * H)}, G0, 4L * H, JB, 4 * H, x, SPK, s)));
    {
        ALoadStereoGated bs{ALoadStereo{ROTZ, B, J, hid}, F0, NF0, H};
        EGO_HIP((fc_gemm(h, "pu0_b2h", bs, segmat1(p.b2h0_w, 4 * H, x), EpiBiasRes{segvec1(p.b2h0_b, 4 * H), G0, 4L * H}, G0, 4L * H, JB, 4 * H, x, SPK, s)));
    }
    // ... then the two J-step recurrences (layer 0 never reads layer-1 state, so the layers run one after the other)
    EGO_HIP(zero_fill(C0, (size_t)(w.ZERO - w.C0) + al256((size_t)B * H * 4), s));   // C0, C1, ZERO are contiguous (256-byte aligned slices)
    pu_chain_probe(h);
    unsigned* FAULT = (unsigned*)(base + w.FAULT);
    const PuChain ch0{F0, (long)B * NF0, NF0, G0, (long)B * 4 * H, nullptr, p.h2h0_w, p.h2h0_b, nullptr, 0, HS0, (long)B * H, HPA, (long)B * H, B, H, J,
                      FAULT, h->pu_fault_dev};
    if (!pu_chain_launch(s, h->pu_resident[0], h->pu_resident[1], ch0, B, h->pu_debug_drop))
    for (int t = 0; t < J; ++t) {       // the gated state of step t + 1 comes out of step t (ping-pong buffers; zeros at t = 0)
        const float* hp_in = t == 0 ? ZERO : ((t & 1) ? HPA : HPB);
        pu_step_launch(s, B, H, hp_in, G0 + (size_t)t * B * 4 * H, p.h2h0_w, p.h2h0_b, C0, C0, HS0 + (size_t)t * B * H,
                       t + 1 < J ? F0 + (size_t)(t + 1) * B * NF0 : nullptr, NF0, (t & 1) ? HPB : HPA, nullptr);
    }
    EGO_HIP(hipGetLastError());
    EGO_HIP((fc_gemm(h, "pu1_x2f", ALoadPlain{HS0, H}, segmat1(p.x2f1_w, H, H), EpiBias{segvec1(p.x2f1_b, H)}, F1, H, JB, H, H, SPK, s)));
    EGO_HIP((fc_gemm(h, "pu1_x2h", ALoadPlain{HS0, H}, segmat1(p.x2h1_w, 4 * H, H), EpiBias{segvec1(p.x2h1_b, 4 * H)}, G1, 4L * H, JB, 4 * H, H, SPK, s)));
    const PuChain ch1{F1, (long)B * H, H, G1, (long)B * 4 * H, nullptr, p.h2h1_w, p.h2h1_b, nullptr, 0, HS1, (long)B * H, HPA, (long)B * H, B, H, J,
                      FAULT, h->pu_fault_dev};
    if (!pu_chain_launch(s, h->pu_resident[0], h->pu_resident[1], ch1, B, h->pu_debug_drop))
    for (int t = 0; t < J; ++t) {
        const float* hp_in = t == 0 ? ZERO : ((t & 1) ? HPA : HPB);
        pu_step_launch(s, B, H, hp_in, G1 + (size_t)t * B * 4 * H, p.h2h1_w, p.h2h1_b, C1, C1, HS1 + (size_t)t * B * H,
                       t + 1 < J ? F1 + (size_t)(t + 1) * B * H : nullptr, H, (t & 1) ? HPB : HPA, nullptr);
    }
    EGO_HIP(hipGetLastError());
    // H15: per-joint pose head (+ global offset and head joint for UnrealEgo)
    hipLaunchKernelGGL(pose_head_kernel, dim3(B, h->J + 1), dim3(256), 0, s, POSZ, HS1, p.pose_w, p.pose_b, p.glob_w, p.glob_b, pose, B,
                       J, hid, H, h->cfg.estimate_head);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// ------------------------------------------------------------------------------------------------ heatmap estimator
static int hm_resolve(Handle* h, int net) {
    if (h->hm_resolved[net]) return EGOTAP_OK;
    bool ok = true;
    HmParams& p = h->hp[net];
    const std::string bb = "backbone.backbone.backbone.";
    auto bn = [&](const std::string& k) {
        auto n = h->bound[net].find(k + ".num_batches_tracked");
        long long* nbt = (n != h->bound[net].end() && n->second.dtype == EGOTAP_I64 && n->second.numel == 1) ? (long long*)n->second.ptr : nullptr;
        return HmParams::Bn{P(h, net, k + ".weight", ok), P(h, net, k + ".bias", ok), P(h, net, k + ".running_mean", ok),
                            P(h, net, k + ".running_var", ok), nbt};
    };
    p.stem_w = P(h, net, bb + "conv1.weight", ok);
    p.stem_bn = bn(bb + "bn1");
    for (int i = 0; i < 4; ++i) p.nblk[i] = hm_nblk(h, i);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < p.nblk[i]; ++b) {
            const std::string k = bb + "layer" + std::to_string(i + 1) + "." + std::to_string(b);
            auto& B = p.blk[i][b];
            B.w1 = P(h, net, k + ".conv1.weight", ok); B.bn1 = bn(k + ".bn1");
            B.w2 = P(h, net, k + ".conv2.weight", ok); B.bn2 = bn(k + ".bn2");
            B.wd = nullptr;
            if (b == 0 && i > 0) { B.wd = P(h, net, k + ".downsample.0.weight", ok); B.bnd = bn(k + ".downsample.1"); }
        }
    const std::string a = "after_backbone.";
    for (int k = 0; k < 4; ++k) {
        const std::string n = a + "layer" + std::to_string(k + 1) + "_1x1.0";
        p.l1x1[k] = {P(h, net, n + ".weight", ok), P(h, net, n + ".bias", ok)};
    }
    for (int k = 0; k < 3; ++k) {
        const std::string n = a + "conv_up" + std::to_string(k + 1) + ".0";
        p.up[k] = {P(h, net, n + ".weight", ok), P(h, net, n + ".bias", ok)};
    }
    p.head = {P(h, net, a + "conv_heatmap.weight", ok), P(h, net, a + "conv_heatmap.bias", ok)};
    if (!ok) return EGOTAP_ERR_UNBOUND;
    h->hm_resolved[net] = true;
    return EGOTAP_OK;
}

//                      taps stride log2W CO_T WCO WPX CI_S
using C3s1_128_co64 = ConvCfg<9, 1, 7,  64, 2, 4, 8>;    // 128-wide maps: 512x512 RGB (BASELINE config 5)
using C3s1_128      = ConvCfg<9, 1, 7, 128, 2, 4, 8>;
using C3s2_64       = ConvCfg<9, 2, 6,  64, 2, 4, 8>;
using C1s1_128      = ConvCfg<1, 1, 7,  64, 2, 4, 32>;
using C1s2_64       = ConvCfg<1, 2, 6,  64, 2, 4, 8>;
using C3s1_64_co64  = ConvCfg<9, 1, 6,  64, 2, 4, 8>;
using C3s1_64       = ConvCfg<9, 1, 6, 128, 2, 4, 8>;
using C3s1_32       = ConvCfg<9, 1, 5, 128, 2, 4, 8>;
using C3s1_16       = ConvCfg<9, 1, 4, 128, 2, 4, 8>;
using C3s1_8        = ConvCfg<9, 1, 3, 128, 2, 4, 8>;
using C3s1_16_co64  = ConvCfg<9, 1, 4,  64, 2, 4, 8>;    // [r3] the same two widths with 64-channel tiles: twice the workgroups where 128-channel
using C3s1_8_co64   = ConvCfg<9, 1, 3,  64, 2, 4, 8>;    // tiles leave CUs without one (training batches, serving batches); same bits (k order unchanged)
using C3s2_32       = ConvCfg<9, 2, 5,  64, 2, 4, 8>;
using C3s2_16       = ConvCfg<9, 2, 4,  64, 2, 4, 8>;
using C3s2_8        = ConvCfg<9, 2, 3,  64, 2, 4, 8>;
using C1s1_64       = ConvCfg<1, 1, 6,  64, 2, 4, 32>;
using C1s1_32       = ConvCfg<1, 1, 5, 128, 2, 4, 32>;
using C1s1_16       = ConvCfg<1, 1, 4, 128, 2, 4, 32>;
using C1s1_8        = ConvCfg<1, 1, 3, 128, 2, 4, 32>;
using C1s2_32       = ConvCfg<1, 2, 5,  64, 2, 4, 8>;
using C1s2_16       = ConvCfg<1, 2, 4,  64, 2, 4, 8>;
using C1s2_8        = ConvCfg<1, 2, 3,  64, 2, 4, 8>;

template <class Cfg>
static hipError_t conv(Handle* h, const char* role, const ConvArgs& a, hipStream_t s) {
    static const std::string kname = std::string("conv_f32_kernel<") + (Cfg::TAPS == 9 ? "3x3" : "1x1") + ",s" +
                                     std::to_string(Cfg::STRIDE) + ",W" + std::to_string(Cfg::W) + ",co" +
                                     std::to_string(Cfg::CO_T) + ">";
    GemmTimer t(h, s, role, kname.c_str(), 2.0 * a.Cout * a.Cin * Cfg::TAPS * (double)a.Nimg * Cfg::W * Cfg::W);
    ConvArgs b = a;
    b.part = h->conv_part;
    b.part_floats = h->conv_part_floats;
    return conv_f32_launch<Cfg>(b, s, device_cu_count());
}

template <class Cfg>
static hipError_t conv_bf(Handle* h, const char* role, const ConvArgs& a, hipStream_t s) {
    static const std::string kname = std::string("conv_bf16_kernel<3x3,s1,W") + std::to_string(Cfg::W) + (Cfg::NP == 3 ? ",bf16x3>" : ",bf16>");
    GemmTimer t(h, s, role, kname.c_str(), 2.0 * a.Cout * a.Cin * 9 * (double)a.Nimg * Cfg::W * Cfg::W);
    return conv_bf16_launch<Cfg>(a, h->conv_pack, s);
}

// dispatch on (taps, stride, output width); wout in {128, 64, 32, 16, 8}
static hipError_t conv_any(Handle* h, const char* role, int taps, int stride, int wout, const ConvArgs& a, hipStream_t s) {
    // opt-in modes: 3x3 stride-1 convs with a multiple of 128 output channels at widths 64 / 32 / 16 run on the bf16 matrix cores
    if (h->precision != EGOTAP_PREC_F32 && h->conv_pack && taps == 9 && stride == 1 && a.Cout >= 128) {
        const bool x3 = h->precision == EGOTAP_PREC_BF16X3;
        if (wout == 64) return x3 ? conv_bf<ConvBfCfg<6, 3>>(h, role, a, s) : conv_bf<ConvBfCfg<6, 1>>(h, role, a, s);
        if (wout == 32) return x3 ? conv_bf<ConvBfCfg<5, 3>>(h, role, a, s) : conv_bf<ConvBfCfg<5, 1>>(h, role, a, s);
        if (wout == 16) return x3 ? conv_bf<ConvBfCfg<4, 3>>(h, role, a, s) : conv_bf<ConvBfCfg<4, 1>>(h, role, a, s);
        if (wout == 8) return x3 ? conv_bf<ConvBfCfg<3, 3>>(h, role, a, s) : conv_bf<ConvBfCfg<3, 1>>(h, role, a, s);
    }
    if (h->precision != EGOTAP_PREC_F32 && h->conv_pack && taps == 9 && stride == 1 && a.Cout == 64 && wout == 64)   // ResNet layer1
        return h->precision == EGOTAP_PREC_BF16X3 ? conv_bf<ConvBfCfg<6, 3, 64>>(h, role, a, s) : conv_bf<ConvBfCfg<6, 1, 64>>(h, role, a, s);
    if (h->precision == EGOTAP_PREC_BF16 && h->conv_pack && taps == 9 && stride == 1 && a.Cout == 64 && wout == 128)   // layer1 at 512x512 RGB
        return conv_bf<ConvBfCfg<7, 1, 64>>(h, role, a, s);
    if (taps == 9 && stride == 1) {
        if (wout == 128) return a.Cout <= 64 ? conv<C3s1_128_co64>(h, role, a, s) : conv<C3s1_128>(h, role, a, s);
        if (wout == 64) return a.Cout <= 64 ? conv<C3s1_64_co64>(h, role, a, s) : conv<C3s1_64>(h, role, a, s);
        if (wout == 32) return conv<C3s1_32>(h, role, a, s);
        // layer3 / layer4 at a 32-frame training batch: 128 / 64 workgroups of 128-channel tiles for 256 CUs (MFMA busy 0.21 on the 8 x 8 maps)
        const long co128 = (a.Cout + 127) / 128;
        if (wout == 16) return (long)a.Nimg * co128 < device_cu_count() ? conv<C3s1_16_co64>(h, role, a, s) : conv<C3s1_16>(h, role, a, s);
        if (wout == 8) return (long)((a.Nimg + 3) / 4) * co128 < device_cu_count() ? conv<C3s1_8_co64>(h, role, a, s) : conv<C3s1_8>(h, role, a, s);
    } else if (taps == 9 && stride == 2) {
        if (wout == 64) return conv<C3s2_64>(h, role, a, s);
        if (wout == 32) return conv<C3s2_32>(h, role, a, s);
        if (wout == 16) return conv<C3s2_16>(h, role, a, s);
        if (wout == 8) return conv<C3s2_8>(h, role, a, s);
    } else if (taps == 1 && stride == 1) {
        if (wout == 128) return conv<C1s1_128>(h, role, a, s);
        if (wout == 64) return conv<C1s1_64>(h, role, a, s);
        if (wout == 32) return conv<C1s1_32>(h, role, a, s);
        if (wout == 16) return conv<C1s1_16>(h, role, a, s);
        if (wout == 8) return conv<C1s1_8>(h, role, a, s);
    } else if (taps == 1 && stride == 2) {
        if (wout == 64) return conv<C1s2_64>(h, role, a, s);
        if (wout == 32) return conv<C1s2_32>(h, role, a, s);
        if (wout == 16) return conv<C1s2_16>(h, role, a, s);
        if (wout == 8) return conv<C1s2_8>(h, role, a, s);
    }
    return hipErrorInvalidValue;
}

struct HmWs {
    size_t L0, P0, S[4][4] /* per stage: Ta, Tb, Td, L */, U4, CAT3, X3, CAT2, X2, CAT1, X1, WPACK, WALL, total;
};
// [r3] every convolution's packed bf16 weights (+ padded bias) and every BatchNorm's folded scale / shift of one estimator, in the order
// the bf16 forward uses them; p == nullptr: sizes only.  Returns the bytes of the region; fills T (segment table of pack_all_bf16s_kernel).
// conv_heatmap's GEMM N: 30 / 34 / 60 / 68 heatmap channels padded to the 64- or 128-column tile (256 before: 4-8 x the MFMAs)
static inline int hm_head_np(int n_out) { return n_out <= 64 ? 64 : n_out <= 128 ? 128 : 256; }
// split count of a decoder convolution at serving batches: gemm_bf16s_ksplit's, capped at 8 so that every small batch lands on the SAME count
// (a frame's heatmaps then do not depend on the batch it arrives in, bit for bit -- the cap, not the batch, decides below ~8 frames)
static inline int hm_dec_ksplit(int M, int N, int K, int cus, size_t floats) {
    int sp = gemm_bf16s_ksplit(M, N, K, cus, floats);
    if (sp > 8) { sp = 8; while (sp > 1 && K % (sp * 32) != 0) --sp; }
    return sp;
}
static constexpr int HM_CAT3P = 1600;      // channels per pixel of the first decoder concat in the bf16 mode: 1024 + 516 = 1540, padded to a multiple of 64
// n2 / sides / cus / split_floats (forward only; the size query leaves them 0): image count, map side per stage and the split-K budget, from
// which the plan decides per BasicBlock 3x3 convolution whether it runs on the 64-deep GEMM (64-channel weight slabs) -- Cin a multiple of 64, Cout = 128
// (layer2, on the 256 x 128 tile) or a multiple of 256 (layer3 / layer4), and enough pixels that the 32-deep kernel's split-K path is not taken.  Same bytes either way.
static size_t hm_pack_plan(const HmParams* p, const int* nblk, PackTable* T, long n2 = 0, const int* sides = nullptr, int cus = 0, size_t split_floats = 0,
                           size_t dec_split_floats = 0, long n2_dec = -1) {
    if (n2_dec < 0) n2_dec = n2;            // [r5] the batch-statistics forward runs the backbone over the whole batch and the decoder in chunks
    size_t o = 0;
    int nw = 0, nb = 0, blk = 0;
    auto al = [&](size_t n) { size_t r = o; o = (o + n + 255) & ~(size_t)255; return r; };
    bool in_range = true;            // the table narrows offsets to 32 bits and channel counts to 16: checked here, once
    auto wseg = [&](const float* w, const float* b, int Cout, int Cin, int Cp, int Np, int taps, int slab = 32) {
        PackSeg sg{};
        sg.w = w; sg.b = b; sg.Cout = Cout; sg.Cin = Cin; sg.Cp = Cp; sg.Np = Np; sg.taps = (short)taps; sg.slab = (short)slab;
        const size_t ow = al((size_t)Np * taps * Cp * 2), ob = al((size_t)Np * 4);
        in_range = in_range && o < ((size_t)1 << 32) && Cout < 65536 && Cin < 65536 && Cp < 65536 && Np < 65536 && Cp % slab == 0;
        sg.dst_w = (unsigned)ow;
        sg.dst_b = (unsigned)ob;
        sg.first_block = blk;
        const long items = taps == 9 ? (long)Np * (Cp / 8) : (long)Np * (Cin / 8);       // 3x3: one thread per (co, 8 input channels), all taps
        blk += (int)((items + 255) / 256);
        if (T && nw < PackTable::MAXW) T->w[nw] = sg;
        ++nw;
    };
    struct Pending { HmParams::Bn bn; int C, Np; } pend[PackTable::MAXB];
    auto bseg = [&](const HmParams::Bn& bn, int C, int Np) { if (nb < PackTable::MAXB) pend[nb] = Pending{bn, C, Np}; ++nb; };
    static const HmParams::Bn nobn{nullptr, nullptr, nullptr, nullptr, nullptr};
    auto npad = [](int c) { return c <= 64 ? 64 : c <= 128 ? 128 : (c + 255) / 256 * 256; };
    int cin = 64;
    for (int i = 0; i < 4; ++i) {
        const int c = HM_CH[i];
        for (int bk = 0; bk < nblk[i]; ++bk) {
            const int bc = bk == 0 ? cin : c;
            const bool down = p ? p->blk[i][bk].wd != nullptr : (bk == 0 && i > 0);
            auto deep = [&](int K) { return sides && n2 > 0 && g_gemm_bf16s_bk != 32 && (c % 256 == 0 || c == 128) && gemm_bf16s_ksplit((int)(n2 * sides[i] * sides[i]), c, K, cus, split_floats) == 1; };
            // (conv1 of a stage's first block is the stride-2 one with Cin = Cout / 2: a multiple of 64 from layer2 on, 9 Cin >= 128)
            wseg(p ? p->blk[i][bk].w1 : nullptr, nullptr, c, bc, bc, npad(c), 9, bc % 64 == 0 && deep(9 * bc) ? 64 : 32); bseg(p ? p->blk[i][bk].bn1 : nobn, c, npad(c));
            if (down) { wseg(p ? p->blk[i][bk].wd : nullptr, nullptr, c, bc, bc, npad(c), 1); bseg(p ? p->blk[i][bk].bnd : nobn, c, npad(c)); }
            wseg(p ? p->blk[i][bk].w2 : nullptr, nullptr, c, c, c, npad(c), 9, deep(9 * c) ? 64 : 32); bseg(p ? p->blk[i][bk].bn2 : nobn, c, npad(c));
        }
        cin = c;
    }
    auto cv = [&](int which, int k) -> HmParams::Cv { return p ? (which == 0 ? p->l1x1[k] : which == 1 ? p->up[k] : p->head) : HmParams::Cv{nullptr, nullptr}; };
    wseg(cv(0, 3).w, cv(0, 3).b, 1024, 1024, 1024, 1024, 1);
    wseg(cv(0, 2).w, cv(0, 2).b, 516, 512, 512, 768, 1);
    // the three 3x3 decoder convolutions run on the 64-deep GEMM (gemm_bf16s64.h, X64Conv3): concat widths padded to a multiple of 64
    // (1540 -> HM_CAT3P = 1600), weights packed in 64-channel slabs
    // [r4] ... unless the map has so few pixels (serving batches: conv_up3 at B = 1 is one row tile x four column tiles walking K = 14400) that the
    // product is split over K: that path is the 32-deep kernel's (XConv3, 32-channel slabs), partial sums in the free head of the WPACK region
    auto dec_slab = [&](int k, int Cout, int Cp) {      // k: 2 = conv_up3 (side s16), 1 = conv_up2 (s32), 0 = conv_up1 (s64); sides[] = {s64, s32, s16, s8}
        if (!sides || n2_dec <= 0) return 64;
        if (g_gemm_bf16s_bk == 32) return 32;
        return hm_dec_ksplit((int)(n2_dec / 2 * sides[k] * sides[k]), Cout, 9 * Cp, cus, dec_split_floats) > 1 ? 32 : 64;
    };
    wseg(cv(1, 2).w, nullptr, 1024, 1540, HM_CAT3P, 1024, 9, dec_slab(2, 1024, HM_CAT3P));
    wseg(cv(0, 1).w, cv(0, 1).b, 256, 256, 256, 256, 1);
    wseg(cv(1, 1).w, nullptr, 512, 1280, 1280, 512, 9, dec_slab(1, 512, 1280));
    wseg(cv(0, 0).w, cv(0, 0).b, 128, 128, 128, 256, 1);
    wseg(cv(1, 0).w, nullptr, 512, 640, 640, 512, 9, dec_slab(0, 512, 640));
    wseg(cv(2, 0).w, cv(2, 0).b, p ? p->n_out : 30, 512, 512, p ? hm_head_np(p->n_out) : 256, 1);      // (sizes-only: the largest padding)
    const int blocks_w = blk;
    for (int k = 0; k < nb && k < PackTable::MAXB; ++k) {
        BnSeg bs{};
        bs.g = pend[k].bn.g; bs.b = pend[k].bn.b; bs.m = pend[k].bn.m; bs.v = pend[k].bn.v;
        bs.C = pend[k].C; bs.Np = pend[k].Np;
        bs.dst_sc = al((size_t)bs.Np * 4);
        bs.dst_sh = al((size_t)bs.Np * 4);
        bs.first_block = blk;
        blk += (bs.Np + 255) / 256;
        if (T) T->bn[k] = bs;
    }
    if (T) { T->nw = nw; T->nb = nb; T->blocks_w = blocks_w; T->blocks = blk; }
    return (nw <= PackTable::MAXW && nb <= PackTable::MAXB && in_range && o < ((size_t)1 << 32)) ? o : 0;
}
static HmWs hm_ws(const Handle* h, int B) {
    HmWs w;
    const size_t N2 = 2 * (size_t)B, S0 = (size_t)h->cfg.hm_size * 4;   // RGB side
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = al256(o + floats * 4); return r; };
    w.L0 = take(N2 * 64 * (S0 / 2) * (S0 / 2));
    w.P0 = take(N2 * 64 * (S0 / 4) * (S0 / 4));
    for (int i = 0; i < 4; ++i) {
        const size_t side = S0 / (4u << i), fl = N2 * HM_CH[i] * side * side;
        for (int k = 0; k < 4; ++k) w.S[i][k] = take(fl);
    }
    const size_t s8 = S0 / 32, s16 = S0 / 16, s32 = S0 / 8, s64 = S0 / 4;
    w.U4 = take((size_t)B * 1024 * s8 * s8);
    w.CAT3 = take((size_t)B * 1540 * s16 * s16); w.X3 = take((size_t)B * 1024 * s16 * s16);
    w.CAT2 = take((size_t)B * 1280 * s32 * s32); w.X2 = take((size_t)B * 512 * s32 * s32);
    w.CAT1 = take((size_t)B * 640 * s64 * s64);  w.X1 = take((size_t)B * 512 * s64 * s64);
    w.WPACK = take(conv_bf16_pack_bytes(1024, 1540) / 4);      // largest conv (conv_up3) repacked for the bf16 kernels
    const int nblk_[4] = {hm_nblk(h, 0), hm_nblk(h, 1), hm_nblk(h, 2), hm_nblk(h, 3)};
    w.WALL = take(hm_pack_plan(nullptr, nblk_, nullptr) / 4 + 64);    // [r3] bf16 mode: every layer's packed weights at once (one pack launch per forward)
    w.total = o;
    return w;
}

// ---- [r5] the bf16 channels-last forward as two pieces -- backbone (stem + four stages) and decoder -- so that the batch-statistics forward of the FROZEN
// estimators (egotap_hm_forward_bnbatch: train.py:91) can run the backbone over the whole batch (its BatchNorm couples the frames of a batch) and the
// decoder, which has no BatchNorm, in chunks.  The eval-mode egotap_hm_forward calls both with one set of buffers: same launches, same bits as before.
struct HmBf16Bufs {
    __bf16 *P0, *A[4], *Ta[4], *Tb[4], *Td[4];      // backbone: pooled stem, per stage the level (= pyramid level = decoder operand) and three temporaries
    __bf16 *T4, *C3, *Y3, *C2, *Y2, *C1, *Y1;        // decoder
    __bf16* ZP;                                     // 256 bytes of zeros
    char* reg;                                      // packed weights / folded BatchNorms (pack_all_bf16s_kernel)
    float* split_slab; size_t split_floats;         // split-K partials of the backbone's few-pixel layers
    float* dec_slab; size_t dec_split_floats;       // ... of the decoder's 3x3 convolutions at serving batches
    const float *ones, *zeros;                      // batch-statistics mode: unit scale / zero shift (1024 floats each) for the raw-output epilogue
};
struct HmBnBatch {                                  // batch-statistics mode: the scratch every BatchNorm of the forward reuses (stream-ordered)
    BnBatchScratch scr;
};
static inline int hm_ilog2(long v) { int l = 0; while ((1L << l) < v) ++l; return l; }
// [r5] the zero page of a convolution operand addressed from a scalar origin (gemm_bf16s64.h, X64ConvES / X64Conv3S) has to lie within 4 GB of the map:
// 256 bytes of zeros right behind every map a 3x3 convolution reads (in the eval-mode workspace the unused upper half of the map's fp32 slot; in the
// batch-statistics workspace 256 bytes taken with the map), written by one launch at the head of the backbone / of each decoder pass.
static inline __bf16* hm_zero_tail(const __bf16* map, size_t bytes) { return (__bf16*)((char*)map + ((bytes + 255) & ~(size_t)255)); }
struct ZeroPages { void* p[24]; };
static __global__ __launch_bounds__(64) void zero_pages_kernel(ZeroPages z) { ((float*)z.p[blockIdx.x])[threadIdx.x] = 0.f; }
static hipError_t hm_bf16_backbone(Handle* h, const HmParams& p, const PackTable& PT, int& li, int& bi, const HmBf16Bufs& q, const float* left, const float* right, int B,
                                   int S0, const HmBnBatch* bnb, hipStream_t s) {
    const int N2 = 2 * B, cus = device_cu_count();
    const int s64 = S0 / 4, s32 = S0 / 8, s16 = S0 / 16, s8 = S0 / 32;
    char* reg = q.reg;
    __bf16* ZP = q.ZP;
    {
        ZeroPages z{};
        int n = 0;
        z.p[n++] = hm_zero_tail(q.P0, (size_t)B * s64 * s64 * 128 * 2);
        for (int i = 0; i < 4; ++i) {
            const size_t side = (size_t)S0 / (4u << i), bytes = (size_t)B * side * side * 2 * HM_CH[i] * 2;
            for (__bf16* m : {q.A[i], q.Ta[i], q.Tb[i], q.Td[i]}) z.p[n++] = hm_zero_tail(m, bytes);
        }
        hipLaunchKernelGGL(zero_pages_kernel, dim3(n), dim3(64), 0, s, z);
    }
    // E1 + E2: stem conv7x7/2 + BN + ReLU + max-pool in one kernel (stem_bf16s.h): bf16 [B * s64^2, 2 x 64], image n = 2b + eye in the eye's column half
    if (!bnb) {
        hipError_t e = stem_pool_bf16s_launch(left, right, p.stem_w, p.stem_bn.g, p.stem_bn.b, p.stem_bn.m, p.stem_bn.v, q.P0, S0, N2, cus, s);
        if (e != hipSuccess) return e;
    } else {
        // batch statistics: the convolution twice -- a statistics-only pass (the 128 x 128 x 64 map still never reaches HBM), then the fused pass with per-eye tables
        int grid = 0;
        hipError_t e = stem_pool_bf16s_launch_mode<1>(left, right, p.stem_w, nullptr, nullptr, nullptr, nullptr, (__bf16*)bnb->scr.part, S0, N2, cus, s, &grid);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bn_finish_bf16s_kernel, dim3(64), dim3(128), 0, s, (const float*)bnb->scr.part, grid, 64, (double)B * (S0 / 2) * (S0 / 2), p.stem_bn.g, p.stem_bn.b,
                           (float*)p.stem_bn.m, (float*)p.stem_bn.v, p.stem_bn.nbt, bnb->scr.sc, bnb->scr.sh);
        e = stem_pool_bf16s_launch_mode<2>(left, right, p.stem_w, bnb->scr.sc, bnb->scr.sh, nullptr, nullptr, q.P0, S0, N2, cus, s);
        if (e != hipSuccess) return e;
    }
    // E3: the four stages on the same GEMM kernel (eye-interleaved rows, see conv_bf16s.h)
    auto bconv = [&](const char* role, const __bf16* in, int cin, int c, int taps, int stride, int side, const float* wgt,
                     const HmParams::Bn& bn, const __bf16* res_, int relu_, __bf16* o) -> hipError_t {
        // [r3] Cout = 64 / 128 run on the 64- / 128-column tile (256 x 64 NI, gemm_bf16s.h) instead of N = 256 with a column guard
        const int Np = c <= 64 ? 64 : c <= 128 ? 128 : (c + 255) / 256 * 256;
        const PackSeg& sg = PT.w[li++];
        const BnSeg& bs = PT.bn[bi++];
        if (sg.w != wgt || sg.Np != Np || sg.taps != taps || bs.g != bn.g) return hipErrorInvalidValue;
        const __bf16* WPl = (const __bf16*)(reg + sg.dst_w);
        // batch statistics: the convolution stores its raw bf16 output (unit scale, zero shift, no residual, no ReLU); bn_batch below does the rest in place
        const float *SC = bnb ? q.ones : (const float*)(reg + bs.dst_sc), *SH = bnb ? q.zeros : (const float*)(reg + bs.dst_sh);
        const __bf16* res = bnb ? nullptr : res_;
        const int relu = bnb ? 0 : relu_;
        const long M = (long)N2 * side * side;
        auto run = [&]() -> hipError_t {
            const bool direct = taps == 9 && stride == 1 && cin == 64 && c == 64;      // [r3] layer1: the direct kernel (conv64_bf16s.h)
            GemmTimer t(h, s, role, direct ? "conv64_direct_bf16s_kernel" : taps == 9 ? "gemm_bf16s_kernel<XConvE,3x3>" : "gemm_bf16s_kernel<XConvE,1x1>",
                        2.0 * M * c * taps * (double)cin);
            if (direct)
                return conv64_direct_bf16s_launch(in, ZP, WPl, SC, SH, res, o, hm_ilog2(side), N2, relu, cus, s);
            const SEpiBnBf16<false> ep{SC, SH, res, o, c, hm_ilog2(side), relu};
            if (sg.slab == 64) {      // [r4] layer3 / layer4's stride-1 convolutions on the 64-deep GEMM (the plan checked the shape rules)
                if (taps != 9 || cin % 64 != 0 || Np != c) return hipErrorInvalidValue;
                const long si = (long)side * stride;
                const __bf16* org;
                unsigned in_off, zero_off;
                const size_t in_bytes = (size_t)(N2 / 2) * si * si * 2 * cin * 2;
                if (s64_conv_origin(in, in_bytes, (size_t)(si + 1) * 2 * cin * 2, hm_zero_tail(in, in_bytes), org, in_off, zero_off)) {   // [r5] scalar origin + lane offsets
                    const X64ConvES xs{org, in_off, zero_off, cin, hm_ilog2(side), stride};
                    if (c == 128) return gemm_bf16s64_launch_x<X64ConvES, SEpiBnBf16<false>, 1>(xs, WPl, 9L * cin, ep, (int)M, Np, 9 * cin, cus, s);
                    return gemm_bf16s64_launch_x(xs, WPl, 9L * cin, ep, (int)M, Np, 9 * cin, cus, s);
                }
                const X64ConvE xl{in, ZP, cin, hm_ilog2(side), stride};
                if (c == 128) return gemm_bf16s64_launch_x<X64ConvE, SEpiBnBf16<false>, 1>(xl, WPl, 9L * cin, ep, (int)M, Np, 9 * cin, cus, s);   // layer2: 256 x 128 tile
                return gemm_bf16s64_launch_x(xl, WPl, 9L * cin, ep, (int)M, Np, 9 * cin, cus, s);
            }
            const XConvE xl{in, ZP, cin, hm_ilog2(side), stride, taps};
            if (Np == 64 && c == 64) return gemm_bf16s_launch<XConvE, SEpiBnBf16<false>, 1>(xl, WPl, (long)taps * cin, ep, (int)M, Np, taps * cin, cus, s);
            if (Np == 128 && c == 128) return gemm_bf16s_launch<XConvE, SEpiBnBf16<false>, 2>(xl, WPl, (long)taps * cin, ep, (int)M, Np, taps * cin, cus, s);
            if (Np == c) {
                // [r3] few pixels (layer3 / layer4 at small batches: 512 x 512 RGB at B = 32 leaves layer4 128 tiles for 256 CUs): split K,
                // partial sums in the (by now free) slot of the stem's map, BatchNorm / residual / ReLU in the fixed-order reduce
                const int sp = gemm_bf16s_ksplit((int)M, Np, taps * cin, cus, q.split_floats);
                if (sp > 1) return gemm_bf16s_splitk_launch(xl, WPl, (long)taps * cin, ep, q.split_slab, sp, (int)M, Np, taps * cin, cus, s);
                return gemm_bf16s_launch(xl, WPl, (long)taps * cin, ep, (int)M, Np, taps * cin, cus, s);
            }
            return gemm_bf16s_launch(xl, WPl, (long)taps * cin, SEpiBnBf16<true>{SC, SH, res, o, c, hm_ilog2(side), relu}, (int)M, Np, taps * cin, cus, s);
        };
        hipError_t e = run();
        if (e != hipSuccess || !bnb) return e;
        return bn_batch_bf16s_launch(o, res_, (long)B * side * side, c, bn.g, bn.b, (float*)bn.m, (float*)bn.v, bn.nbt, relu_, bnb->scr, s);
    };
    const int sides[4] = {s64, s32, s16, s8};
    const __bf16* x = q.P0;
    int cin = 64;
    for (int i = 0; i < 4; ++i) {
        const int c = HM_CH[i], side = sides[i], nb = p.nblk[i];
        __bf16 *Ta = q.Ta[i], *Tb = q.Tb[i], *Td = q.Td[i];                       // q.A[i] is the level itself
        const __bf16* xin = x;
        for (int bk = 0; bk < nb; ++bk) {
            const auto& K = p.blk[i][bk];
            const int stride = (bk == 0 && i > 0) ? 2 : 1, bc = bk == 0 ? cin : c;
            // block outputs alternate between Tb and the level so that the last block writes the level (a block never writes
            // the buffer it reads its identity from)
            __bf16* y = ((nb - 1 - bk) & 1) ? Tb : q.A[i];
            hipError_t e = bconv(HM_R1[i][bk], xin, bc, c, 9, stride, side, K.w1, K.bn1, nullptr, 1, Ta);
            if (e != hipSuccess) return e;
            const __bf16* idt = xin;
            if (K.wd) {
                e = bconv(HM_RD[i], xin, bc, c, 1, 2, side, K.wd, K.bnd, nullptr, 0, Td);
                if (e != hipSuccess) return e;
                idt = Td;
            }
            e = bconv(HM_R2[i][bk], Ta, c, c, 9, 1, side, K.w2, K.bn2, idt, 1, y);
            if (e != hipSuccess) return e;
            xin = y;
        }
        x = q.A[i];
        cin = c;
    }
    return hipSuccess;
}

// E4-E9 on Bc frames whose pyramid levels start at lv[0..3] (layer1 .. layer4 outputs, bf16 [Bc * s^2, 2 C]); li = first decoder entry of the pack table
static hipError_t hm_bf16_decoder(Handle* h, const HmParams& p, const PackTable& PT, int li, const HmBf16Bufs& q, const __bf16* const* lv, int B, int S0, float* out,
                                  int64_t out_image_stride, hipStream_t s) {
    const int cus = device_cu_count();
    const int s64 = S0 / 4, s32 = S0 / 8, s16 = S0 / 16, s8 = S0 / 32;
    const long p8 = (long)s8 * s8, p16 = (long)s16 * s16, p32 = (long)s32 * s32, p64 = (long)s64 * s64;
    char* reg = q.reg;
    __bf16* ZP = q.ZP;
    __bf16 *T4 = q.T4, *C3 = q.C3, *Y3 = q.Y3, *C2 = q.C2, *Y2 = q.Y2, *C1 = q.C1, *Y1 = q.Y1;
    auto up2 = [&](const __bf16* in, __bf16* o, int C, int hin, long ld) {
        const long total = (long)B * 4 * hin * hin * (C / 8);
        hipLaunchKernelGGL(upsample2x_nhwc_bf16s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, o, C, hin, ld, total);
        return hipGetLastError();
    };
    // 1x1 convrelu: rows = pixels; the output goes to columns [0, Cout) of o (row stride ld)
    auto conv1 = [&](const char* role, const __bf16* in, long M, const HmParams::Cv& cv, int Cin, int Cout, __bf16* o, long ld) {
        const int Np = (Cout + 255) / 256 * 256;
        const PackSeg& sg = PT.w[li++];
        if (sg.w != cv.w || sg.Np != Np || sg.Cin != Cin) return hipErrorInvalidValue;      // the plan and the forward walk the layers in one order
        const __bf16* WPl = (const __bf16*)(reg + sg.dst_w);
        const float* BPl = (const float*)(reg + sg.dst_b);
        GemmTimer t(h, s, role, "gemm_bf16s_kernel<XPlain,conv1x1>", 2.0 * M * Cout * Cin);
        if (Np == Cout) return gemm_bf16s_launch(XPlain{in, Cin}, WPl, (long)Cin, SEpiConvBf16<false>{BPl, o, ld, Np, 1}, (int)M, Np, Cin, cus, s);
        return gemm_bf16s_launch(XPlain{in, Cin}, WPl, (long)Cin, SEpiConvBf16<true>{BPl, o, ld, (Cout + 7) / 8 * 8, 1}, (int)M, Np, Cin, cus, s);
    };
    // 3x3 convrelu on a concat buffer of Cp channels per pixel (Cp a multiple of 32; channels past Cin are zero)
    auto conv3 = [&](const char* role, const __bf16* in, long M, int side, const HmParams::Cv& cv, int Cin, int Cp, int Cout, __bf16* o) {
        const PackSeg& sg = PT.w[li++];
        if (sg.w != cv.w || sg.Cp != Cp || sg.Np != Cout) return hipErrorInvalidValue;
        if (sg.slab == 32) {      // [r4] few pixels: the 32-deep kernel, split over K when the plan's rule says so (serving batches)
            if (Cp % 32 != 0 || Cout % 256 != 0) return hipErrorInvalidValue;
            GemmTimer t(h, s, role, "gemm_bf16s_kernel<XConv3>", 2.0 * M * Cout * 9.0 * Cin);
            const XConv3 xl{in, ZP, Cp, hm_ilog2(side)};
            const SEpiConvBf16<false> ep{cv.b, o, (long)Cout, Cout, 1};
            const int sp = hm_dec_ksplit((int)M, Cout, 9 * Cp, cus, q.dec_split_floats);
            if (sp > 1) return gemm_bf16s_splitk_launch(xl, (const __bf16*)(reg + sg.dst_w), 9L * Cp, ep, q.dec_slab, sp, (int)M, Cout, 9 * Cp, cus, s);
            return gemm_bf16s_launch(xl, (const __bf16*)(reg + sg.dst_w), 9L * Cp, ep, (int)M, Cout, 9 * Cp, cus, s);
        }
        if (sg.slab != 64 || Cp % 64 != 0 || Cout % 256 != 0) return hipErrorInvalidValue;
        GemmTimer t(h, s, role, "gemm_bf16s64_kernel<X64Conv3>", 2.0 * M * Cout * 9.0 * Cin);
        const SEpiConvBf16<false> ep{cv.b, o, (long)Cout, Cout, 1};
        const __bf16* org;
        unsigned in_off, zero_off;
        if (s64_conv_origin(in, (size_t)M * Cp * 2, (size_t)(side + 1) * Cp * 2, hm_zero_tail(in, (size_t)M * Cp * 2), org, in_off, zero_off))      // [r5] scalar origin + lane offsets
            return gemm_bf16s64_launch_x(X64Conv3S{org, in_off, zero_off, Cp, hm_ilog2(side)}, (const __bf16*)(reg + sg.dst_w), 9L * Cp, ep, (int)M, Cout, 9 * Cp, cus, s);
        const X64Conv3 xl{in, ZP, Cp, hm_ilog2(side)};
        return gemm_bf16s64_launch_x(xl, (const __bf16*)(reg + sg.dst_w), 9L * Cp, ep, (int)M, Cout, 9 * Cp, cus, s);
    };
    const __bf16 *A1 = lv[0], *A2 = lv[1], *A3 = lv[2], *A4 = lv[3];
#define HMD(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
    {
        ZeroPages z{};
        z.p[0] = hm_zero_tail(C3, (size_t)B * p16 * HM_CAT3P * 2);
        z.p[1] = hm_zero_tail(C2, (size_t)B * p32 * 1280 * 2);
        z.p[2] = hm_zero_tail(C1, (size_t)B * p64 * 640 * 2);
        hipLaunchKernelGGL(zero_pages_kernel, dim3(3), dim3(64), 0, s, z);
    }
    HMD(zero_fill(C3, (size_t)B * p16 * HM_CAT3P * 2, s));                    // channels 1544..1599 of the first concat are padding
    HMD(conv1("hm.layer4_1x1", A4, B * p8, p.l1x1[3], 1024, 1024, T4, 1024));
    HMD(up2(T4, C3, 1024, s8, HM_CAT3P));
    HMD(conv1("hm.layer3_1x1", A3, B * p16, p.l1x1[2], 512, 516, C3 + 1024, HM_CAT3P));
    HMD(conv3("hm.conv_up3", C3, B * p16, s16, p.up[2], 1540, HM_CAT3P, 1024, Y3));
    HMD(up2(Y3, C2, 1024, s16, 1280));
    HMD(conv1("hm.layer2_1x1", A2, B * p32, p.l1x1[1], 256, 256, C2 + 1024, 1280));
    HMD(conv3("hm.conv_up2", C2, B * p32, s32, p.up[1], 1280, 1280, 512, Y2));
    HMD(up2(Y2, C1, 512, s32, 640));
    HMD(conv1("hm.layer1_1x1", A1, B * p64, p.l1x1[0], 128, 128, C1 + 512, 640));
    HMD(conv3("hm.conv_up1", C1, B * p64, s64, p.up[0], 640, 640, 512, Y1));
    {   // conv_heatmap: fp32 NCHW into the caller's channel slice
        const PackSeg& sg = PT.w[li++];
        if (sg.w != p.head.w || li != PT.nw) return hipErrorInvalidValue;      // the pack plan is out of step with the forward
        GemmTimer t(h, s, "hm.conv_heatmap", "gemm_bf16s_kernel<XPlain,heatmap>", 2.0 * B * p64 * p.n_out * 512);
        const SEpiHeatNCHW he{(const float*)(reg + sg.dst_b), out, (long)out_image_stride, p.n_out, hm_ilog2(p64)};
        const __bf16* hw = (const __bf16*)(reg + sg.dst_w);
        const int hn = hm_head_np(p.n_out);
        if (hn == 64) HMD((gemm_bf16s_launch<XPlain, SEpiHeatNCHW, 1>(XPlain{Y1, 512}, hw, 512L, he, (int)(B * p64), 64, 512, cus, s)));
        else if (hn == 128) HMD((gemm_bf16s_launch<XPlain, SEpiHeatNCHW, 2>(XPlain{Y1, 512}, hw, 512L, he, (int)(B * p64), 128, 512, cus, s)));
        else HMD(gemm_bf16s_launch(XPlain{Y1, 512}, hw, 512L, he, (int)(B * p64), 256, 512, cus, s));
    }
#undef HMD
    return hipSuccess;
}

#if EGOTAP_IN(0)
extern "C" int egotap_hm_workspace_bytes(egotap_handle h, int B, size_t* bytes) {
    EGO_CHECK(h && bytes, "null argument");
    EGO_CHECK(B >= 0, "negative batch");
    *bytes = hm_ws(h, B > 0 ? B : 1).total;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_hm_intermediate(egotap_handle h, int B, const char* name, size_t* offset, int64_t* numel) {
    EGO_CHECK(h && name && offset && numel, "null argument");
    const HmWs w = hm_ws(h, B > 0 ? B : 1);
    const int64_t N2 = 2LL * B, S0 = h->cfg.hm_size * 4;
    auto sq = [](int64_t v) { return v * v; };
    if (!strcmp(name, "layer0")) { *offset = w.L0; *numel = N2 * 64 * sq(S0 / 2); }
    else if (!strcmp(name, "pool0")) { *offset = w.P0; *numel = N2 * 64 * sq(S0 / 4); }      // stem + max-pool: fp32 NCHW, or (bf16 mode) bf16 [B * (S0/4)^2, 2 x 64]
    else if (!strcmp(name, "layer1_bf16")) { *offset = w.S[0][0]; *numel = N2 * 64 * sq(S0 / 4); }      // bf16 mode: bf16 [B * (S0/4)^2, 2 x 64]
    else if (!strcmp(name, "layer1")) { *offset = w.S[0][3]; *numel = N2 * 64 * sq(S0 / 4); }
    else if (!strcmp(name, "layer2")) { *offset = w.S[1][3]; *numel = N2 * 128 * sq(S0 / 8); }
    else if (!strcmp(name, "layer3")) { *offset = w.S[2][3]; *numel = N2 * 256 * sq(S0 / 16); }
    else if (!strcmp(name, "layer4")) { *offset = w.S[3][3]; *numel = N2 * 512 * sq(S0 / 32); }
    else if (!strcmp(name, "u4")) { *offset = w.U4; *numel = (int64_t)B * 1024 * sq(S0 / 32); }
    else if (!strcmp(name, "cat3")) { *offset = w.CAT3; *numel = (int64_t)B * 1540 * sq(S0 / 16); }
    else if (!strcmp(name, "cat2")) { *offset = w.CAT2; *numel = (int64_t)B * 1280 * sq(S0 / 8); }
    else if (!strcmp(name, "cat1")) { *offset = w.CAT1; *numel = (int64_t)B * 640 * sq(S0 / 4); }
    else if (!strcmp(name, "conv_up3")) { *offset = w.X3; *numel = (int64_t)B * 1024 * sq(S0 / 16); }
    else if (!strcmp(name, "conv_up2")) { *offset = w.X2; *numel = (int64_t)B * 512 * sq(S0 / 8); }
    else if (!strcmp(name, "conv_up1")) { *offset = w.X1; *numel = (int64_t)B * 512 * sq(S0 / 4); }
    else { egotap_set_error("unknown intermediate '%s'", name); return EGOTAP_ERR_INVALID; }
    return EGOTAP_OK;
}
#endif

// HeatMap_UnrealEgo_Shared.forward(left, right) (model/net_architecture.py:32-36, 45-51, 75-85, 139-173), eval mode.
#if EGOTAP_IN(0)
extern "C" int egotap_hm_forward(egotap_handle h, int net, const float* left, const float* right, int B, float* out,
                                 int64_t out_image_stride, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(net == EGOTAP_NET_HM_POS || net == EGOTAP_NET_HM_ROT, "egotap_hm_forward: net must be EGOTAP_NET_HM_POS or _ROT");
    if (B == 0) return EGOTAP_OK;
    EGO_CHECK(B > 0 && left && right && out && ws, "egotap_hm_forward: null argument or negative batch");
    EGO_CHECK((((uintptr_t)left | (uintptr_t)right | (uintptr_t)out) & 15) == 0 && ((uintptr_t)ws & 255) == 0, "pointers must be 16-byte (ws: 256-byte) aligned");
    const int S0 = h->cfg.hm_size * 4, s64 = S0 / 4, s32 = S0 / 8, s16 = S0 / 16, s8 = S0 / 32;
    EGO_CHECK(s64 == 64 || s64 == 128, "this build instantiates the conv kernels for 256x256 and 512x512 RGB (64x64 / 128x128 heatmaps)");
    int rc = hm_resolve(h, net);
    if (rc != EGOTAP_OK) return rc;
    const HmWs w = hm_ws(h, B);
    if (ws_bytes < w.total) {
        egotap_set_error("workspace too small: %zu bytes given, %zu needed for B=%d", ws_bytes, w.total, B);
        return EGOTAP_ERR_WORKSPACE;
    }
    const HmParams& p = h->hp[net];
    EGO_CHECK(out_image_stride >= (int64_t)p.n_out * s64 * s64, "out_image_stride smaller than the output image");
    h->conv_pack = (__bf16*)((char*)ws + w.WPACK);
    // [r4] exact-fp32 mode: the two weight-pack regions at the end of the workspace are unused -> scratch of the serving-batch channel split (conv_f32.h)
    struct PartGuard { Handle* h; ~PartGuard() { h->conv_part = nullptr; h->conv_part_floats = 0; } } part_guard{h};
    if (h->precision == EGOTAP_PREC_F32) {
        h->conv_part = (float*)((char*)ws + w.WPACK);
        h->conv_part_floats = (w.total - w.WPACK) / 4;
    }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    auto F = [&](size_t off) { return (float*)(base + off); };
    const int N2 = 2 * B;

    // E1: stem conv7x7/2 + BN + ReLU on image n = 2b + eye (the L/R channel concat of every pyramid level is then a view); in the
    // bf16 mode it writes bf16 channels-last itself (half the bytes, and the layout the max-pool and the stages read)
    // [r3] ... bf16 mode: stem, BatchNorm, ReLU AND the max-pool in one kernel on the bf16 matrix cores (stem_bf16s.h): the 128 x 128 x 64
    // map never reaches HBM (round 2's two-kernel form -- fp32-MFMA stem writing bf16 channels-last, then a channels-last max-pool: 2.35 ms
    // against 0.66 per 512 images -- was retired in round 4).
    const bool fused_stem = h->precision == EGOTAP_PREC_BF16;
    if (!fused_stem)
        EGO_HIP(stem_conv7_launch(left, right, p.stem_w, p.stem_bn.g, p.stem_bn.b, p.stem_bn.m, p.stem_bn.v, F(w.L0), S0, N2, device_cu_count(), s));
    if (h->precision == EGOTAP_PREC_BF16) {
        // bf16 mode: everything after the stem on bf16 channels-last activations, every convolution on the bf16-storage GEMM
        // (conv_bf16s.h).  Buffers live in the fp32 path's slots (each at most half as large).
        auto Hb = [&](size_t off) { return (__bf16*)(base + off); };
        HmBf16Bufs q{};
        q.P0 = Hb(w.P0);
        for (int i = 0; i < 4; ++i) { q.A[i] = Hb(w.S[i][0]); q.Ta[i] = Hb(w.S[i][1]); q.Tb[i] = Hb(w.S[i][2]); q.Td[i] = Hb(w.S[i][3]); }
        q.T4 = Hb(w.U4); q.C3 = Hb(w.CAT3); q.Y3 = Hb(w.X3); q.C2 = Hb(w.CAT2); q.Y2 = Hb(w.X2); q.C1 = Hb(w.CAT1); q.Y1 = Hb(w.X1);
        q.ZP = (__bf16*)(base + w.WPACK + ((size_t)40 << 20) + 32768);              // 256 bytes of zeros (taps outside the image)
        q.reg = base + w.WALL;
        q.split_slab = F(w.L0);                                                    // the fp32 stem map's slot: unused (fused stem)
        q.split_floats = (size_t)N2 * 64 * (S0 / 2) * (S0 / 2);
        q.dec_slab = (float*)(base + w.WPACK);                                     // [r4] the first 40 MB of the WPACK region (the zero page sits behind them): split-K partials of the decoder
        q.dec_split_floats = ((size_t)40 << 20) / 4;
        const int cus = device_cu_count();
        EGO_HIP(zero_fill(q.ZP, 256, s));
        // [r3] all 27 weight repacks and 19 BatchNorm folds of this forward in ONE launch (conv_bf16s.h, pack_all_bf16s_kernel): the
        // parameters stay the caller's live fp32 tensors, nothing is kept between calls
        PackTable PT;
        const int stage_sides[4] = {s64, s32, s16, s8};
        EGO_CHECK(hm_pack_plan(&p, p.nblk, &PT, N2, stage_sides, cus, q.split_floats, q.dec_split_floats) != 0, "egotap_hm_forward: the estimator has more layers than the pack table holds");
        hipLaunchKernelGGL(pack_all_bf16s_kernel, dim3(PT.blocks), dim3(256), 0, s, PT, q.reg);
        EGO_HIP(hipGetLastError());
        int li = 0, bi = 0;
        EGO_HIP(hm_bf16_backbone(h, p, PT, li, bi, q, left, right, B, S0, nullptr, s));
        const __bf16* lv[4] = {q.A[0], q.A[1], q.A[2], q.A[3]};
        EGO_HIP(hm_bf16_decoder(h, p, PT, li, q, lv, B, S0, out, out_image_stride, s));
        EGO_CHECK(bi == PT.nb, "egotap_hm_forward: pack plan out of step with the forward");
        return EGOTAP_OK;
    }
    // E2: maxpool 3x3/2
    maxpool3s2_launch(F(w.L0), F(w.P0), (long)N2 * 64, S0 / 2, s);
    EGO_HIP(hipGetLastError());
    // E3: four stages of two BasicBlocks
    const float* x = F(w.P0);
    int cin = 64;
    const int sides[4] = {s64, s32, s16, s8};
    for (int i = 0; i < 4; ++i) {
        const int c = HM_CH[i], side = sides[i], nb = p.nblk[i];
        const long ist_in = (long)cin * (i == 0 ? side : 2 * side) * (i == 0 ? side : 2 * side);
        const long ist = (long)c * side * side;
        float *Ta = F(w.S[i][0]), *Tb = F(w.S[i][1]), *Td = F(w.S[i][2]), *L = F(w.S[i][3]);
        const float* xin = x;
        long xin_ist = ist_in;
        for (int b = 0; b < nb; ++b) {
            const auto& K = p.blk[i][b];
            const int stride = (b == 0 && i > 0) ? 2 : 1;
            const int bc = b == 0 ? cin : c;
            ConvArgs a1{xin, K.w1, Ta, nullptr, K.bn1.g, K.bn1.b, K.bn1.m, K.bn1.v, nullptr, xin_ist, ist, 0, N2, bc, c, 1, 0, 0};
            EGO_HIP(conv_any(h, HM_R1[i][b], 9, stride, side, a1, s));
            const float* idt = xin;
            long idt_ist = xin_ist;
            if (K.wd) {
                ConvArgs ad{xin, K.wd, Td, nullptr, K.bnd.g, K.bnd.b, K.bnd.m, K.bnd.v, nullptr, xin_ist, ist, 0, N2, bc, c, 0, 0, 0};
                EGO_HIP(conv_any(h, HM_RD[i], 1, 2, side, ad, s));
                idt = Td;
                idt_ist = ist;
            }
            float* y = ((nb - 1 - b) & 1) ? Tb : L;       // outputs alternate so that the last block writes the level
            ConvArgs a2{Ta, K.w2, y, idt, K.bn2.g, K.bn2.b, K.bn2.m, K.bn2.v, nullptr, ist, ist, idt_ist, N2, c, c, 1, 0, 0};
            EGO_HIP(conv_any(h, HM_R2[i][b], 9, 1, side, a2, s));
            xin = y;
            xin_ist = ist;
        }
        x = L;
        cin = c;
    }
    // E4-E9: decoder on the channel-concatenated pyramids ([2B, C, s, s] viewed as [B, 2C, s, s])
    const float *L1 = F(w.S[0][3]), *L2 = F(w.S[1][3]), *L3 = F(w.S[2][3]), *L4 = F(w.S[3][3]);
    float *U4 = F(w.U4), *CAT3 = F(w.CAT3), *X3 = F(w.X3), *CAT2 = F(w.CAT2), *X2 = F(w.X2), *CAT1 = F(w.CAT1), *X1 = F(w.X1);
    auto up = [&](const float* in, float* o, int C, int hin, long ist_in, long ist_out) {
        const long threads = (long)B * C * (2 * hin) * (2 * hin / 4);
        hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, in, o, B, C, hin, ist_in, ist_out);
        return hipGetLastError();
    };
    auto biasconv = [&](const char* role, int taps, int side, const float* in, long in_ist, const HmParams::Cv& cv, int Cin, int Cout,
                        float* o, long o_ist, int relu) {
        ConvArgs a{in, cv.w, o, nullptr, nullptr, nullptr, nullptr, nullptr, cv.b, in_ist, o_ist, 0, B, Cin, Cout, relu, 0, 0};
        return conv_any(h, role, taps, 1, side, a, s);
    };
    const long p8 = (long)s8 * s8, p16 = (long)s16 * s16, p32 = (long)s32 * s32, p64 = (long)s64 * s64;
    EGO_HIP(biasconv("hm.layer4_1x1", 1, s8, L4, 1024 * p8, p.l1x1[3], 1024, 1024, U4, 1024 * p8, 1));
    EGO_HIP(up(U4, CAT3, 1024, s8, 1024 * p8, 1540 * p16));
    EGO_HIP(biasconv("hm.layer3_1x1", 1, s16, L3, 512 * p16, p.l1x1[2], 512, 516, CAT3 + 1024 * p16, 1540 * p16, 1));
    EGO_HIP(biasconv("hm.conv_up3", 9, s16, CAT3, 1540 * p16, p.up[2], 1540, 1024, X3, 1024 * p16, 1));
    EGO_HIP(up(X3, CAT2, 1024, s16, 1024 * p16, 1280 * p32));
    EGO_HIP(biasconv("hm.layer2_1x1", 1, s32, L2, 256 * p32, p.l1x1[1], 256, 256, CAT2 + 1024 * p32, 1280 * p32, 1));
    EGO_HIP(biasconv("hm.conv_up2", 9, s32, CAT2, 1280 * p32, p.up[1], 1280, 512, X2, 512 * p32, 1));
    EGO_HIP(up(X2, CAT1, 512, s32, 512 * p32, 640 * p64));
    EGO_HIP(biasconv("hm.layer1_1x1", 1, s64, L1, 128 * p64, p.l1x1[0], 128, 128, CAT1 + 512 * p64, 640 * p64, 1));
    EGO_HIP(biasconv("hm.conv_up1", 9, s64, CAT1, 640 * p64, p.up[0], 640, 512, X1, 512 * p64, 1));
    EGO_HIP(biasconv("hm.conv_heatmap", 1, s64, X1, 512 * p64, p.head, 512, p.n_out, out, out_image_stride, 0));
    return EGOTAP_OK;
}
#endif

// ---- [r5] batch-statistics forward of a frozen estimator on the bf16 channels-last kernels (train.py:91; egotap_autoencoder_model.py:127-129, 177-216)
struct HmBnWs {
    size_t P0, S[4][4], T4, C3, Y3, C2, Y2, C1, Y1, WALL, DEC, ZP, SPLIT, PART, SC, SH, ONES, ZEROS, total;
};
static constexpr size_t HM_BN_SPLIT_FLOATS = (size_t)1 << 24;      // split-K partials of the backbone: tiles x splits <= CUs x 65536 floats
static HmBnWs hm_bn_ws(const Handle* h, int B, int chunk) {
    HmBnWs w;
    const size_t S0 = (size_t)h->cfg.hm_size * 4, s64 = S0 / 4, s32 = S0 / 8, s16 = S0 / 16, s8 = S0 / 32;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = al256(o + bytes); return r; };
    constexpr size_t ZT = 256;                                  // the zero page behind every map a 3x3 convolution reads (hm_zero_tail)
    w.P0 = take((size_t)B * s64 * s64 * 128 * 2 + ZT);
    for (int i = 0; i < 4; ++i) {
        const size_t side = S0 / (4u << i), bytes = (size_t)B * side * side * 2 * HM_CH[i] * 2;
        for (int k = 0; k < 4; ++k) w.S[i][k] = take(bytes + ZT);
    }
    const size_t c = (size_t)chunk;
    w.T4 = take(c * s8 * s8 * 1024 * 2);
    w.C3 = take(c * s16 * s16 * HM_CAT3P * 2 + ZT); w.Y3 = take(c * s16 * s16 * 1024 * 2);
    w.C2 = take(c * s32 * s32 * 1280 * 2 + ZT);     w.Y2 = take(c * s32 * s32 * 512 * 2);
    w.C1 = take(c * s64 * s64 * 640 * 2 + ZT);      w.Y1 = take(c * s64 * s64 * 512 * 2);
    const int nblk_[4] = {hm_nblk(h, 0), hm_nblk(h, 1), hm_nblk(h, 2), hm_nblk(h, 3)};
    w.WALL = take(hm_pack_plan(nullptr, nblk_, nullptr) + 256);
    w.DEC = take((size_t)40 << 20);
    w.ZP = take(256);
    w.SPLIT = take(HM_BN_SPLIT_FLOATS * 4);
    w.PART = take(BnBatchScratch::part_floats() * 4);
    w.SC = take(4096); w.SH = take(4096); w.ONES = take(4096); w.ZEROS = take(4096);
    w.total = o;
    return w;
}
static __global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

#if EGOTAP_IN(0)
extern "C" int egotap_hm_forward_bnbatch_workspace_bytes(egotap_handle h, int B, int chunk, size_t* bytes) {
    EGO_CHECK(h && bytes, "null argument");
    EGO_CHECK(B >= 0 && chunk >= 0, "negative batch or chunk");
    const int b = B > 0 ? B : 1;
    *bytes = hm_bn_ws(h, b, chunk > 0 && chunk < b ? chunk : b).total;
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_hm_forward_bnbatch_intermediate(egotap_handle h, int B, int chunk, const char* name, size_t* offset, int64_t* numel) {
    EGO_CHECK(h && name && offset && numel, "null argument");
    EGO_CHECK(B >= 1, "batch");
    const HmBnWs w = hm_bn_ws(h, B, chunk > 0 && chunk < B ? chunk : B);
    const int64_t S0 = h->cfg.hm_size * 4;
    auto px = [&](int i) { const int64_t side = S0 / (4 << i); return (int64_t)B * side * side; };
    if (!strcmp(name, "pool0")) { *offset = w.P0; *numel = px(0) * 128; }
    else if (!strcmp(name, "layer1")) { *offset = w.S[0][0]; *numel = px(0) * 2 * HM_CH[0]; }
    else if (!strcmp(name, "layer2")) { *offset = w.S[1][0]; *numel = px(1) * 2 * HM_CH[1]; }
    else if (!strcmp(name, "layer3")) { *offset = w.S[2][0]; *numel = px(2) * 2 * HM_CH[2]; }
    else if (!strcmp(name, "layer4")) { *offset = w.S[3][0]; *numel = px(3) * 2 * HM_CH[3]; }
    else { egotap_set_error("unknown intermediate '%s'", name); return EGOTAP_ERR_INVALID; }
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_hm_forward_bnbatch(egotap_handle h, int net, const float* left, const float* right, int B, float* out, int64_t out_image_stride, int chunk,
                                         void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(net == EGOTAP_NET_HM_POS || net == EGOTAP_NET_HM_ROT, "egotap_hm_forward_bnbatch: net must be EGOTAP_NET_HM_POS or _ROT");
    EGO_CHECK(h->precision == EGOTAP_PREC_BF16, "egotap_hm_forward_bnbatch runs on the bf16 channels-last kernels: egotap_set_precision(EGOTAP_PREC_BF16) first "
              "(fp32 / bf16x3: compose the train-mode forward from egotap_hmtrain_conv_fwd + egotap_hmtrain_bn2d_fwd)");
    if (B == 0) return EGOTAP_OK;
    EGO_CHECK(B >= 2, "batch-statistics BatchNorm needs more than one value per channel and every map has them only from two frames on at the deepest level "
              "(torch raises for one 1 x 1 map; keep B >= 2)");
    EGO_CHECK(left && right && out && ws, "egotap_hm_forward_bnbatch: null argument");
    EGO_CHECK((((uintptr_t)left | (uintptr_t)right | (uintptr_t)out) & 15) == 0 && ((uintptr_t)ws & 255) == 0, "pointers must be 16-byte (ws: 256-byte) aligned");
    const int S0 = h->cfg.hm_size * 4, s64 = S0 / 4, s32 = S0 / 8, s16 = S0 / 16, s8 = S0 / 32;
    EGO_CHECK(s64 == 64 || s64 == 128, "this build instantiates the conv kernels for 256x256 and 512x512 RGB (64x64 / 128x128 heatmaps)");
    if (chunk <= 0 || chunk > B) chunk = B;
    int rc = hm_resolve(h, net);
    if (rc != EGOTAP_OK) return rc;
    const HmBnWs w = hm_bn_ws(h, B, chunk);
    if (ws_bytes < w.total) {
        egotap_set_error("workspace too small: %zu bytes given, %zu needed for B=%d, chunk=%d", ws_bytes, w.total, B, chunk);
        return EGOTAP_ERR_WORKSPACE;
    }
    const HmParams& p = h->hp[net];
    EGO_CHECK(out_image_stride >= (int64_t)p.n_out * s64 * s64, "out_image_stride smaller than the output image");
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    auto Hb = [&](size_t off) { return (__bf16*)(base + off); };
    HmBf16Bufs q{};
    q.P0 = Hb(w.P0);
    for (int i = 0; i < 4; ++i) { q.A[i] = Hb(w.S[i][0]); q.Ta[i] = Hb(w.S[i][1]); q.Tb[i] = Hb(w.S[i][2]); q.Td[i] = Hb(w.S[i][3]); }
    q.T4 = Hb(w.T4); q.C3 = Hb(w.C3); q.Y3 = Hb(w.Y3); q.C2 = Hb(w.C2); q.Y2 = Hb(w.Y2); q.C1 = Hb(w.C1); q.Y1 = Hb(w.Y1);
    q.ZP = Hb(w.ZP);
    q.reg = base + w.WALL;
    q.split_slab = (float*)(base + w.SPLIT); q.split_floats = HM_BN_SPLIT_FLOATS;
    q.dec_slab = (float*)(base + w.DEC);     q.dec_split_floats = ((size_t)40 << 20) / 4;
    q.ones = (const float*)(base + w.ONES);  q.zeros = (const float*)(base + w.ZEROS);
    HmBnBatch bnb{BnBatchScratch{(float*)(base + w.PART), (float*)(base + w.SC), (float*)(base + w.SH)}};
    const int cus = device_cu_count();
    EGO_HIP(zero_fill(q.ZP, 256, s));
    EGO_HIP(zero_fill((void*)q.zeros, 4096, s));
    hipLaunchKernelGGL(fill_f32_kernel, dim3(4), dim3(256), 0, s, (float*)q.ones, 1.f, 1024);
    EGO_HIP(hipGetLastError());
    PackTable PT;
    const int stage_sides[4] = {s64, s32, s16, s8};
    EGO_CHECK(hm_pack_plan(&p, p.nblk, &PT, 2L * B, stage_sides, cus, q.split_floats, q.dec_split_floats, 2L * chunk) != 0,
              "egotap_hm_forward_bnbatch: the estimator has more layers than the pack table holds");
    hipLaunchKernelGGL(pack_all_bf16s_kernel, dim3(PT.blocks), dim3(256), 0, s, PT, q.reg);      // (its eval-mode BatchNorm folds are written and not read here)
    EGO_HIP(hipGetLastError());
    int li = 0, bi = 0;
    EGO_HIP(hm_bf16_backbone(h, p, PT, li, bi, q, left, right, B, S0, &bnb, s));      // whole batch: the statistics couple its frames
    EGO_CHECK(bi == PT.nb, "egotap_hm_forward_bnbatch: pack plan out of step with the forward");
    const long px[4] = {(long)s64 * s64, (long)s32 * s32, (long)s16 * s16, (long)s8 * s8};
    for (int lo = 0; lo < B; lo += chunk) {                                             // the decoder has no BatchNorm: chunks keep its scratch small
        const int bc = B - lo < chunk ? B - lo : chunk;
        const __bf16* lv[4];
        for (int i = 0; i < 4; ++i) lv[i] = q.A[i] + (size_t)lo * px[i] * 2 * HM_CH[i];
        EGO_HIP(hm_bf16_decoder(h, p, PT, li, q, lv, bc, S0, out + (size_t)lo * out_image_stride, out_image_stride, s));
    }
    return EGOTAP_OK;
}
#endif

// ------------------------------------------------------------------------------------------------ single operators
template <class Cfg>
static hipError_t linear_tile(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int epi,
                              const float* r, const float* g, const float* beta, const float* mean, const float* var,
                              hipStream_t s) {
    const ALoadPlain al{x, K};
    const SegMat W = segmat1(w, N, K);
    switch (epi) {
        case 0: return gemm_f32_launch<Cfg>(al, W, EpiBias{segvec1(b, N)}, y, N, M, N, K, s);
        case 1: return gemm_f32_launch<Cfg>(al, W, EpiBiasRes{segvec1(b, N), r, N}, y, N, M, N, K, s);
        case 2: return gemm_f32_launch<Cfg>(al, W, EpiBiasGelu{segvec1(b, N)}, y, N, M, N, K, s);
        case 3: return gemm_f32_launch<Cfg>(al, W, EpiBnLrelu{b, g, beta, mean, var, 1e-5f, 0.2f}, y, N, M, N, K, s);
        default: return hipErrorInvalidValue;
    }
}

#if EGOTAP_IN(0)
extern "C" int egotap_linear_f32(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int epi,
                                 const float* r, const float* g, const float* beta, const float* mean, const float* var,
                                 int tile, void* stream) {
    EGO_CHECK(x && w && b && y, "egotap_linear_f32: null argument");
    EGO_CHECK(M >= 0 && N > 0 && K > 0, "egotap_linear_f32: bad shape");
    EGO_CHECK(epi >= 0 && epi <= 3, "egotap_linear_f32: epi must be 0..3");
    EGO_CHECK(epi != 1 || r, "egotap_linear_f32: residual pointer missing");
    EGO_CHECK(epi != 3 || (g && beta && mean && var), "egotap_linear_f32: BatchNorm pointers missing");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (tile != 0 && tile != 1 && epi != 0) { egotap_set_error("non-default tiles are built for epi 0 only"); return EGOTAP_ERR_INVALID; }
    switch (tile) {
        case 0: case 1: e = linear_tile<TileA>(x, w, b, y, M, N, K, epi, r, g, beta, mean, var, s); break;
        case 2: e = gemm_f32_launch<TileB>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 3: e = gemm_f32_launch<TileC>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 4: e = gemm_f32_launch<TileD>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 5: e = gemm_f32_launch<TileE>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 6: e = gemm_f32_launch<TileF>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 7: e = gemm_f32_launch<TileG>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 8: e = gemm_f32_pipe_launch<PipeA>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 9: e = gemm_f32_pipe_launch<PipeB>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 10: e = gemm_f32_pipe_launch<PipeC>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 11: e = gemm_f32_pipe_launch<PipeD>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, s); break;
        case 12: e = gemm_f32_persist_launch<PipeD>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        case 13: e = gemm_bf16_persist_launch<BfCfg<3>>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        case 14: e = gemm_bf16_persist_launch<BfCfg<1>>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        case 15: e = gemm_bf16_persist_launch<BfCfg<3, 1>>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        case 16: e = gemm_bf16_persist_launch<BfCfg<1, 1>>(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        case 19: e = gemm_f32_dma_launch(ALoadPlain{x, K}, segmat1(w, N, K), EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), s); break;
        default: egotap_set_error("unknown tile id %d", tile); return EGOTAP_ERR_INVALID;
    }
    if (e == hipErrorInvalidValue) {
        egotap_set_error("egotap_linear_f32: N=%d / K=%d not a multiple of the tile (%s)", N, K, egotap_gemm_tile_name(tile));
        return EGOTAP_ERR_INVALID;
    }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
// y = x w^T + b on gemm_bf16_dma_kernel with CALLER-OWNED bf16 copies of both operands (x [M,K], w [N,K], row-major) -- the
// kernel the plain-bf16 mode runs after rounding its operands; exported for the operator tests
extern "C" int egotap_linear_bf16_dma(const void* x_bf16, const void* w_bf16, const float* b, float* y, int M, int N, int K, void* stream) {
    EGO_CHECK(x_bf16 && w_bf16 && b && y, "egotap_linear_bf16_dma: null argument");
    EGO_CHECK(M >= 0 && N > 0 && K > 0 && K % 8 == 0, "egotap_linear_bf16_dma: bad shape (K must be a multiple of 8)");
    EGO_CHECK((((uintptr_t)x_bf16 | (uintptr_t)w_bf16) & 15) == 0, "egotap_linear_bf16_dma: operands must be 16-byte aligned");
    SegMatB Bm;
    for (int i = 0; i < 3; ++i) Bm.p[i] = (const __bf16*)w_bf16;
    Bm.seg = N; Bm.ld = K;
    hipError_t e = gemm_bf16_dma_launch((const __bf16*)x_bf16, (long)K, Bm, EpiBias{segvec1(b, N)}, y, N, M, N, K, device_cu_count(), (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_linear_bf16_dma: N=%d / K=%d not a multiple of the 256 x 256 x 32 tile", N, K); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int rows, int dim,
                                    float eps, void* stream) {
    EGO_CHECK(x && y && gamma && beta, "egotap_layernorm_f32: null argument");
    EGO_CHECK(dim == 1024, "egotap_layernorm_f32: dim must be 1024 (ViT hidden size)");
    EGO_HIP(launch_ln(x, y, gamma, beta, rows, eps, (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(0)
extern "C" int egotap_attention_f32(const float* qkv, float* ctx, int B, int N, int heads, void* stream) {
    EGO_CHECK(qkv && ctx, "egotap_attention_f32: null argument");
    EGO_CHECK(N >= 32 && N % 4 == 0, "egotap_attention_f32: sequence length must be at least 32 and a multiple of 4 (any heatmap side that is a multiple of 16)");
    EGO_CHECK(heads > 0, "egotap_attention_f32: heads must be positive");
    EGO_HIP(attention_f32_launch(qkv, ctx, B, N, heads, (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

// per-sample MPJPE / PA-MPJPE of a batch of poses (egotap_autoencoder_model.py:329-350, utils/util.py:328-379)
#if EGOTAP_IN(0)
extern "C" int egotap_pose_metrics(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pa_mpjpe, float* aligned,
                                   void* stream) {
    if (B == 0) return EGOTAP_OK;
    EGO_CHECK(pred && gt && mpjpe && pa_mpjpe, "egotap_pose_metrics: null argument");
    EGO_CHECK(B > 0 && J >= 1 && J <= EGOTAP_MAX_JOINTS, "egotap_pose_metrics: bad shape B=%d J=%d", B, J);
    hipLaunchKernelGGL(pose_metrics_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pred, gt, B, J, mpjpe, pa_mpjpe, aligned);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// the same metrics as the reference computes them for a batch of 2 or 3 frames (utils/util.py:337 skips its transpose there)
#if EGOTAP_IN(0)
extern "C" int egotap_pose_metrics_batch_axes(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pa_mpjpe, float* aligned,
                                              void* stream) {
    EGO_CHECK(pred && gt && mpjpe && pa_mpjpe, "egotap_pose_metrics_batch_axes: null argument");
    EGO_CHECK((B == 2 || B == 3) && J >= 1 && J <= EGOTAP_MAX_JOINTS,
              "egotap_pose_metrics_batch_axes: the reference takes this branch for batches of 2 or 3 frames only (B=%d J=%d)", B, J);
    hipLaunchKernelGGL(pose_metrics_batch_axes_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pred, gt, B, J, mpjpe, pa_mpjpe, aligned);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// joints -> ground-truth heatmaps in the lifting head's input layout (dataloader/data_loader.py:76-215 with --use_gt_heatmap)
#if EGOTAP_IN(0)
extern "C" int egotap_synth_heatmaps(const float* pts2d_left, const float* pts2d_right, const float* pose3d, const int* parents, int B,
                                     int J, int res, float* hm, float* plength, float* theta, void* stream) {
    if (B == 0) return EGOTAP_OK;
    EGO_CHECK(pts2d_left && pts2d_right && pose3d && parents && hm, "egotap_synth_heatmaps: null argument");
    EGO_CHECK(B > 0 && J >= 1 && J <= 64 && res >= 16 && res <= 128, "egotap_synth_heatmaps: bad shape B=%d J=%d res=%d", B, J, res);
    GaussTaps g;                                   // scipy.ndimage._gaussian_kernel1d(sigma = 1, order 0, radius 4)
    double sum = 0.0;
    for (int k = -4; k <= 4; ++k) { g.w[k + 4] = exp(-0.5 * k * k); sum += g.w[k + 4]; }
    for (int k = 0; k < 9; ++k) g.w[k] /= sum;
    const size_t lds = (size_t)2 * res * res * 4;
    static bool attr_done = false;
    if (!attr_done) {
        EGO_HIP(hipFuncSetAttribute((const void*)heatmap_synth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 128 * 4));
        attr_done = true;
    }
    hipLaunchKernelGGL(heatmap_synth_kernel, dim3(B * 2 * J), dim3(256), lds, (hipStream_t)stream, pts2d_left, pts2d_right, pose3d, parents,
                       B, J, res, g, hm, plength, theta);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// same operator with the arithmetic of egotap_set_precision (EGOTAP_PREC_*): fp32 MFMA, bf16x3 split or plain bf16
#if EGOTAP_IN(0)
extern "C" int egotap_attention(const float* qkv, float* ctx, int B, int N, int heads, int precision, void* stream) {
    EGO_CHECK(qkv && ctx, "egotap_attention: null argument");
    EGO_CHECK(N >= 32 && N % 4 == 0, "egotap_attention: sequence length must be at least 32 and a multiple of 4");
    EGO_CHECK(heads > 0, "egotap_attention: heads must be positive");
    // [r5] the bf16 / bf16x3 kernels tile the sequence in whole 32-key blocks; sequence lengths that are not (heatmap sides 32, 48, 96 ...: the
    // reference allows every multiple of 16) run on the exact-fp32 kernel, which masks the last key tile -- a FALLBACK BY NAME, documented in egotap.h
    if (precision == EGOTAP_PREC_F32 || N % 32 != 0) EGO_HIP(attention_f32_launch(qkv, ctx, B, N, heads, (hipStream_t)stream));
    else if (precision == EGOTAP_PREC_BF16X3) EGO_HIP(attention_bf16_launch<3>(qkv, ctx, B, N, heads, (hipStream_t)stream));
    else if (precision == EGOTAP_PREC_BF16) EGO_HIP(attention_bf16_launch<1>(qkv, ctx, B, N, heads, (hipStream_t)stream));
    else { egotap_set_error("egotap_attention: unknown precision %d", precision); return EGOTAP_ERR_INVALID; }
    return EGOTAP_OK;
}
#endif

// ================================================================================================ training operators
// Building blocks of the training step (egotap_autoencoder_model.py:299-323), called by the autograd glue in
// egotap_amd/autograd.py.  Each takes caller-owned device buffers and the caller's stream.
#include "attention_bwd_f32.h"
#include "attention_bwd_bf16.h"
#include "gemm_tn_f32.h"
#include "gemm_tn_bf16.h"
#include "train_ops.h"

using TnBig = TnCfg<256, 256, 16, 4, 2>;     // 8 waves, 64x128 per wave
using TnSmall = TnCfg<128, 128, 16, 2, 2>;   // 4 waves, 64x64 per wave

enum { LD_PLAIN = 0, LD_PATCH = 1, LD_TOKENS = 2, LD_ROT = 3, LD_STEREO = 4, LD_STEREO_GATED = 5 };
enum { TE_NONE = 0, TE_BIAS = 1, TE_BIAS_RES = 2, TE_BIAS_GELU_SAVE = 3, TE_ACCUM = 4, TE_GELU_GRAD = 5, TE_PATCH = 6 };

template <class AL, class Epi>
static hipError_t nt_any(Handle* h, const AL& al, const SegMat& W, const Epi& epi, float* C, long ldc, int M, int N, int K, hipStream_t s) {
    if (N % 256 == 0 && K % 32 == 0 && M >= 1024 && h->precision == EGOTAP_PREC_BF16X3)
        return gemm_bf16_persist_launch<BfCfg<3, 1>, AL, Epi>(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
    if (N % 256 == 0 && K % 32 == 0 && M >= 1024 && h->precision == EGOTAP_PREC_BF16)
        return gemm_bf16_plain(h, al, W, epi, C, ldc, M, N, K, s);
    if (N % 256 == 0 && K % 16 == 0 && M >= 1024) {
        // [r3] the training step's fp32 GEMMs (forward with saved pre-activations, input gradients) on the LDS-DMA kernel of the inference path
        // where the loader is pure address math: same tile, same k order, same bits as the register-staged persistent kernel, 6 % faster
        if constexpr (AL::HAS_PTR) {
            if (N % DmaF32Cfg::BN == 0 && K % DmaF32Cfg::BK == 0 && W.seg % DmaF32Cfg::BN == 0 && W.ld % 4 == 0 && al.dma_ok())
                return gemm_f32_dma_launch(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
        }
        return gemm_f32_persist_launch<PipeD, AL, Epi>(al, W, epi, C, ldc, M, N, K, device_cu_count(), s);
    }
    if (N % 128 == 0 && K % 32 == 0) return gemm_f32_launch<TileA, AL, Epi>(al, W, epi, C, ldc, M, N, K, s);
    return hipErrorInvalidValue;
}

template <class AL>
static hipError_t nt_epi(const AL& al, const float* w, const float* b, float* y, int M, int N, int K, int epi, const float* r,
                         float* z, const LiftParams* lp, Handle* h, hipStream_t s) {
    const SegMat W = segmat1(w, N, K);
    if constexpr (!std::is_same<AL, ALoadPlain>::value) {
        // gathering loaders feed first layers only (fc1 of the two encoders, the PU projections): bias epilogue, nothing else is
        // instantiated (each combination is four GEMM kernels: compile time)
        if (epi == TE_BIAS) return nt_any(h, al, W, EpiBias{segvec1(b, N)}, y, N, M, N, K, s);
        return hipErrorInvalidValue;
    } else {
        switch (epi) {
            case TE_NONE: return nt_any(h, al, W, EpiNone{}, y, N, M, N, K, s);
            case TE_BIAS: return nt_any(h, al, W, EpiBias{segvec1(b, N)}, y, N, M, N, K, s);
            case TE_BIAS_RES: return nt_any(h, al, W, EpiBiasRes{segvec1(b, N), r, N}, y, N, M, N, K, s);
            case TE_BIAS_GELU_SAVE: return nt_any(h, al, W, EpiBiasGeluSave{segvec1(b, N), z, N}, y, N, M, N, K, s);
            case TE_ACCUM: return nt_any(h, al, W, EpiAccum{r, N}, y, N, M, N, K, s);
            case TE_GELU_GRAD: return nt_any(h, al, W, EpiGeluGrad{r, N}, y, N, M, N, K, s);
            default: return hipErrorInvalidValue;
        }
    }
}

// y[M,N] = epi(A(x) W^T (+ b)); `loader` gathers A from x exactly as the eval forward does; aux = gate source (F) for LD_STEREO_GATED.
// r: residual / accumulate source / saved pre-activation (by epi); z: pre-activation output for TE_BIAS_GELU_SAVE.
#if EGOTAP_IN(1)
extern "C" int egotap_train_gemm_nt(egotap_handle h, int loader, const float* x, int64_t lda, const float* aux, const float* w, const float* b,
                                    float* y, int M, int N, int K, int epi, const float* r, float* z, int Bsz, void* stream) {
    EGO_CHECK(h && x && w && y, "egotap_train_gemm_nt: null argument");
    if (lda <= 0) lda = K;
    hipStream_t s = (hipStream_t)stream;
    const int S = h->cfg.hm_size;
    hipError_t e;
    switch (loader) {
        case LD_PLAIN: e = nt_epi(ALoadPlain{x, (long)lda}, w, b, y, M, N, K, epi, r, z, nullptr, h, s); break;
        case LD_TOKENS: e = nt_epi(ALoadTokens{x, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, w, b, y, M, N, K, epi, r, z, nullptr, h, s); break;
        case LD_ROT: e = nt_epi(ALoadRot{x, h->C, h->J, S * S}, w, b, y, M, N, K, epi, r, z, nullptr, h, s); break;
        case LD_STEREO: e = nt_epi(ALoadStereo{x, Bsz, h->J, h->hid}, w, b, y, M, N, K, epi, r, z, nullptr, h, s); break;
        case LD_STEREO_GATED:
            e = nt_epi(ALoadStereoGated{ALoadStereo{x, Bsz, h->J, h->hid}, aux, h->H + 2 * h->hid, h->H}, w, b, y, M, N, K, epi, r, z, nullptr, h, s);
            break;
        default: egotap_set_error("egotap_train_gemm_nt: bad loader %d", loader); return EGOTAP_ERR_INVALID;
    }
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_train_gemm_nt: unsupported shape M=%d N=%d K=%d epi=%d", M, N, K, epi); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

// patch embedding forward in training (same kernel as eval): hm [B,C,S,S] -> x [B*seq, D]
#if EGOTAP_IN(1)
extern "C" int egotap_train_patch_fwd(egotap_handle h, const float* hm, int B, const float* w, const float* b, const float* mask_tok,
                                      const float* pos, float* x, void* stream) {
    EGO_CHECK(h && hm && w && b && mask_tok && pos && x, "egotap_train_patch_fwd: null argument");
    const int S = h->cfg.hm_size, D = h->D, M = B * h->seq;
    ALoadPatch al{hm, h->C, S, h->seq, h->side, h->ppd, h->grid, h->T};
    EpiPatch ep{b, mask_tok, pos, D, h->seq, h->side, h->ppd, h->grid, h->T};
    EGO_HIP((gemm_big(h, "patch_embed", al, segmat1(w, D, 256), ep, x, D, M, D, 256, (hipStream_t)stream)));
    return EGOTAP_OK;
}
#endif

template <class XL>
static hipError_t tn_any(Handle* h, const float* dy, const XL& xl, float* dw, float* ws, size_t ws_bytes, int M, int N, int K, int acc, hipStream_t s,
                         long ldy = 0) {
    if (ldy <= 0) ldy = N;
    if (N % 256 == 0 && K % 256 == 0 && M >= 1024 && h->precision == EGOTAP_PREC_BF16X3)
        return gemm_tn_bf16_launch<TnBfCfg<3>, XL>(dy, ldy, xl, dw, ws, ws_bytes, M, N, K, device_cu_count(), acc, s);
    if (N % 256 == 0 && K % 256 == 0 && M >= 1024 && h->precision == EGOTAP_PREC_BF16)
        return gemm_tn_bf16_launch<TnBfCfg<1>, XL>(dy, ldy, xl, dw, ws, ws_bytes, M, N, K, device_cu_count(), acc, s);
    if constexpr (std::is_same<XL, ALoadPlain>::value) {      // [r3] plain operands: the DMA-staged kernel (gemm_tn_f32.h), same summation order per element
        if (M >= 1024 && gemm_tn_f32_dma_ok(dy, ldy, xl.A, xl.lda, M, N, K))
            return gemm_tn_f32_dma_launch(dy, ldy, xl.A, xl.lda, dw, ws, ws_bytes, M, N, K, device_cu_count(), acc, s);
    }
    if (N % 256 == 0 && K % 256 == 0) return gemm_tn_f32_launch<TnBig, XL>(dy, ldy, xl, dw, ws, ws_bytes, M, N, K, device_cu_count(), acc, s);
    if (N % 128 == 0 && K % 128 == 0) return gemm_tn_f32_launch<TnSmall, XL>(dy, ldy, xl, dw, ws, ws_bytes, M, N, K, device_cu_count(), acc, s);
    return hipErrorInvalidValue;
}

// dW[N,K] (+)= dY[M,N]^T A(x)[M,K]   (weight gradient; x goes through the forward's loader)
#if EGOTAP_IN(1)
extern "C" int egotap_train_gemm_tn(egotap_handle h, int loader, const float* dy, int64_t ldy, const float* x, const float* aux, float* dw, int M,
                                    int N, int K, int accumulate, int Bsz, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(h && dy && x && dw && ws, "egotap_train_gemm_tn: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int S = h->cfg.hm_size;
    float* w = (float*)ws;
    hipError_t e;
    switch (loader) {
        case LD_PLAIN: e = tn_any(h, dy, ALoadPlain{x, K}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy); break;
        case LD_PATCH: e = tn_any(h, dy, ALoadPatch{x, h->C, S, h->seq, h->side, h->ppd, h->grid, h->T}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy); break;
        case LD_TOKENS: e = tn_any(h, dy, ALoadTokens{x, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy); break;
        case LD_ROT: e = tn_any(h, dy, ALoadRot{x, h->C, h->J, S * S}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy); break;
        case LD_STEREO: e = tn_any(h, dy, ALoadStereo{x, Bsz, h->J, h->hid}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy); break;
        case LD_STEREO_GATED:
            e = tn_any(h, dy, ALoadStereoGated{ALoadStereo{x, Bsz, h->J, h->hid}, aux, h->H + 2 * h->hid, h->H}, dw, w, ws_bytes, M, N, K, accumulate, s, ldy);
            break;
        default: egotap_set_error("egotap_train_gemm_tn: bad loader %d", loader); return EGOTAP_ERR_INVALID;
    }
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_train_gemm_tn: unsupported shape N=%d K=%d", N, K); return EGOTAP_ERR_INVALID; }
    if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_train_gemm_tn: workspace too small (%zu bytes) for N*K=%ld", ws_bytes, (long)N * K); return EGOTAP_ERR_WORKSPACE; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_colsum(const float* y, int64_t ldy, float* out, int M, int N, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(y && out && ws, "egotap_train_colsum: null argument");
    EGO_CHECK(N % 4 == 0, "egotap_train_colsum: N must be a multiple of 4");
    EGO_HIP(colsum_f32_launch(y, ldy > 0 ? ldy : N, out, (float*)ws, ws_bytes, M, N, accumulate, (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

// (test aid, host only) the split count the weight-gradient launches choose: wgrad_pick_splits of gemm_tn_f32.h
#if EGOTAP_IN(1)
extern "C" int egotap_debug_wgrad_splits(int tiles, int64_t slabs, int64_t n_floats, size_t slab_bytes, int num_cu, int lds_bytes, double slab_us,
                                         double fixed_us, int* per) {
    int p = 0;
    const int s = (tiles > 0 && slabs > 0 && n_floats > 0 && num_cu > 0) ? wgrad_pick_splits(tiles, (long)slabs, (long)n_floats, slab_bytes, num_cu, lds_bytes, slab_us, fixed_us, &p) : 0;
    if (per) *per = p;
    return s;
}
#endif

// [r3] dW[N,K] (+)= dY^T X and db[N] (+)= column sums of dY in one call: weight and bias gradient of a Linear layer with a plain input.  In fp32,
// when the DMA-staged kernel applies, the workgroups that stage dY for the product also sum its columns (no second pass over dY: 5.4 GB per
// ViT layer at B = 256); otherwise the two operators one after the other.
#if EGOTAP_IN(1)
extern "C" int egotap_train_gemm_tn_bias(egotap_handle h, const float* dy, int64_t ldy, const float* x, float* dw, float* db, int M, int N, int K,
                                         int accumulate, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(h && dy && x && dw && db && ws, "egotap_train_gemm_tn_bias: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (ldy <= 0) ldy = N;
    if (h->precision == EGOTAP_PREC_F32 && M >= 1024 && gemm_tn_f32_dma_ok(dy, ldy, x, K, M, N, K)) {
        hipError_t e = gemm_tn_f32_dma_launch(dy, ldy, x, K, dw, (float*)ws, ws_bytes, M, N, K, device_cu_count(), accumulate, s, db);
        if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_train_gemm_tn_bias: workspace too small (%zu bytes) for N*K=%ld", ws_bytes, (long)N * K); return EGOTAP_ERR_WORKSPACE; }
        EGO_HIP(e);
        return EGOTAP_OK;
    }
    if (h->precision != EGOTAP_PREC_F32 && N % 256 == 0 && K % 256 == 0 && M >= 1024 && ldy % 4 == 0) {      // bf16x3 / bf16 products: the same, on fp32 dY before the rounding
        const ALoadPlain xl{x, K};
        hipError_t e = h->precision == EGOTAP_PREC_BF16X3
                           ? gemm_tn_bf16_launch<TnBfCfg<3>, ALoadPlain>(dy, ldy, xl, dw, (float*)ws, ws_bytes, M, N, K, device_cu_count(), accumulate, s, db)
                           : gemm_tn_bf16_launch<TnBfCfg<1>, ALoadPlain>(dy, ldy, xl, dw, (float*)ws, ws_bytes, M, N, K, device_cu_count(), accumulate, s, db);
        if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_train_gemm_tn_bias: workspace too small (%zu bytes) for N*K=%ld", ws_bytes, (long)N * K); return EGOTAP_ERR_WORKSPACE; }
        EGO_HIP(e);
        return EGOTAP_OK;
    }
    int rc = egotap_train_gemm_tn(h, 0, dy, ldy, x, nullptr, dw, M, N, K, accumulate, 0, ws, ws_bytes, stream);
    if (rc != EGOTAP_OK) return rc;
    return egotap_train_colsum(dy, ldy, db, M, N, accumulate, ws, ws_bytes, stream);
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_transpose(const float* in, float* out, int R, int C, int64_t ldo, void* stream) {
    EGO_CHECK(in && out, "egotap_train_transpose: null argument");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, C, (long)(ldo > 0 ? ldo : R));
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_layernorm_fwd(const float* x, float* y, const float* g, const float* b, float* mean, float* rstd, int rows,
                                          float eps, void* stream) {
    EGO_CHECK(x && y && g && b && mean && rstd, "egotap_train_layernorm_fwd: null argument");
    if (rows > 0) hipLaunchKernelGGL(layernorm_fwd_stats_kernel<1024>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, g, b, mean, rstd, rows, eps);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// dx = LN'(dy) (+ dres); dgamma / dbeta (+)= column sums.  ws: >= (b + 1 + ceil(b/64)) * 2048 floats, b = ceil(rows/64)
#if EGOTAP_IN(1)
extern "C" int egotap_train_layernorm_bwd(const float* x, const float* dy, const float* g, const float* mean, const float* rstd,
                                          const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int accumulate,
                                          void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(x && dy && g && mean && rstd && dx && dgamma && dbeta && ws, "egotap_train_layernorm_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int rows_per_wave = 16, blocks = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
    EGO_CHECK(ws_bytes >= (size_t)blocks * 2 * 1024 * 4, "egotap_train_layernorm_bwd: workspace too small");
    float* part = (float*)ws;
    hipLaunchKernelGGL(layernorm_bwd_kernel<1024>, dim3(blocks), dim3(256), 0, s, x, dy, g, mean, rstd, dres, dx, part, rows, rows_per_wave);
    EGO_HIP(hipGetLastError());
    // part[block][2][1024] = 'blocks' rows of 2048 floats (dgamma_k | dbeta_k): column sums in two fixed-order stages -- 64 rows per
    // workgroup into part2[gy][2048], then the gy rows into tmp (a single 2-workgroup pass over all rows took 0.85 ms at B = 256)
    float* tmp = part + (size_t)blocks * 2048;
    float* part2 = tmp + 2048;
    const int gy = (blocks + 63) / 64;
    EGO_CHECK(ws_bytes >= ((size_t)blocks * 2048 + 2048 + (size_t)gy * 2048) * 4, "egotap_train_layernorm_bwd: workspace too small");
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(2048 / 4 / 64, gy), dim3(256), 0, s, (const float*)part, 2048L, part2, blocks, 2048, 64);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(2), dim3(256), 0, s, (const float*)part2, tmp, 2048L, gy, 0);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(256), 0, s, tmp, dgamma, 1024L, 1, accumulate);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(256), 0, s, tmp + 1024, dbeta, 1024L, 1, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// BatchNorm1d (training mode) + LeakyReLU(0.2) over z [R, C]: y, saved mean / rstd, running-stat update
// (network_utils.py:123-142; momentum 0.1, unbiased running variance).  ws >= (2 * ceil(R/256) * C + 2C) floats
#if EGOTAP_IN(1)
extern "C" int egotap_train_bn_lrelu_fwd(const float* z, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                                         float* run_mean, float* run_var, int R, int C, float eps, float momentum, void* ws,
                                         size_t ws_bytes, void* stream) {
    EGO_CHECK(z && y && gamma && beta && mean && rstd && ws, "egotap_train_bn_lrelu_fwd: null argument");
    EGO_CHECK(C % 4 == 0 && R > 1, "egotap_train_bn_lrelu_fwd: C must be a multiple of 4 and R > 1");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256, gy = (R + rpb - 1) / rpb;
    EGO_CHECK(ws_bytes >= ((size_t)gy * 2 * C + 2 * C) * 4, "egotap_train_bn_lrelu_fwd: workspace too small");
    float* part = (float*)ws;
    float* sums = part + (size_t)gy * 2 * C;
    const dim3 grid((C / 4 + 63) / 64, gy);
    hipLaunchKernelGGL(colstats_kernel<0>, grid, dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, part, R, C, rpb, 0.f);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((2 * C / 4 + 255) / 256), dim3(256), 0, s, part, sums, 2L * C, gy, 0);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, mean, C, R, 0, eps, momentum, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(colstats_kernel<0>, grid, dim3(256), 0, s, z, nullptr, nullptr, mean, nullptr, part, R, C, rpb, 0.f);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((2 * C / 4 + 255) / 256), dim3(256), 0, s, part, sums, 2L * C, gy, 0);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums + C, rstd, C, R, 1, eps, momentum, run_mean, run_var, mean);
    const long n4 = (long)R * C / 4;
    hipLaunchKernelGGL(bn_apply_lrelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, y, mean, rstd, gamma, beta, n4, C, 0.2f);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_bn_lrelu_bwd(const float* z, const float* y, const float* dy, const float* gamma, const float* mean,
                                         const float* rstd, float* dz, float* dgamma, float* dbeta, int R, int C, int accumulate,
                                         void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(z && y && dy && gamma && mean && rstd && dz && dgamma && dbeta && ws, "egotap_train_bn_lrelu_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = 256, gy = (R + rpb - 1) / rpb;
    EGO_CHECK(ws_bytes >= ((size_t)gy * 2 * C + 2 * C) * 4, "egotap_train_bn_lrelu_bwd: workspace too small");
    float* part = (float*)ws;
    float* sums = part + (size_t)gy * 2 * C;       // [0:C] = sum dyb (dbeta), [C:2C] = sum dyb*xhat (dgamma)
    const dim3 grid((C / 4 + 63) / 64, gy);
    hipLaunchKernelGGL(colstats_kernel<1>, grid, dim3(256), 0, s, z, y, dy, mean, rstd, part, R, C, rpb, 0.2f);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((2 * C / 4 + 255) / 256), dim3(256), 0, s, part, sums, 2L * C, gy, 0);
    const long n4 = (long)R * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, y, dy, mean, rstd, gamma, sums, sums + C, dz, n4, C, R, 0.2f);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((C / 4 + 255) / 256), dim3(256), 0, s, sums, dbeta, (long)C, 1, accumulate);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((C / 4 + 255) / 256), dim3(256), 0, s, sums + C, dgamma, (long)C, 1, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_attention_fwd(const float* qkv, float* ctx, float* lse, int B, int N, int heads, int precision, void* stream) {
    EGO_CHECK(qkv && ctx && lse, "egotap_train_attention_fwd: null argument");
    EGO_CHECK(N >= 32 && N % 4 == 0 && heads > 0, "egotap_train_attention_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (N % 32 != 0) EGO_HIP(attention_f32_launch(qkv, ctx, B, N, heads, s, lse));      // (the exact-fp32 kernel masks a ragged last key tile; see egotap_attention)
    else if (precision == EGOTAP_PREC_BF16X3) EGO_HIP(attention_bf16_launch<3>(qkv, ctx, B, N, heads, s, lse));
    else if (precision == EGOTAP_PREC_BF16) EGO_HIP(attention_bf16_launch<1>(qkv, ctx, B, N, heads, s, lse));
    else EGO_HIP(attention_f32_launch(qkv, ctx, B, N, heads, s, lse));
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_attention_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* delta,
                                          float* dqkv, int B, int N, int heads, int precision, void* stream) {
    EGO_CHECK(qkv && ctx && dctx && lse && delta && dqkv, "egotap_train_attention_bwd: null argument");
    EGO_CHECK(N > 0 && heads > 0, "egotap_train_attention_bwd: bad shape");
    EGO_CHECK(N % 32 == 0, "egotap_train_attention_bwd: the attention BACKWARD kernels need a sequence length that is a multiple of 32 (heatmap sides 64, 128, ...: "
              "every shipped configuration); N = %d (evaluation runs at any heatmap side that is a multiple of 16)", N);
    hipStream_t s = (hipStream_t)stream;
    if (precision == EGOTAP_PREC_BF16X3) EGO_HIP(attention_bwd_bf16_launch<3>(qkv, ctx, dctx, lse, delta, dqkv, B, N, heads, s));
    else if (precision == EGOTAP_PREC_BF16) EGO_HIP(attention_bwd_bf16_launch<1>(qkv, ctx, dctx, lse, delta, dqkv, B, N, heads, s));
    else EGO_HIP(attention_bwd_f32_launch(qkv, ctx, dctx, lse, delta, dqkv, B, N, heads, s));
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_pose_loss(egotap_handle h, const float* pred, const float* gt, float* dpred, float* out, float* partial,
                                      int B, float lambda_mpjpe, float lambda_cos_sim, void* stream) {
    EGO_CHECK(h && pred && gt && dpred && out && partial, "egotap_train_pose_loss: null argument");
    static const int ue[16] = {0, 0, 1, 1, 2, 3, 4, 5, 2, 3, 8, 9, 10, 11, 12, 13};                 // utils/util.py:51
    static const int ec[18] = {0, 0, 1, 2, 3, 4, 1, 6, 7, 8, 2, 10, 11, 12, 6, 14, 15, 16};         // utils/util.py:52
    LossArgs a;
    a.pred = pred; a.gt = gt; a.dpred = dpred; a.partial = partial;
    a.B = B; a.J = h->out_joints; a.estimate_head = h->cfg.estimate_head;
    a.lam_pose = lambda_mpjpe; a.lam_cos = lambda_cos_sim * lambda_mpjpe;
    const int n = a.estimate_head ? 16 : 18;
    EGO_CHECK(a.J + (a.estimate_head ? 0 : 1) == n, "egotap_train_pose_loss: joint count does not match the preset's kinematic list");
    for (int i = 0; i < 20; ++i) a.parents[i] = i < n ? (a.estimate_head ? ue[i] : ec[i]) : 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pose_loss_kernel, dim3(B), dim3(64), 0, s, a);
    hipLaunchKernelGGL(pose_loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, out, B, a.J, a.lam_pose, a.lam_cos);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                                  double weight_decay, int step, void* stream) {
    EGO_CHECK(p && g && m && v && step >= 1, "egotap_train_adamw: bad argument");
    // bias corrections and step size in double on the host, as torch.optim.AdamW computes them (python floats); the kernel gets
    // step_size = lr / bc1 and 1 / sqrt(bc2) already rounded once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2s = sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1, (float)bc2s);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_adamw_multi(const void* table, int nseg, const float* g, float* m, float* v, int64_t span, double lr, double beta1,
                                        double beta2, double eps, double weight_decay, int step, void* stream) {
    EGO_CHECK(table && g && m && v && nseg >= 1 && span >= 1 && step >= 1, "egotap_train_adamw_multi: bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2s = sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)(((span + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long*)table, nseg, g, m, v,
                       (long)span, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1, (float)bc2s);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(1)
extern "C" int egotap_train_add_inplace(float* out, const float* in, int64_t n, void* stream) {
    EGO_CHECK(out && in && n % 4 == 0, "egotap_train_add_inplace: bad argument");
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, in, (long)(n / 4));
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// ------------------------------------------------------------------------------------------------ PU chain + pose head (training)
struct PuSaved { size_t F0, G0, GP0, HS0, C0, F1, G1, GP1, HS1, C1, ZERO, HPA, HPB, FAULT, total; };
static PuSaved pu_saved(const Handle* h, int B) {
    PuSaved w;
    const size_t JB = (size_t)h->J * B, H = h->H, NF0 = H + 2 * h->hid;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = al256(o + floats * 4); return r; };
    // G0 / G1: the state-independent gate inputs Gin of every step (kept intact, so that a faulted one-launch recurrence can be
    // redone from them); GP0 / GP1: the full gate pre-activations the backward reads
    w.F0 = take(JB * NF0); w.G0 = take(JB * 4 * H); w.GP0 = take(JB * 4 * H); w.HS0 = take(JB * H); w.C0 = take(JB * H);
    w.F1 = take(JB * H); w.G1 = take(JB * 4 * H); w.GP1 = take(JB * 4 * H); w.HS1 = take(JB * H); w.C1 = take(JB * H);
    w.ZERO = take((size_t)B * H);
    w.HPA = take((size_t)h->J * B * H); w.HPB = w.HPA + al256((size_t)B * H * 4);
    w.FAULT = take(64);
    w.total = o;
    return w;
}
#if EGOTAP_IN(1)
extern "C" int egotap_train_pu_saved_bytes(egotap_handle h, int B, size_t* bytes, size_t* hs1_offset) {
    EGO_CHECK(h && bytes && hs1_offset, "null argument");
    const PuSaved w = pu_saved(h, B > 0 ? B : 1);
    *bytes = w.total;
    *hs1_offset = w.HS1;
    return EGOTAP_OK;
}
#endif

// SkelNet(mode "PU") forward keeping what the backward needs: posz, rotz [B*2J, hid] -> saved[HS1] = skel_embed [J, B, H]
#if EGOTAP_IN(1)
extern "C" int egotap_train_pu_fwd(egotap_handle h, const float* posz, const float* rotz, int B, void* saved, size_t saved_bytes,
                                   void* stream) {
    EGO_CHECK(h && posz && rotz && saved, "egotap_train_pu_fwd: null argument");
    int rc = lift_resolve(h);
    if (rc != EGOTAP_OK) return rc;
    const PuSaved w = pu_saved(h, B);
    EGO_CHECK(saved_bytes >= w.total, "egotap_train_pu_fwd: saved buffer too small");
    hipStream_t s = (hipStream_t)stream;
    const LiftParams& p = h->lp;
    char* base = (char*)saved;
    auto F = [&](size_t off) { return (float*)(base + off); };
    float *F0 = F(w.F0), *G0 = F(w.G0), *HS0 = F(w.HS0), *C0 = F(w.C0), *F1 = F(w.F1), *G1 = F(w.G1), *HS1 = F(w.HS1), *C1 = F(w.C1), *ZERO = F(w.ZERO);
    float *HPA = F(w.HPA), *HPB = F(w.HPB), *GP0 = F(w.GP0), *GP1 = F(w.GP1);
    unsigned* FAULT = (unsigned*)(base + w.FAULT);
    const int J = h->J, H = h->H, hid = h->hid, JB = J * B, x = 2 * hid, NF0 = H + x;
    using Tile = TileA;
    ALoadStereo xs{posz, B, J, hid};
    EGO_HIP((gemm<Tile>(h, "pu0_x2f", xs, segmat1(p.x2f0_w, NF0, x), EpiBias{segvec1(p.x2f0_b, NF0)}, F0, NF0, JB, NF0, x, s)));
    EGO_HIP((gemm<Tile>(h, "pu0_x2h", xs, segmat1(p.x2h0_w, 4 * H, x), EpiBias{segvec1(p.x2h0_b, 4 * H)}, G0, 4L * H, JB, 4 * H, x, s)));
    {
        ALoadStereoGated bs{ALoadStereo{rotz, B, J, hid}, F0, NF0, H};
        EGO_HIP((gemm<Tile>(h, "pu0_b2h", bs, segmat1(p.b2h0_w, 4 * H, x), EpiBiasRes{segvec1(p.b2h0_b, 4 * H), G0, 4L * H}, G0, 4L * H, JB, 4 * H, x, s)));
    }
    EGO_HIP(zero_fill(ZERO, (size_t)B * H * 4, s));
    pu_chain_probe(h);
    // G0 / G1 hold Gin, GP0 / GP1 receive the full gate pre-activations; C0 / C1 keep the cell state of every step
    const PuChain ch0{F0, (long)B * NF0, NF0, G0, (long)B * 4 * H, GP0, p.h2h0_w, p.h2h0_b, C0, (long)B * H, HS0, (long)B * H, HPA, (long)B * H, B, H, J,
                      FAULT, h->pu_fault_dev};
    if (!pu_chain_launch(s, h->pu_resident[0], h->pu_resident[1], ch0, B, h->pu_debug_drop))
    for (int t = 0; t < J; ++t) {
        const float* hp_in = t == 0 ? ZERO : ((t & 1) ? HPA : HPB);
        const float* cprev = t == 0 ? ZERO : C0 + (size_t)(t - 1) * B * H;
        pu_step_launch(s, B, H, hp_in, G0 + (size_t)t * B * 4 * H, p.h2h0_w, p.h2h0_b, cprev, C0 + (size_t)t * B * H, HS0 + (size_t)t * B * H,
                       t + 1 < J ? F0 + (size_t)(t + 1) * B * NF0 : nullptr, NF0, (t & 1) ? HPB : HPA, GP0 + (size_t)t * B * 4 * H);
    }
    EGO_HIP(hipGetLastError());
    EGO_HIP((gemm<Tile>(h, "pu1_x2f", ALoadPlain{HS0, H}, segmat1(p.x2f1_w, H, H), EpiBias{segvec1(p.x2f1_b, H)}, F1, H, JB, H, H, s)));
    EGO_HIP((gemm<Tile>(h, "pu1_x2h", ALoadPlain{HS0, H}, segmat1(p.x2h1_w, 4 * H, H), EpiBias{segvec1(p.x2h1_b, 4 * H)}, G1, 4L * H, JB, 4 * H, H, s)));
    const PuChain ch1{F1, (long)B * H, H, G1, (long)B * 4 * H, GP1, p.h2h1_w, p.h2h1_b, C1, (long)B * H, HS1, (long)B * H, HPA, (long)B * H, B, H, J,
                      FAULT, h->pu_fault_dev};
    if (!pu_chain_launch(s, h->pu_resident[0], h->pu_resident[1], ch1, B, h->pu_debug_drop))
    for (int t = 0; t < J; ++t) {
        const float* hp_in = t == 0 ? ZERO : ((t & 1) ? HPA : HPB);
        const float* cprev = t == 0 ? ZERO : C1 + (size_t)(t - 1) * B * H;
        pu_step_launch(s, B, H, hp_in, G1 + (size_t)t * B * 4 * H, p.h2h1_w, p.h2h1_b, cprev, C1 + (size_t)t * B * H, HS1 + (size_t)t * B * H,
                       t + 1 < J ? F1 + (size_t)(t + 1) * B * H : nullptr, H, (t & 1) ? HPB : HPA, GP1 + (size_t)t * B * 4 * H);
    }
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

struct PuBwdWs { size_t dG, dF, HP, dHS0, dXs, dBp, DHP, dHrec[2], dC[2], WT, part, total; };
static PuBwdWs pu_bwd_ws(const Handle* h, int B) {
    PuBwdWs w;
    const size_t JB = (size_t)h->J * B, H = h->H, x = 2 * h->hid, NF0 = H + x, BH = (size_t)B * H;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = al256(o + floats * 4); return r; };
    w.dG = take(JB * 4 * H); w.dF = take(JB * NF0); w.HP = take(JB * H); w.dHS0 = take(JB * H); w.dXs = take(JB * x); w.dBp = take(JB * x);
    w.DHP = take(BH); w.dHrec[0] = take(BH); w.dHrec[1] = take(BH); w.dC[0] = take(BH); w.dC[1] = take(BH);
    w.WT = take(4 * H * H);                       // largest transposed weight: [H, 4H]
    w.part = take(4 * H * NF0 * 8 + 64 * 4 * H);  // split-M slabs of the weight-gradient GEMMs + column-sum partials
    w.total = o;
    return w;
}
#if EGOTAP_IN(1)
extern "C" int egotap_train_pu_bwd_ws_bytes(egotap_handle h, int B, size_t* bytes) {
    EGO_CHECK(h && bytes, "null argument");
    *bytes = pu_bwd_ws(h, B > 0 ? B : 1).total;
    return EGOTAP_OK;
}
#endif

// Backward of egotap_train_pu_fwd.  dhs1: gradient w.r.t. skel_embed [J,B,H]; dposz is ACCUMULATED into (the pose head's
// contribution is already there), drotz is written; grads[14] (state_dict order of skel_sequential_layer) are accumulated
// when accumulate != 0, else overwritten.
#if EGOTAP_IN(1)
extern "C" int egotap_train_pu_bwd(egotap_handle h, const float* posz, const float* rotz, int B, const void* saved,
                                   const float* dhs1, float* dposz, float* drotz, float* const* grads, int accumulate, void* ws,
                                   size_t ws_bytes, void* stream) {
    EGO_CHECK(h && posz && rotz && saved && dhs1 && dposz && drotz && grads && ws, "egotap_train_pu_bwd: null argument");
    int rc = lift_resolve(h);
    if (rc != EGOTAP_OK) return rc;
    const PuSaved sv = pu_saved(h, B);
    const PuBwdWs w = pu_bwd_ws(h, B);
    EGO_CHECK(ws_bytes >= w.total, "egotap_train_pu_bwd: workspace too small (%zu < %zu)", ws_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    const LiftParams& p = h->lp;
    const char* sb = (const char*)saved;
    auto SF = [&](size_t off) { return (const float*)(sb + off); };
    char* wb = (char*)ws;
    auto WF = [&](size_t off) { return (float*)(wb + off); };
    const float *F0 = SF(sv.F0), *G0 = SF(sv.GP0), *HS0 = SF(sv.HS0), *C0 = SF(sv.C0), *F1 = SF(sv.F1), *G1 = SF(sv.GP1), *HS1 = SF(sv.HS1),
                *C1 = SF(sv.C1), *ZERO = SF(sv.ZERO);
    float *dG = WF(w.dG), *dF = WF(w.dF), *HP = WF(w.HP), *dHS0 = WF(w.dHS0), *dXs = WF(w.dXs), *dBp = WF(w.dBp), *DHP = WF(w.DHP), *WT = WF(w.WT),
          *part = WF(w.part);
    float* dHrec[2] = {WF(w.dHrec[0]), WF(w.dHrec[1])};
    float* dC[2] = {WF(w.dC[0]), WF(w.dC[1])};
    const size_t part_bytes = w.total - w.part;
    const int J = h->J, H = h->H, hid = h->hid, JB = J * B, x = 2 * hid, NF0 = H + x;
    const size_t BH = (size_t)B * H;
    const int pw_blocks = (int)((BH + 255) / 256);
    enum { X2F0_W, X2F0_B, X2H0_W, X2H0_B, B2H0_W, B2H0_B, H2H0_W, H2H0_B, X2F1_W, X2F1_B, X2H1_W, X2H1_B, H2H1_W, H2H1_B };
    auto transpose = [&](const float* in, int R, int C) {
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, in, WT, R, C, (long)R);
    };
    auto nt = [&](const float* a, int M, int N, int K, float* y, const float* acc_src) -> hipError_t {   // y[M,N] = a[M,K] WT[N,K]^T (+ acc)
        // the per-step input gradient of the recurrence is [B, H] x K = 4H: 8 tiles of 128 x 128 at B = 256 would walk K = 2048 on 8
        // CUs (144 us, 32 times per step) -- K is split over the CUs instead, partial sums in the (idle) slab scratch
        const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (tiles < 128 && K >= 1024 && N % 128 == 0 && K % 32 == 0 && (size_t)2 * M * N * 4 <= part_bytes) {
            if (acc_src) return gemm_f32_splitk_launch<TileA>(ALoadPlain{a, K}, segmat1(WT, N, K), EpiAccum{acc_src, N}, y, N, part, part_bytes / 4, M, N, K, s);
            return gemm_f32_splitk_launch<TileA>(ALoadPlain{a, K}, segmat1(WT, N, K), EpiNone{}, y, N, part, part_bytes / 4, M, N, K, s);
        }
        if (acc_src) return gemm_f32_launch<TileA>(ALoadPlain{a, K}, segmat1(WT, N, K), EpiAccum{acc_src, N}, y, N, M, N, K, s);
        return gemm_f32_launch<TileA>(ALoadPlain{a, K}, segmat1(WT, N, K), EpiNone{}, y, N, M, N, K, s);
    };
    auto recurrence = [&](const float* Gpre, const float* Cs, const float* Hs, const float* Fs, int ldf, const float* Whh, const float* dh_ext) -> int {
        transpose(Whh, 4 * H, H);                  // WT = Whh^T viewed as [H out][4H contract]
        for (int t = J - 1; t >= 0; --t) {
            const int cur = t & 1, nxt = cur ^ 1;
            const float* cprev = t > 0 ? Cs + (size_t)(t - 1) * BH : ZERO;
            const float* hprev = t > 0 ? Hs + (size_t)(t - 1) * BH : ZERO;
            hipLaunchKernelGGL(pu_gates_bwd_kernel, dim3(pw_blocks), dim3(256), 0, s, Gpre + (size_t)t * B * 4 * H, cprev, Cs + (size_t)t * BH,
                               dh_ext + (size_t)t * BH, t == J - 1 ? nullptr : dHrec[cur], t == J - 1 ? nullptr : dC[cur],
                               dG + (size_t)t * B * 4 * H, dC[nxt], B, H);
            EGO_HIP(nt(dG + (size_t)t * B * 4 * H, B, H, 4 * H, DHP, nullptr));
            hipLaunchKernelGGL(pu_hp_bwd_kernel, dim3(pw_blocks), dim3(256), 0, s, DHP, Fs + (size_t)t * B * ldf, ldf, hprev, dHrec[nxt],
                               dF + (size_t)t * B * ldf, ldf, HP + (size_t)t * BH, B, H);
        }
        EGO_HIP(hipGetLastError());
        return EGOTAP_OK;
    };
    auto wgrad = [&](const float* dy, int N, int loader_plain_K, const float* xin, float* dwp) -> hipError_t {   // plain X [JB, K]
        return tn_any(h, dy, ALoadPlain{xin, loader_plain_K}, dwp, part, part_bytes, JB, N, loader_plain_K, accumulate, s);
    };
    auto bias = [&](const float* dy, int N, float* dbp) -> hipError_t { return colsum_f32_launch(dy, N, dbp, part, part_bytes, JB, N, accumulate, s); };

    // ---- layer 1
    rc = recurrence(G1, C1, HS1, F1, H, p.h2h1_w, dhs1);
    if (rc != EGOTAP_OK) return rc;
    EGO_HIP(wgrad(dG, 4 * H, H, HP, grads[H2H1_W]));
    EGO_HIP(bias(dG, 4 * H, grads[H2H1_B]));
    EGO_HIP(wgrad(dG, 4 * H, H, HS0, grads[X2H1_W]));
    EGO_HIP(bias(dG, 4 * H, grads[X2H1_B]));
    EGO_HIP(wgrad(dF, H, H, HS0, grads[X2F1_W]));
    EGO_HIP(bias(dF, H, grads[X2F1_B]));
    transpose(p.x2h1_w, 4 * H, H);
    EGO_HIP(nt(dG, JB, H, 4 * H, dHS0, nullptr));
    transpose(p.x2f1_w, H, H);
    EGO_HIP(nt(dF, JB, H, H, dHS0, dHS0));
    // ---- layer 0 (its outputs only feed layer 1)
    rc = recurrence(G0, C0, HS0, F0, NF0, p.h2h0_w, dHS0);
    if (rc != EGOTAP_OK) return rc;
    EGO_HIP(wgrad(dG, 4 * H, H, HP, grads[H2H0_W]));
    EGO_HIP(bias(dG, 4 * H, grads[H2H0_B]));
    // bridge branch: gates += b2h(sigmoid(fb) * bridge)
    EGO_HIP(tn_any(h, dG, ALoadStereoGated{ALoadStereo{rotz, B, J, hid}, F0, NF0, H}, grads[B2H0_W], part, part_bytes, JB, 4 * H, x, accumulate, s));
    EGO_HIP(bias(dG, 4 * H, grads[B2H0_B]));
    transpose(p.b2h0_w, 4 * H, x);
    EGO_HIP(nt(dG, JB, x, 4 * H, dBp, nullptr));
    {
        const long n = (long)JB * x;
        hipLaunchKernelGGL(pu_bridge_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dBp, F0, NF0, H, rotz, dF, drotz, B, J, hid);
    }
    // input branch: gates += x2h(x), F = x2f(x)
    EGO_HIP(tn_any(h, dG, ALoadStereo{posz, B, J, hid}, grads[X2H0_W], part, part_bytes, JB, 4 * H, x, accumulate, s));
    EGO_HIP(bias(dG, 4 * H, grads[X2H0_B]));
    EGO_HIP(tn_any(h, dF, ALoadStereo{posz, B, J, hid}, grads[X2F0_W], part, part_bytes, JB, NF0, x, accumulate, s));
    EGO_HIP(bias(dF, NF0, grads[X2F0_B]));
    transpose(p.x2h0_w, 4 * H, x);
    EGO_HIP(nt(dG, JB, x, 4 * H, dXs, nullptr));
    transpose(p.x2f0_w, NF0, x);
    EGO_HIP(nt(dF, JB, x, NF0, dXs, dXs));
    {
        const long n = (long)JB * x;
        hipLaunchKernelGGL(stereo_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dXs, dposz, B, J, hid, 1);
    }
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// pose head (training): forward = the eval kernel; backward: data gradients + weight gradients
#if EGOTAP_IN(1)
extern "C" int egotap_train_pose_head_fwd(egotap_handle h, const float* posz, const float* hs1, int B, float* pose, void* stream) {
    EGO_CHECK(h && posz && hs1 && pose, "egotap_train_pose_head_fwd: null argument");
    int rc = lift_resolve(h);
    if (rc != EGOTAP_OK) return rc;
    const LiftParams& p = h->lp;
    hipLaunchKernelGGL(pose_head_kernel, dim3(B, h->J + 1), dim3(256), 0, (hipStream_t)stream, posz, hs1, p.pose_w, p.pose_b, p.glob_w, p.glob_b, pose, B,
                       h->J, h->hid, h->H, h->cfg.estimate_head);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(1)
extern "C" int egotap_train_pose_head_bwd(egotap_handle h, const float* posz, const float* hs1, const float* dpose, int B, float* dposz,
                                          float* dhs1, float* dWp, float* dbp, float* dWg, float* dbg, int accumulate, void* stream) {
    EGO_CHECK(h && posz && hs1 && dpose && dposz && dhs1 && dWp && dbp, "egotap_train_pose_head_bwd: null argument");
    int rc = lift_resolve(h);
    if (rc != EGOTAP_OK) return rc;
    const LiftParams& p = h->lp;
    hipStream_t s = (hipStream_t)stream;
    const int J = h->J, hid = h->hid, H = h->H, eh = h->cfg.estimate_head;
    EGO_CHECK(!eh || (dWg && dbg), "egotap_train_pose_head_bwd: global_mlp gradient buffers missing");
    hipLaunchKernelGGL(pose_head_bwd_data_kernel, dim3(B), dim3(256), 0, s, dpose, p.pose_w, p.glob_w, dposz, dhs1, B, J, hid, H, eh);
    const int cols = 2 * hid + H + (eh ? J * H : 0) + 1;
    hipLaunchKernelGGL(pose_head_bwd_weight_kernel, dim3((cols + 7) / 8), dim3(256), 0, s, dpose, posz, hs1, dWp, dbp, dWg, dbg, B, J, hid,
                       H, eh, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif


// patch-embedding bias / mask-token gradients from dpos[seq, D] = sum_b dx[b]: rows of real cells -> dbias, dummy cells -> dmask
static __global__ __launch_bounds__(256) void patch_split_kernel(const float* __restrict__ dpos, float* __restrict__ dbias, float* __restrict__ dmask,
                                                          int D, int seq, int side, int ppd, int grid, int T, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= D) return;
    float sb = 0.f, sm = 0.f;
    for (int tok = 0; tok < seq; ++tok) {
        const int pr = tok / side, pc = tok - pr * side;
        const float v = dpos[(long)tok * D + n];
        if ((pr / ppd) * grid + pc / ppd >= T) sm += v; else sb += v;
    }
    dbias[n] = (accumulate ? dbias[n] : 0.f) + sb;
    dmask[n] = (accumulate ? dmask[n] : 0.f) + sm;
}
#if EGOTAP_IN(1)
extern "C" int egotap_train_patch_split(egotap_handle h, const float* dpos, float* dbias, float* dmask, int accumulate, void* stream) {
    EGO_CHECK(h && dpos && dbias && dmask, "egotap_train_patch_split: null argument");
    hipLaunchKernelGGL(patch_split_kernel, dim3((h->D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dpos, dbias, dmask, h->D, h->seq,
                       h->side, h->ppd, h->grid, h->T, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// input gradient of fc1 (position encoder): dA [B*T, ppd*ppd*D] (heatmap-major) -> dtokens [B*seq, D]; dummy tokens get zero
static __global__ __launch_bounds__(256) void tokens_scatter_kernel(const float* __restrict__ dA, float* __restrict__ dtok, int B, int T, int D,
                                                             int seq, int side, int ppd, int grid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index over [B*seq, D/4]
    const int D4 = D / 4;
    if (i >= (long)B * seq * D4) return;
    const int c4 = (int)(i % D4);
    const long bt = i / D4;
    const int tok = (int)(bt % seq), b = (int)(bt / seq);
    const int pr = tok / side, pc = tok - pr * side;
    const int cell = (pr / ppd) * grid + pc / ppd;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cell < T) {
        const int sseg = (pr % ppd) * ppd + (pc % ppd);
        v = *(const f32x4*)(dA + ((long)b * T + cell) * (long)(ppd * ppd) * D + (long)sseg * D + c4 * 4);
    }
    *(f32x4*)(dtok + bt * D + c4 * 4) = v;
}
#if EGOTAP_IN(1)
extern "C" int egotap_train_tokens_scatter(egotap_handle h, const float* dA, float* dtok, int B, void* stream) {
    EGO_CHECK(h && dA && dtok, "egotap_train_tokens_scatter: null argument");
    const long n = (long)B * h->seq * (h->D / 4);
    hipLaunchKernelGGL(tokens_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dA, dtok, B, h->T, h->D, h->seq,
                       h->side, h->ppd, h->grid);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif


// fused Q|K|V projection (three nn.Linear of ViTSelfAttention, modeling_vit.py:212-214) into one [M, 3D] buffer
#if EGOTAP_IN(1)
extern "C" int egotap_train_qkv_fwd(egotap_handle h, const float* y, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                                    const float* bv, float* qkv, int M, int D, void* stream) {
    EGO_CHECK(h && y && wq && bq && wk && bk && wv && bv && qkv, "egotap_train_qkv_fwd: null argument");
    EGO_CHECK(D % 256 == 0, "egotap_train_qkv_fwd: hidden size must be a multiple of 256");
    SegMat W; W.p[0] = wq; W.p[1] = wk; W.p[2] = wv; W.seg = D; W.ld = D;
    SegVec b; b.p[0] = bq; b.p[1] = bk; b.p[2] = bv; b.seg = D;
    EGO_HIP((gemm_big(h, "qkv", ALoadPlain{y, D}, W, EpiBias{b}, qkv, 3L * D, M, 3 * D, D, (hipStream_t)stream)));
    return EGOTAP_OK;
}
#endif


// ================================================================================================ heatmap-estimator training
// Building blocks of one optimisation step of the stage-1 model (model/heatmap_shared_model.py:98-172: HeatMap_UnrealEgo_Shared
// in train mode, MSE losses, Adam), called by the autograd glue in egotap_amd/hm_training.py.  All tensors are caller-owned
// NCHW fp32 device buffers with explicit image strides (so concat slices are read and written in place).
#include "hm_train.h"

// scratch for the repacked weights of the bf16 convolution kernels when the training operators run under a bf16 precision mode
// (egotap_hm_forward takes it from its workspace); bytes >= egotap_hmtrain_pack_bytes()
#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_set_pack_buffer(egotap_handle h, void* buf, size_t bytes) {
    EGO_CHECK(h, "null handle");
    EGO_CHECK(buf == nullptr || bytes >= conv_bf16_pack_bytes(1024, 1540), "egotap_hmtrain_set_pack_buffer: buffer too small");
    EGO_CHECK(((uintptr_t)buf & 15) == 0, "egotap_hmtrain_set_pack_buffer: 16-byte alignment");
    h->conv_pack = (__bf16*)buf;
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(2)
extern "C" size_t egotap_hmtrain_pack_bytes(void) { return conv_bf16_pack_bytes(1024, 1540); }
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_conv_fwd(egotap_handle h, const float* x, const float* w, const float* bias, const float* res, float* y,
                                       int Nimg, int Cin, int Cout, int wout, int taps, int stride, int relu, int64_t in_istride,
                                       int64_t out_istride, int64_t res_istride, void* stream) {
    EGO_CHECK(h && x && w && bias && y, "egotap_hmtrain_conv_fwd: null argument");
    ConvArgs a{x, w, y, res, nullptr, nullptr, nullptr, nullptr, bias, in_istride, out_istride, res_istride, Nimg, Cin, Cout, relu, 0, 0};
    hipError_t e = conv_any(h, "hmtrain.conv", taps, stride, wout, a, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_hmtrain_conv_fwd: unsupported conv taps=%d stride=%d wout=%d Cin=%d Cout=%d", taps, stride, wout, Cin, Cout); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

// [r3] Eval-mode building blocks for the backbones the one-call forward does not cover (the Bottleneck ResNets of --model_name resnet50 /
// resnet101, net_architecture.py:61-64): y = [relu](BatchNorm_eval(conv(x, w)) [+ res]) on the fp32 convolution kernels, BatchNorm folded
// in the epilogue exactly as egotap_hm_forward folds it (gamma / sqrt(var + 1e-5)), and the stem with its BatchNorm + ReLU.
#if EGOTAP_IN(2)
extern "C" int egotap_hm_conv_bn_fwd(egotap_handle h, const float* x, const float* w, const float* gamma, const float* beta, const float* mean,
                                     const float* var, const float* res, float* y, int Nimg, int Cin, int Cout, int wout, int taps, int stride,
                                     int relu, int64_t in_istride, int64_t out_istride, int64_t res_istride, void* stream) {
    EGO_CHECK(h && x && w && gamma && beta && mean && var && y, "egotap_hm_conv_bn_fwd: null argument");
    ConvArgs a{x, w, y, res, gamma, beta, mean, var, nullptr, in_istride, out_istride, res_istride, Nimg, Cin, Cout, relu, 0, 0};
    hipError_t e = conv_any(h, "hm.conv_bn", taps, stride, wout, a, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_hm_conv_bn_fwd: unsupported conv taps=%d stride=%d wout=%d Cin=%d Cout=%d", taps, stride, wout, Cin, Cout); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
extern "C" int egotap_hm_stem_bn_fwd(const float* left, const float* right, const float* w, const float* gamma, const float* beta, const float* mean,
                                     const float* var, float* y, int B, int S0, void* stream) {
    EGO_CHECK(left && right && w && gamma && beta && mean && var && y && B > 0 && S0 % 32 == 0, "egotap_hm_stem_bn_fwd: bad argument");
    EGO_HIP(stem_conv7_launch(left, right, w, gamma, beta, mean, var, y, S0, 2 * B, device_cu_count(), (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

// conv 7x7 / 2 of the ResNet stem without BatchNorm: z [2B, 64, S0/2, S0/2], image n = 2b + eye
#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_stem_fwd(const float* left, const float* right, const float* w, float* z, int B, int S0, void* stream) {
    EGO_CHECK(left && right && w && z && B > 0 && S0 % 32 == 0, "egotap_hmtrain_stem_fwd: bad argument");
    EGO_HIP(stem_conv7_launch(left, right, w, nullptr, nullptr, nullptr, nullptr, z, S0, 2 * B, device_cu_count(), (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

static int bn_splits(int N, int C) { int s = (1024 + C - 1) / C; if (s > N) s = N; if (s < 1) s = 1; return s; }

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_bn2d_fwd(const float* z, float* y, const float* res, const float* gamma, const float* beta, float* mean, float* rstd,
                                       float* run_mean, float* run_var, int N, int C, int HW, int64_t z_istride, int64_t y_istride,
                                       int64_t res_istride, int relu, float eps, float momentum, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(z && y && gamma && beta && mean && rstd && ws, "egotap_hmtrain_bn2d_fwd: null argument");
    EGO_CHECK(HW % 4 == 0 && N > 0 && C > 0, "egotap_hmtrain_bn2d_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int splits = bn_splits(N, C), per = (N + splits - 1) / splits;
    EGO_CHECK((size_t)splits * C * 2 * 8 <= ws_bytes, "egotap_hmtrain_bn2d_fwd: workspace too small");
    hipLaunchKernelGGL(chan_sums_kernel<0>, dim3(C, splits), dim3(256), 0, s, z, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (double*)ws, N, C, HW, (long)z_istride, (long)z_istride, 0, per);
    hipLaunchKernelGGL(bn2d_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)ws, splits, C, (double)N * HW, eps, momentum, mean, rstd,
                       run_mean, run_var);
    const long total = (long)N * C * (HW / 4);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, y, res, gamma, beta, (const float*)mean, (const float*)rstd,
                       N, C, HW, (long)z_istride, (long)y_istride, (long)res_istride, relu);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_bn2d_bwd(const float* z, const float* y, const float* dy, const float* gamma, const float* mean, const float* rstd,
                                       float* dz, float* dres, float* dgamma, float* dbeta, int N, int C, int HW, int64_t z_istride,
                                       int64_t dy_istride, int relu, int accumulate, int dres_accumulate, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(z && dy && gamma && mean && rstd && dz && dgamma && dbeta && ws, "egotap_hmtrain_bn2d_bwd: null argument");
    EGO_CHECK(!relu || y, "egotap_hmtrain_bn2d_bwd: the ReLU mask needs y");
    EGO_CHECK(HW % 4 == 0 && N > 0 && C > 0, "egotap_hmtrain_bn2d_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int splits = bn_splits(N, C), per = (N + splits - 1) / splits;
    const size_t need = (size_t)splits * C * 2 * 8 + (size_t)C * 2 * 4;
    EGO_CHECK(need <= ws_bytes, "egotap_hmtrain_bn2d_bwd: workspace too small");
    double* part = (double*)ws;
    float* sums = (float*)((char*)ws + (size_t)splits * C * 2 * 8);
    hipLaunchKernelGGL(chan_sums_kernel<1>, dim3(C, splits), dim3(256), 0, s, dy, z, y, mean, rstd, part, N, C, HW, (long)dy_istride, (long)z_istride,
                       relu, per);
    hipLaunchKernelGGL(bn2d_bwd_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)part, splits, C, sums, dgamma, dbeta, accumulate);
    const long total = (long)N * C * (HW / 4);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, y, dy, gamma, mean, rstd, (const float*)sums, dz, dres,
                       N, C, HW, (long)z_istride, (long)dy_istride, 1.0f / ((float)N * HW), relu, dres_accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

static __global__ void chansum_finish_kernel(const double* __restrict__ part, int splits, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0;
    for (int k = 0; k < splits; ++k) s0 += part[((long)k * C + c) * 2];
    out[c] = (accumulate ? out[c] : 0.f) + (float)s0;
}

// per-channel sums over N*H*W (bias gradients)
#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_chansum(const float* dy, float* out, int N, int C, int HW, int64_t istride, int accumulate, void* ws, size_t ws_bytes,
                                      void* stream) {
    EGO_CHECK(dy && out && ws && HW % 4 == 0, "egotap_hmtrain_chansum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int splits = bn_splits(N, C), per = (N + splits - 1) / splits;
    EGO_CHECK((size_t)splits * C * 2 * 8 <= ws_bytes, "egotap_hmtrain_chansum: workspace too small");
    hipLaunchKernelGGL(chan_sums_kernel<0>, dim3(C, splits), dim3(256), 0, s, dy, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (double*)ws, N, C, HW, (long)istride, (long)istride, 0, per);
    hipLaunchKernelGGL(chansum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)ws, splits, C, out, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_conv_wt(const float* w, float* wt, int Cout, int Cin, int taps, void* stream) {
    EGO_CHECK(w && wt && Cout > 0 && Cin > 0 && taps > 0, "egotap_hmtrain_conv_wt: bad argument");
    const long total = (long)Cout * Cin * taps;
    hipLaunchKernelGGL(conv_wt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, taps);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_zero_upsample(const float* in, float* out, int N, int C, int H, int64_t in_istride, int64_t out_istride, void* stream) {
    EGO_CHECK(in && out && N > 0 && C > 0 && H > 0, "egotap_hmtrain_zero_upsample: bad argument");
    const long total = (long)N * C * 4 * H * H;
    hipLaunchKernelGGL(zero_upsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long)N * C, H,
                       (long)in_istride, (long)out_istride, C);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

template <int KS, int STRIDE, int CI_T>
static hipError_t wgrad_w(int wout, const WgArgs& a, float* dw, size_t sb, int acc, hipStream_t s) {
    const int cu = device_cu_count();
    switch (wout) {
        case 128: if constexpr (KS == 7) return conv_wgrad_launch<WgCfg<KS, STRIDE, 7, CI_T, 64>>(a, dw, sb, cu, acc, s); else return hipErrorInvalidValue;
        case 64: if constexpr (KS != 7 && STRIDE == 1) return conv_wgrad_launch<WgCfg<KS, STRIDE, 6, CI_T>>(a, dw, sb, cu, acc, s); else return hipErrorInvalidValue;
        case 32: if constexpr (KS != 7) return conv_wgrad_launch<WgCfg<KS, STRIDE, 5, CI_T>>(a, dw, sb, cu, acc, s); else return hipErrorInvalidValue;
        case 16: if constexpr (KS != 7) return conv_wgrad_launch<WgCfg<KS, STRIDE, 4, CI_T>>(a, dw, sb, cu, acc, s); else return hipErrorInvalidValue;
        case 8: if constexpr (KS != 7) return conv_wgrad_launch<WgCfg<KS, STRIDE, 3, CI_T>>(a, dw, sb, cu, acc, s); else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// dW[Cout][Cin][ks][ks] (+)= sum over images and pixels of dY x shifted X   (ks in {1, 3, 7}, stride in {1, 2}, pad (ks-1)/2)
template <int NP>
static hipError_t wgrad_bf(int wout, const WgArgs& a, float* dw, size_t sb, int acc, hipStream_t s) {
    const int cu = device_cu_count();
    if (wout == 64) return conv_wgrad_bf16_launch<WgBfCfg<6, NP>>(a, dw, sb, cu, acc, s);
    if (wout == 32) return conv_wgrad_bf16_launch<WgBfCfg<5, NP>>(a, dw, sb, cu, acc, s);
    if (wout == 16) return conv_wgrad_bf16_launch<WgBfCfg<4, NP>>(a, dw, sb, cu, acc, s);
    return hipErrorInvalidValue;
}

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_conv_wgrad(const float* dy, const float* x, float* dw, int Nimg, int Cin, int Cout, int wout, int ks, int stride,
                                         int64_t dy_istride, int64_t x_istride, int accumulate, int precision, void* ws, size_t ws_bytes,
                                         void* stream) {
    EGO_CHECK(dy && x && dw && ws, "egotap_hmtrain_conv_wgrad: null argument");
    WgArgs a{dy, x, (float*)ws, dy_istride, x_istride, Nimg, Cin, Cout, 0, 0, 1, 0};
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipErrorInvalidValue;
    if (ks == 3 && stride == 1 && precision != EGOTAP_PREC_F32 && (wout == 64 || wout == 32 || wout == 16))
        e = precision == EGOTAP_PREC_BF16X3 ? wgrad_bf<3>(wout, a, dw, ws_bytes, accumulate, s) : wgrad_bf<1>(wout, a, dw, ws_bytes, accumulate, s);
    // [r3] at most 64 output channels on 64 x 64 maps (layer1, conv_heatmap): 2 x 2 waves over 64 co x 2 ci groups instead of four waves over 128 co
    else if (ks == 3 && stride == 1 && wout == 64 && Cout <= 64) e = conv_wgrad_launch<WgCfg<3, 1, 6, 64, 64, 32>>(a, dw, ws_bytes, device_cu_count(), accumulate, s);
    else if (ks == 1 && stride == 1 && wout == 64 && Cout <= 64) e = conv_wgrad_launch<WgCfg<1, 1, 6, 192, 64, 96>>(a, dw, ws_bytes, device_cu_count(), accumulate, s);
    else if (ks == 3 && stride == 1) e = wgrad_w<3, 1, 32>(wout, a, dw, ws_bytes, accumulate, s);
    else if (ks == 3 && stride == 2) e = wgrad_w<3, 2, 32>(wout, a, dw, ws_bytes, accumulate, s);
    else if (ks == 1 && stride == 1) e = wgrad_w<1, 1, 96>(wout, a, dw, ws_bytes, accumulate, s);
    else if (ks == 1 && stride == 2) e = wgrad_w<1, 2, 96>(wout, a, dw, ws_bytes, accumulate, s);
    else if (ks == 7 && stride == 2) e = wgrad_w<7, 2, 3>(wout, a, dw, ws_bytes, accumulate, s);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_hmtrain_conv_wgrad: unsupported ks=%d stride=%d wout=%d", ks, stride, wout); return EGOTAP_ERR_INVALID; }
    if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_hmtrain_conv_wgrad: workspace too small (%zu bytes)", ws_bytes); return EGOTAP_ERR_WORKSPACE; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_relu_bwd(const float* y, const float* dy, float* dz, int N, int C, int HW, int64_t y_istride, int64_t dy_istride,
                                       int64_t dz_istride, void* stream) {
    EGO_CHECK(y && dy && dz && HW % 4 == 0, "egotap_hmtrain_relu_bwd: bad argument");
    const long total = (long)N * C * (HW / 4);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dz, N, C, HW, (long)y_istride,
                       (long)dy_istride, (long)dz_istride);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_maxpool_bwd(const float* x, const float* dy, float* dx, int64_t planes, int HIN, void* stream) {
    EGO_CHECK(x && dy && dx && planes > 0 && HIN % 2 == 0, "egotap_hmtrain_maxpool_bwd: bad argument");
    const long total = planes * HIN * HIN;
    constexpr int RI = 16;
    const size_t lds = ((size_t)(RI + 3) * HIN + 2 * (size_t)(RI / 2 + 1) * (HIN / 2)) * 4;
    const long strips = planes * ((HIN + RI - 1) / RI);
    if (HIN % 4 == 0 && lds <= 60 * 1024 && strips < (1L << 31))
        hipLaunchKernelGGL(maxpool3s2_bwd_strip_kernel<RI>, dim3((unsigned)strips), dim3(256), lds, (hipStream_t)stream, x, dy, dx, (long)planes, HIN);
    else
        hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)planes, HIN);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_upsample_bwd(const float* dy, float* dx, int N, int C, int HIN, int64_t dy_istride, int64_t dx_istride, void* stream) {
    EGO_CHECK(dy && dx && N > 0 && C > 0 && HIN > 1, "egotap_hmtrain_upsample_bwd: bad argument");
    const long total = (long)N * C * HIN * HIN;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, C, HIN, (long)dy_istride,
                       (long)dx_istride);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

// loss[0] = lambda * (mean_left + mean_right) of (pred - gt)^2 / plen,  dpred = d loss / d pred   (pred, gt contiguous [B, Cn, HW])
#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_mse(const float* pred, const float* gt, const float* plen, float* dpred, float* loss, int B, int Cn, int HW,
                                  float lambda, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(pred && gt && dpred && loss && ws && HW % 4 == 0 && Cn > 0, "egotap_hmtrain_mse: bad argument");
    const int blocks = 512;
    EGO_CHECK((size_t)blocks * 8 <= ws_bytes, "egotap_hmtrain_mse: workspace too small");
    const float coef = lambda / ((float)B * (0.5f * Cn) * HW);      // two halves (left, right), each a mean over B * Cn/2 * HW
    hipLaunchKernelGGL(mse_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, gt, plen, dpred, (double*)ws, B, Cn, HW, coef);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)ws, blocks, coef, loss);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_maxpool_fwd(const float* x, float* y, int64_t planes, int HIN, void* stream) {
    EGO_CHECK(x && y && planes > 0 && HIN % 2 == 0, "egotap_hmtrain_maxpool_fwd: bad argument");
    maxpool3s2_launch(x, y, (long)planes, HIN, (hipStream_t)stream);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(2)
extern "C" int egotap_hmtrain_upsample_fwd(const float* x, float* y, int N, int C, int HIN, int64_t in_istride, int64_t out_istride, void* stream) {
    EGO_CHECK(x && y && N > 0 && C > 0 && HIN > 1, "egotap_hmtrain_upsample_fwd: bad argument");
    const long threads = (long)N * C * (2 * HIN) * (2 * HIN / 4);
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, HIN, (long)in_istride,
                       (long)out_istride);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif


// ================================================================================================ bf16-storage operators (part 3)
// EGOTAP_PREC_BF16 with bf16 tensors in HBM: activations are written as bf16 by their producers, weights are rounded once per
// step; every GEMM reads bf16 through the LDS DMA (gemm_bf16s.h).  Single operators first (tests, tools), the whole step below.
#if EGOTAP_IN(3)
int g_conv_addressing = 0;           // the one definition (gemm_bf16s64.h)
extern "C" int egotap_debug_conv_addressing(int mode) {
    EGO_CHECK(mode == 0 || mode == 1, "egotap_debug_conv_addressing: mode must be 0 (scalar origin where it fits) or 1 (per-lane pointers)");
    g_conv_addressing = mode;
    return EGOTAP_OK;
}
int g_gemm_bf16s_bk = 0;             // the one definition (gemm_bf16s64.h declares it for every part)
extern "C" int egotap_debug_gemm_bk(int bk) {
    EGO_CHECK(bk == 0 || bk == 32 || bk == 64, "egotap_debug_gemm_bk: 0 (by shape), 32 or 64");
    g_gemm_bf16s_bk = bk;
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(3)
extern "C" int egotap_bf16_gemm_nt(const void* x, int64_t ldx, const void* w, const float* bias, int M, int N, int K, int epi, const void* aux,
                                   void* out0, void* out1, int64_t ldo, void* stream) {
    EGO_CHECK(x && w && out0, "egotap_bf16_gemm_nt: null argument");
    EGO_CHECK(M >= 0 && N > 0 && K > 0 && N % 256 == 0 && K % 32 == 0, "egotap_bf16_gemm_nt: N must be a multiple of 256 and K of 32 (N=%d K=%d)", N, K);
    EGO_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)aux)) & 15) == 0,
              "egotap_bf16_gemm_nt: operands must be 16-byte aligned, leading dimensions multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const XPlain xl{(const __bf16*)x, (long)ldx};
    const __bf16* wb = (const __bf16*)w;
    const int cu = device_cu_count();
    hipError_t e;
    switch (epi) {
        case 0: e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiBf16{bias, (__bf16*)out0, (long)ldo}, M, N, K, cu, s); break;
        case 1: EGO_CHECK(bias && aux, "egotap_bf16_gemm_nt: epi 1 needs bias and residual");
                e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiResF32{bias, (const float*)aux, (float*)out0, (long)ldo}, M, N, K, cu, s); break;
        case 2: EGO_CHECK(bias && out1, "egotap_bf16_gemm_nt: epi 2 needs bias and the second output");
                e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiGeluSave{bias, (__bf16*)out0, (__bf16*)out1, (long)ldo}, M, N, K, cu, s); break;
        case 3: EGO_CHECK(aux, "egotap_bf16_gemm_nt: epi 3 needs the saved pre-activation");
                // out1 (optional): fp32 [2 * ceil(M / 256)][N] partial column sums of the stored output, one row per 128-row wave block
                if (out1) e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiGeluGradCS{{(const __bf16*)aux, (__bf16*)out0, (long)ldo}, (float*)out1}, M, N, K, cu, s);
                else e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiGeluGrad{(const __bf16*)aux, (__bf16*)out0, (long)ldo}, M, N, K, cu, s);
                break;
        case 4: EGO_CHECK(bias, "egotap_bf16_gemm_nt: epi 4 needs bias");
                e = gemm_bf16s_plain_launch(xl, wb, (long)K, SEpiF32{bias, (float*)out0, (long)ldo}, M, N, K, cu, s); break;
        default: egotap_set_error("egotap_bf16_gemm_nt: unknown epilogue %d", epi); return EGOTAP_ERR_INVALID;
    }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_gemm_tn(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, int M, int N, int K, int accumulate,
                                   const void* zeros, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(dy && x && dw && zeros, "egotap_bf16_gemm_tn: null argument");
    EGO_CHECK(M > 0 && N % 256 == 0 && K % 256 == 0, "egotap_bf16_gemm_tn: N and K must be multiples of 256 (N=%d K=%d)", N, K);
    EGO_CHECK(ldy % 8 == 0 && ldx % 8 == 0 && ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)zeros | (uintptr_t)ws)) & 15) == 0,
              "egotap_bf16_gemm_tn: operands must be 16-byte aligned, leading dimensions multiples of 8");
    hipError_t e = gemm_tn_bf16s_launch((const __bf16*)dy, (long)ldy, TXPlain{(const __bf16*)x, (long)ldx}, (const __bf16*)zeros, dw, (float*)ws, ws_bytes, M, N, K,
                                        device_cu_count(), accumulate, (hipStream_t)stream);
    if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_bf16_gemm_tn: workspace too small for one partial slab (%zu bytes given)", ws_bytes); return EGOTAP_ERR_WORKSPACE; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_layernorm_fwd(const float* x, void* y, const float* g, const float* b, float* mean, float* rstd, int rows, float eps, void* stream) {
    EGO_CHECK(x && y && g && b, "egotap_bf16_layernorm_fwd: null argument");
    if (rows <= 0) return EGOTAP_OK;
    hipLaunchKernelGGL(ln_fwd_bf16_kernel<1024>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, g, b, mean, rstd, rows, eps);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
// ws >= (3 * blocks + 3 + 3 * ceil(blocks / 64)) * 1024 floats, blocks = ceil(rows / 64)
extern "C" int egotap_bf16_layernorm_bwd(const float* x, const void* dy, const float* g, const float* mean, const float* rstd, const float* dres, float* dx,
                                         void* dxb, float* dgamma, float* dbeta, float* dcolsum, int rows, int accumulate, void* ws, size_t ws_bytes,
                                         void* stream) {
    EGO_CHECK(x && dy && g && mean && rstd && dx && dgamma && dbeta && ws, "egotap_bf16_layernorm_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int rows_per_wave = 16, blocks = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
    const int gy = (blocks + 63) / 64;
    EGO_CHECK(ws_bytes >= ((size_t)blocks * 3072 + 3072 + (size_t)gy * 3072) * 4, "egotap_bf16_layernorm_bwd: workspace too small");
    float* part = (float*)ws;
    hipLaunchKernelGGL(ln_bwd_bf16_kernel<1024>, dim3(blocks), dim3(256), 0, s, x, (const __bf16*)dy, g, mean, rstd, dres, dx, (__bf16*)dxb, part, rows, rows_per_wave);
    EGO_HIP(hipGetLastError());
    float* tmp = part + (size_t)blocks * 3072;
    float* part2 = tmp + 3072;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(3072 / 4 / 64, gy), dim3(256), 0, s, (const float*)part, 3072L, part2, blocks, 3072, 64);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(3), dim3(256), 0, s, (const float*)part2, tmp, 3072L, gy, 0);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(256), 0, s, tmp, dgamma, 1024L, 1, accumulate);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(256), 0, s, tmp + 1024, dbeta, 1024L, 1, accumulate);
    if (dcolsum) hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1), dim3(256), 0, s, tmp + 2048, dcolsum, 1024L, 1, accumulate);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_colsum(const void* y, int64_t ldy, float* out, int M, int N, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(y && out && ws && N % 8 == 0 && ldy % 8 == 0, "egotap_bf16_colsum: bad argument");
    hipError_t e = colsum_bf16_launch((const __bf16*)y, (long)ldy, out, M, N, accumulate, (float*)ws, ws_bytes, (hipStream_t)stream);
    if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_bf16_colsum: workspace too small"); return EGOTAP_ERR_WORKSPACE; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_prep_weight(const float* w, void* wb, void* wt, int N, int K, int64_t ldt, void* stream) {
    EGO_CHECK(w && wb && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "egotap_bf16_prep_weight: bad argument (N, K multiples of 8)");
    EGO_CHECK(wt == nullptr || (ldt >= N && ldt % 8 == 0), "egotap_bf16_prep_weight: ldt must be >= N and a multiple of 8");
    hipLaunchKernelGGL(prep_weight_kernel, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wb, (__bf16*)wt, N, K, (long)ldt);
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_from_f32(const float* src, void* dst, int64_t n, void* stream) {
    EGO_CHECK(src && dst && n % 8 == 0, "egotap_bf16_from_f32: n must be a multiple of 8");
    if (n == 0) return EGOTAP_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, (long)(n / 8));
    EGO_HIP(hipGetLastError());
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int N, int heads, void* stream) {
    EGO_CHECK(qkv && ctx, "egotap_bf16_attention_fwd: null argument");
    hipError_t e = attention_bf16s_fwd_launch((const __bf16*)qkv, (__bf16*)ctx, lse, B, N, heads, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_bf16_attention_fwd: sequence length %d is not a multiple of 32", N); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, int B, int N, int heads,
                                         void* stream) {
    EGO_CHECK(qkv && ctx && dctx && lse && delta && dqkv, "egotap_bf16_attention_bwd: null argument");
    hipError_t e = attention_bf16s_bwd_launch((const __bf16*)qkv, (const __bf16*)ctx, (const __bf16*)dctx, lse, delta, (__bf16*)dqkv, B, N, heads, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_bf16_attention_bwd: sequence length %d is not a multiple of 32", N); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
// the same, also producing the q | k | v BIAS gradients (column sums of dqkv) without a pass over dqkv: the kernels' epilogues leave
// per-block partial sums in ws (fp32 [B * N / 32][3 * heads * 128]); three small fp32 column sums finish them.  Shapes
// without that epilogue (N % 64 != 0) take the column-sum pass over dqkv instead.
extern "C" int egotap_bf16_attention_bwd_bias(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, float* dq_bias,
                                              float* dk_bias, float* dv_bias, int B, int N, int heads, void* ws, size_t ws_bytes, void* stream) {
    EGO_CHECK(qkv && ctx && dctx && lse && delta && dqkv && dq_bias && dk_bias && dv_bias && ws, "egotap_bf16_attention_bwd_bias: null argument");
    const int D = heads * 128, M = B * N;
    float* db[3] = {dq_bias, dk_bias, dv_bias};
    const size_t part_bytes = (size_t)B * (N / 32) * 3 * D * 4;
    const bool fused = N % 64 == 0 && ws_bytes >= part_bytes + (64u << 20);
    hipError_t e = attention_bf16s_bwd_launch((const __bf16*)qkv, (const __bf16*)ctx, (const __bf16*)dctx, lse, delta, (__bf16*)dqkv, B, N, heads, (hipStream_t)stream,
                                              fused ? (float*)ws : nullptr);
    if (e == hipErrorInvalidValue) { egotap_set_error("egotap_bf16_attention_bwd_bias: sequence length %d is not a multiple of 32", N); return EGOTAP_ERR_INVALID; }
    EGO_HIP(e);
    for (int q = 0; q < 3; ++q) {
        if (fused) {
            EGO_HIP(colsum_f32_launch((const float*)ws + (size_t)q * D, 3L * D, db[q], (float*)((char*)ws + part_bytes), ws_bytes - part_bytes, B * (N / 32), D, 0,
                                      (hipStream_t)stream));
        } else {
            hipError_t e2 = colsum_bf16_launch((const __bf16*)dqkv + (size_t)q * D, 3L * D, db[q], M, D, 0, (float*)ws, ws_bytes, (hipStream_t)stream);
            if (e2 == hipErrorOutOfMemory) { egotap_set_error("egotap_bf16_attention_bwd_bias: workspace too small"); return EGOTAP_ERR_WORKSPACE; }
            EGO_HIP(e2);
        }
    }
    return EGOTAP_OK;
}
#endif

// fc1 of the two heatmap encoders on bf16 operands, the gathers folded into the loaders (net_architecture.py:388-406, 690-694):
//   which 0: rows = per-heatmap patch tokens gathered from tokens bf16 [B*seq, D];  1: rows = [cos | sin] maps from hm bf16 [B, C, S, S]
#if EGOTAP_IN(3)
// [r3] patch embedding of the bf16-storage step on the bf16-storage GEMM: hmb = bf16 copy of the heatmaps [B, C, S, S], w = bf16 copy of
// projection.weight [D, 256] (egotap_bf16_prep_weight), zeros >= 16 bytes of zeros; x fp32 [B * seq, D] = the embeddings
// (net_architecture.py:326-336, modeling_vit.py:137-153).  Replaces the fp32-operand kernel of the opt-in modes (VALU conversion per element).
extern "C" int egotap_bf16_patch_fwd(egotap_handle h, const void* hmb, const void* w, const float* bias, const float* mask_tok, const float* pos,
                                     const void* zeros, float* x, int B, void* stream) {
    EGO_CHECK(h && hmb && w && bias && mask_tok && pos && zeros && x && B > 0, "egotap_bf16_patch_fwd: bad argument");
    const int S = h->cfg.hm_size, D = h->D, M = B * h->seq;
    const XPatch xl{(const __bf16*)hmb, (const __bf16*)zeros, h->C, S, h->seq, h->side, h->ppd, h->grid, h->T};
    const SEpiPatchF32 ep{bias, mask_tok, pos, x, D, h->seq, h->side, h->ppd, h->grid, h->T};
    EGO_HIP(gemm_bf16s_launch(xl, (const __bf16*)w, 256L, ep, M, D, 256, device_cu_count(), (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif
#if EGOTAP_IN(3)
extern "C" int egotap_bf16_fc1_fwd(egotap_handle h, int which, const void* src, const void* w, const float* bias, float* z, int B, void* stream) {
    EGO_CHECK(h && src && w && bias && z && (which == 0 || which == 1), "egotap_bf16_fc1_fwd: bad argument");
    const int BT = B * h->T, S = h->cfg.hm_size, K = which == 0 ? h->ppd * h->ppd * h->D : 2 * S * S;
    EGO_HIP(fc1_nt(h, which, (const __bf16*)src, (const __bf16*)w, (long)K, SEpiF32{bias, z, 2048L}, BT, device_cu_count(), (hipStream_t)stream));
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
extern "C" int egotap_bf16_fc1_wgrad(egotap_handle h, int which, const void* dz, const void* src, float* dw, int B, const void* zeros, void* ws,
                                     size_t ws_bytes, void* stream) {
    EGO_CHECK(h && dz && src && dw && zeros && ws && (which == 0 || which == 1), "egotap_bf16_fc1_wgrad: bad argument");
    const int BT = B * h->T, S = h->cfg.hm_size, K = which == 0 ? h->ppd * h->ppd * h->D : 2 * S * S;
    hipError_t e;
    if (which == 0) e = gemm_tn_bf16s_launch((const __bf16*)dz, 2048L, TXTokens{(const __bf16*)src, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, (const __bf16*)zeros, dw, (float*)ws, ws_bytes, BT, 2048, K, device_cu_count(), 0, (hipStream_t)stream);
    else e = gemm_tn_bf16s_launch((const __bf16*)dz, 2048L, TXRot{(const __bf16*)src, h->C, h->J, S * S}, (const __bf16*)zeros, dw, (float*)ws, ws_bytes, BT, 2048, K, device_cu_count(), 0, (hipStream_t)stream);
    if (e == hipErrorOutOfMemory) { egotap_set_error("egotap_bf16_fc1_wgrad: workspace too small"); return EGOTAP_ERR_WORKSPACE; }
    EGO_HIP(e);
    return EGOTAP_OK;
}
#endif

#if EGOTAP_IN(3)
// input gradient of fc1 of the position encoder, scattered back to token order: dtok bf16 [B*seq, D] (cleared here: dummy cells get zero)
extern "C" int egotap_bf16_fc1_dgrad_tokens(egotap_handle h, const void* dz, const void* wt, void* dtok, int B, void* stream) {
    EGO_CHECK(h && dz && wt && dtok, "egotap_bf16_fc1_dgrad_tokens: null argument");
    const int BT = B * h->T, K1 = h->ppd * h->ppd * h->D;
    hipStream_t s = (hipStream_t)stream;
    EGO_HIP(zero_fill(dtok, (size_t)B * h->seq * h->D * 2, s));
    EGO_HIP(gemm_bf16s_plain_launch(XPlain{(const __bf16*)dz, 2048L}, (const __bf16*)wt, 2048L, SEpiScatterTokens{(__bf16*)dtok, h->T, h->D, h->seq, h->side, h->ppd, h->grid}, BT, K1,
                                    2048, device_cu_count(), s));      // [r5] the 64-deep kernel where its shape rules hold (plain operand)
    return EGOTAP_OK;
}
#endif

// ------------------------------------------------------------------------------------------------ one-call training step
// egotap_lift_forward_train + egotap_lift_backward: the lifting head's training-mode forward (activations kept in `saved`) and its
// whole backward as ONE call each (egotap_lift_forward(…, saved, …) / egotap_lift_backward; reference:
// egotap_autoencoder_model.py:284-311 loss.backward()).  They compose the granular training operators above in the order
// egotap_amd/training.py used to (the per-operator tests keep calling those), read the parameters bound with egotap_bind_param
// and write the gradients into the buffers bound with egotap_bind_grad.  fp32 tensors; arithmetic of the large GEMMs follows
// egotap_set_precision (f32 / bf16x3; bf16 here means bf16 operand copies, the bf16-STORAGE step is egotap_amd/training.py's
// LiftTrainBf16Fn).  Everything is enqueued on the caller's stream; no allocation, no synchronisation.
#if EGOTAP_IN(0)
struct LiftTrainPlan {      // byte offsets into `saved`
    size_t X[9];            // X[i]: input of ViT layer i (X[0] = embeddings), X[L]: input of the final LayerNorm
    struct Layer { size_t m1, r1, y1, qkv, ctx, lse, xm, m2, r2, y2, z, hid; } layer[8];
    size_t mf, rf, tokens;
    struct Fc { size_t z, y, mean, rstd; } pos[3], rot[3];
    size_t pu, pu_hs1, pu_bytes, total;
};
struct LiftBwdPlan {        // byte offsets into the backward workspace
    size_t R[3], A4, A3, WT, E[2], dposz, drotz, dhs1, delta, pu, pu_bytes, scr, scr_bytes, total;
};
static const int FC_OUT[2] = {2048, 512};

// the bf16-STORAGE step (EGOTAP_PREC_BF16 with vit_dim 1024: BASELINE config 3, the wrapper's --use_amp): bf16 activations and per-step
// bf16 weight copies, everything else as above.  Byte offsets into `saved` / the workspace.
struct LiftTrain16Plan {
    size_t hmb, X[9];
    struct Layer { size_t m1, r1, y1, qkv, ctx, lse, xm, m2, r2, y2, z, hid, w_qkv, w_qkv_t, w_o, w_o_t, w_up, w_up_t, w_dn, w_dn_t, b_qkv; } layer[8];
    size_t mf, rf, tokens, w_fc1p, w_fc1p_t, w_fc1r;
    LiftTrainPlan::Fc pos[3], rot[3];
    size_t pu, pu_hs1, pu_bytes, total;
};
struct LiftBwd16Plan {
    size_t F[2], Rb[2], A4b, A3b, E[2], dzb, WT, dposz, drotz, dhs1, delta, zero, pu, pu_bytes, scr, scr_bytes, total;
};
static bool lift_train_bf16s(const Handle* h) { return h->precision == EGOTAP_PREC_BF16 && h->D == 1024; }

static int lift_train16_plan(Handle* h, int B, LiftTrain16Plan& t, LiftBwd16Plan& w) {
    const size_t M = (size_t)B * h->seq, D = h->D, BT = (size_t)B * h->T, heads = h->cfg.vit_heads, L = h->cfg.vit_layers;
    const size_t K1 = (size_t)h->ppd * h->ppd * D, K1r = 2 * (size_t)h->cfg.hm_size * h->cfg.hm_size;
    size_t o = 0;
    auto bytes = [&](size_t n) { size_t r = o; o = al256(o + n); return r; };
    auto f32 = [&](size_t n) { return bytes(n * 4); };
    auto b16 = [&](size_t n) { return bytes(n * 2); };
    t.hmb = b16((size_t)B * h->C * h->cfg.hm_size * h->cfg.hm_size);
    for (size_t i = 0; i <= L; ++i) t.X[i] = f32(M * D);
    for (size_t i = 0; i < L; ++i) {
        auto& l = t.layer[i];
        l.m1 = f32(M); l.r1 = f32(M); l.y1 = b16(M * D); l.qkv = b16(M * 3 * D); l.ctx = b16(M * D); l.lse = f32((size_t)B * heads * h->seq);
        l.xm = f32(M * D); l.m2 = f32(M); l.r2 = f32(M); l.y2 = b16(M * D); l.z = b16(M * 4 * D); l.hid = b16(M * 4 * D);
        l.w_qkv = b16(3 * D * D); l.w_qkv_t = b16(3 * D * D); l.w_o = b16(D * D); l.w_o_t = b16(D * D);
        l.w_up = b16(4 * D * D); l.w_up_t = b16(4 * D * D); l.w_dn = b16(4 * D * D); l.w_dn_t = b16(4 * D * D); l.b_qkv = f32(3 * D);
    }
    t.mf = f32(M); t.rf = f32(M); t.tokens = b16(M * D);
    t.w_fc1p = b16(2048 * K1); t.w_fc1p_t = b16(2048 * K1); t.w_fc1r = b16(2048 * K1r);
    for (int e = 0; e < 2; ++e)
        for (int j = 0; j < 3; ++j) {
            const size_t n = j < 2 ? (size_t)FC_OUT[j] : (size_t)h->hid;
            LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            f.z = f32(BT * n); f.y = f32(BT * n); f.mean = f32(n); f.rstd = f32(n);
        }
    size_t pub = 0, hs1 = 0;
    int rc = egotap_train_pu_saved_bytes(h, B, &pub, &hs1);
    if (rc != EGOTAP_OK) return rc;
    t.pu = bytes(pub + 4); t.pu_hs1 = hs1; t.pu_bytes = pub;
    t.total = o;

    o = 0;
    w.F[0] = f32(M * D); w.F[1] = f32(M * D); w.Rb[0] = b16(M * D); w.Rb[1] = b16(M * D);
    w.A4b = b16(M * 4 * D); w.A3b = w.A4b;       // dz (MLP) is dead before dqkv (attention) is produced: one slot
    w.E[0] = f32(BT * 2048); w.E[1] = f32(BT * 2048); w.dzb = b16(BT * 2048);
    w.WT = f32((size_t)2048 * 512);
    w.dposz = f32(BT * h->hid); w.drotz = f32(BT * h->hid); w.dhs1 = f32((size_t)h->J * B * h->H);
    w.delta = f32((size_t)B * heads * h->seq);
    w.zero = bytes(4096);
    size_t pwb = 0;
    rc = egotap_train_pu_bwd_ws_bytes(h, B, &pwb);
    if (rc != EGOTAP_OK) return rc;
    w.pu = bytes(pwb + 4); w.pu_bytes = pwb;
    const size_t nb = (M + 63) / 64;
    w.scr_bytes = std::max(std::max((size_t)4 * 4 * D * D * 8, (size_t)4 * 2048 * K1 * 2), (3 * nb + 3 + 3 * ((nb + 63) / 64)) * 4096 + 4096);
    w.scr_bytes = std::max(w.scr_bytes, (size_t)64 << 20);
    w.scr_bytes = std::max(w.scr_bytes, (size_t)B * (h->seq / 32) * 3 * D * 4 + ((size_t)64 << 20));      // per-block bias-gradient sums of the attention backward
    w.scr = bytes(w.scr_bytes);
    w.total = o;
    return EGOTAP_OK;
}

static int lift_train_plan(Handle* h, int B, LiftTrainPlan& t, LiftBwdPlan& w) {
    const size_t M = (size_t)B * h->seq, D = h->D, BT = (size_t)B * h->T, heads = h->cfg.vit_heads, L = h->cfg.vit_layers;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = al256(o + floats * 4); return r; };
    for (size_t i = 0; i <= L; ++i) t.X[i] = take(M * D);
    for (size_t i = 0; i < L; ++i) {
        auto& l = t.layer[i];
        l.m1 = take(M); l.r1 = take(M); l.y1 = take(M * D); l.qkv = take(M * 3 * D); l.ctx = take(M * D); l.lse = take((size_t)B * heads * h->seq);
        l.xm = take(M * D); l.m2 = take(M); l.r2 = take(M); l.y2 = take(M * D); l.z = take(M * 4 * D); l.hid = take(M * 4 * D);
    }
    t.mf = take(M); t.rf = take(M); t.tokens = take(M * D);
    for (int e = 0; e < 2; ++e)
        for (int j = 0; j < 3; ++j) {
            const size_t n = j < 2 ? (size_t)FC_OUT[j] : (size_t)h->hid;
            LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            f.z = take(BT * n); f.y = take(BT * n); f.mean = take(n); f.rstd = take(n);
        }
    size_t pub = 0, hs1 = 0;
    int rc = egotap_train_pu_saved_bytes(h, B, &pub, &hs1);
    if (rc != EGOTAP_OK) return rc;
    t.pu = take(pub / 4 + 1); t.pu_hs1 = hs1; t.pu_bytes = pub;
    t.total = o;

    o = 0;
    const size_t K1 = (size_t)h->ppd * h->ppd * D, K1r = 2 * (size_t)h->cfg.hm_size * h->cfg.hm_size;
    for (int i = 0; i < 3; ++i) w.R[i] = take(M * D);
    w.A4 = take(M * 4 * D); w.A3 = take(M * 3 * D);
    w.WT = take(std::max((size_t)4 * D * D, K1 * 2048));
    w.E[0] = take(BT * 2048); w.E[1] = take(BT * 2048);
    w.dposz = take(BT * h->hid); w.drotz = take(BT * h->hid); w.dhs1 = take((size_t)h->J * B * h->H);
    w.delta = take((size_t)B * heads * h->seq);
    size_t pwb = 0;
    rc = egotap_train_pu_bwd_ws_bytes(h, B, &pwb);
    if (rc != EGOTAP_OK) return rc;
    w.pu = take(pwb / 4 + 1); w.pu_bytes = pwb;
    // split-M slabs of the weight-gradient GEMMs (8 slabs of the largest [N, K]) / column-sum partials / LayerNorm partials
    w.scr_bytes = std::max((size_t)64 << 20, (size_t)4 * 2048 * std::max(K1, K1r) * 8);
    w.scr = take(w.scr_bytes / 4);
    w.total = o;
    return EGOTAP_OK;
}

extern "C" int egotap_lift_train_bytes(egotap_handle h, int B, size_t* saved_bytes, size_t* ws_bytes) {
    EGO_CHECK(h && saved_bytes && ws_bytes && B > 0, "egotap_lift_train_bytes: bad argument");
    if (lift_train_bf16s(h)) {
        LiftTrain16Plan t; LiftBwd16Plan w;
        const int rc = lift_train16_plan(h, B, t, w);
        if (rc != EGOTAP_OK) return rc;
        *saved_bytes = t.total;
        *ws_bytes = w.total;
        return EGOTAP_OK;
    }
    LiftTrainPlan t; LiftBwdPlan w;
    const int rc = lift_train_plan(h, B, t, w);
    if (rc != EGOTAP_OK) return rc;
    *saved_bytes = t.total;
    *ws_bytes = w.total;
    return EGOTAP_OK;
}

extern "C" int egotap_bind_grad(egotap_handle h, const char* key, void* dev_ptr, int64_t numel) {
    EGO_CHECK(h && key && dev_ptr, "egotap_bind_grad: null argument");
    auto it = h->bound[EGOTAP_NET_LIFT].find(key);
    EGO_CHECK(it != h->bound[EGOTAP_NET_LIFT].end(), "egotap_bind_grad(%s): bind the parameter first", key);
    EGO_CHECK(it->second.numel == numel, "egotap_bind_grad(%s): %lld elements, the parameter has %lld", key, (long long)numel, (long long)it->second.numel);
    EGO_CHECK(((uintptr_t)dev_ptr & 15) == 0, "egotap_bind_grad(%s): pointer must be 16-byte aligned", key);
    Param p;
    p.ptr = dev_ptr; p.numel = numel; p.dtype = EGOTAP_F32;
    h->bound_grad[key] = p;
    h->grad_resolved = false;
    return EGOTAP_OK;
}

#define EGO_RC(call) do { const int rc_ = (call); if (rc_ != EGOTAP_OK) return rc_; } while (0)

static int lift_forward_train16(Handle* h, const float* hm, int B, float* pose, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                                void* stream) {
    LiftTrain16Plan t; LiftBwd16Plan w;
    EGO_RC(lift_train16_plan(h, B, t, w));
    EGO_CHECK(saved_bytes >= t.total, "egotap_lift_forward_train: saved buffer too small (%zu < %zu)", saved_bytes, t.total);
    EGO_CHECK(ws_bytes >= w.total, "egotap_lift_forward_train: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const LiftParams& p = h->lp;
    char* sb = (char*)saved;
    auto S = [&](size_t off) { return (float*)(sb + off); };
    auto Hb = [&](size_t off) { return (void*)(sb + off); };
    void* scr = (char*)ws + w.scr;
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->seq, D = h->D, BT = B * h->T, heads = h->cfg.vit_heads, L = h->cfg.vit_layers;
    const int K1 = h->ppd * h->ppd * D, K1r = 2 * h->cfg.hm_size * h->cfg.hm_size;
    // per-step bf16 copies of the GEMM weights (and transposed copies for the input-gradient GEMMs), kept with the activations
    for (int i = 0; i < L; ++i) {
        const auto& P_ = p.layer[i];
        const auto& l = t.layer[i];
        const float* qkvw[3] = {P_.q_w, P_.k_w, P_.v_w};
        for (int q = 0; q < 3; ++q)
            EGO_RC(egotap_bf16_prep_weight(qkvw[q], (__bf16*)Hb(l.w_qkv) + (size_t)q * D * D, (__bf16*)Hb(l.w_qkv_t) + (size_t)q * D, D, D, 3 * D, stream));
        EGO_RC(egotap_bf16_prep_weight(P_.o_w, Hb(l.w_o), Hb(l.w_o_t), D, D, D, stream));
        EGO_RC(egotap_bf16_prep_weight(P_.up_w, Hb(l.w_up), Hb(l.w_up_t), 4 * D, D, 4 * D, stream));
        EGO_RC(egotap_bf16_prep_weight(P_.dn_w, Hb(l.w_dn), Hb(l.w_dn_t), D, 4 * D, D, stream));
        hipLaunchKernelGGL(concat3_kernel, dim3((D + 255) / 256), dim3(256), 0, s, P_.q_b, P_.k_b, P_.v_b, S(l.b_qkv), D);
        EGO_HIP(hipGetLastError());
    }
    EGO_RC(egotap_bf16_prep_weight(p.pos_fc[0].w, Hb(t.w_fc1p), Hb(t.w_fc1p_t), 2048, K1, 2048, stream));
    EGO_RC(egotap_bf16_prep_weight(p.rot_fc[0].w, Hb(t.w_fc1r), nullptr, 2048, K1r, 2048, stream));
    EGO_RC(egotap_bf16_from_f32(hm, Hb(t.hmb), (int64_t)B * h->C * h->cfg.hm_size * h->cfg.hm_size, stream));
    {   // [r3] patch embedding on the bf16-storage GEMM (bf16 heatmaps by LDS DMA); its bf16 weight copy (512 KB) and a page of zeros live in
        // the step's scratch for the duration of the launch (the weight gradient still reads the fp32 heatmaps: egotap_lift_backward)
        char* pscr = (char*)ws + w.scr;
        EGO_HIP(zero_fill(pscr, 256, s));
        EGO_RC(egotap_bf16_prep_weight(p.patch_w, pscr + 256, nullptr, D, 256, D, stream));
        EGO_RC(egotap_bf16_patch_fwd(h, Hb(t.hmb), pscr + 256, p.patch_b, p.mask_tok, p.pos_emb, pscr, S(t.X[0]), B, stream));
    }
    for (int i = 0; i < L; ++i) {
        const auto& P_ = p.layer[i];
        const auto& l = t.layer[i];
        float* x = S(t.X[i]);
        EGO_RC(egotap_bf16_layernorm_fwd(x, Hb(l.y1), P_.ln1_g, P_.ln1_b, S(l.m1), S(l.r1), M, 1e-12f, stream));
        EGO_RC(egotap_bf16_gemm_nt(Hb(l.y1), D, Hb(l.w_qkv), S(l.b_qkv), M, 3 * D, D, 0, nullptr, Hb(l.qkv), nullptr, 3 * D, stream));
        EGO_RC(egotap_bf16_attention_fwd(Hb(l.qkv), Hb(l.ctx), S(l.lse), B, h->seq, heads, stream));
        EGO_RC(egotap_bf16_gemm_nt(Hb(l.ctx), D, Hb(l.w_o), P_.o_b, M, D, D, 1, x, S(l.xm), nullptr, D, stream));
        EGO_RC(egotap_bf16_layernorm_fwd(S(l.xm), Hb(l.y2), P_.ln2_g, P_.ln2_b, S(l.m2), S(l.r2), M, 1e-12f, stream));
        EGO_RC(egotap_bf16_gemm_nt(Hb(l.y2), D, Hb(l.w_up), P_.up_b, M, 4 * D, D, 2, nullptr, Hb(l.z), Hb(l.hid), 4 * D, stream));
        EGO_RC(egotap_bf16_gemm_nt(Hb(l.hid), 4 * D, Hb(l.w_dn), P_.dn_b, M, D, 4 * D, 1, S(l.xm), S(t.X[i + 1]), nullptr, D, stream));
    }
    EGO_RC(egotap_bf16_layernorm_fwd(S(t.X[L]), Hb(t.tokens), p.lnf_g, p.lnf_b, S(t.mf), S(t.rf), M, 1e-12f, stream));
    for (int e = 0; e < 2; ++e) {
        const float* a_in = nullptr;
        int K = 0;
        for (int j = 0; j < 3; ++j) {
            const int n = j < 2 ? FC_OUT[j] : h->hid;
            const LiftParams::Fc& fc = e == 0 ? p.pos_fc[j] : p.rot_fc[j];
            const LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            if (j == 0) EGO_RC(egotap_bf16_fc1_fwd(h, e, e == 0 ? Hb(t.tokens) : Hb(t.hmb), e == 0 ? Hb(t.w_fc1p) : Hb(t.w_fc1r), fc.b, S(f.z), B, stream));
            else EGO_RC(egotap_train_gemm_nt(h, 0, a_in, 0, nullptr, fc.w, fc.b, S(f.z), BT, n, K, 1, nullptr, nullptr, 0, stream));
            EGO_RC(egotap_train_bn_lrelu_fwd(S(f.z), S(f.y), fc.g, fc.beta, S(f.mean), S(f.rstd), (float*)fc.mean, (float*)fc.var, BT, n, 1e-5f,
                                             0.1f, scr, w.scr_bytes, stream));
            a_in = S(f.y); K = n;
        }
    }
    EGO_RC(egotap_train_pu_fwd(h, S(t.pos[2].y), S(t.rot[2].y), B, sb + t.pu, t.pu_bytes, stream));
    EGO_RC(egotap_train_pose_head_fwd(h, S(t.pos[2].y), (const float*)(sb + t.pu + t.pu_hs1), B, pose, stream));
    return EGOTAP_OK;
}

static int lift_backward16(Handle* h, const float* hm, const float* dpose, int B, const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                           void* const* bucket_events, int n_events, void* stream) {
    LiftTrain16Plan t; LiftBwd16Plan w;
    EGO_RC(lift_train16_plan(h, B, t, w));
    EGO_CHECK(saved_bytes >= t.total, "egotap_lift_backward: saved buffer too small (%zu < %zu)", saved_bytes, t.total);
    EGO_CHECK(ws_bytes >= w.total, "egotap_lift_backward: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const int L = h->cfg.vit_layers;
    const LiftParams& p = h->lp;
    const LiftParams& g = h->lg;
    auto G = [](const float* q) { return (float*)q; };
    const char* sb = (const char*)saved;
    auto S = [&](size_t off) { return (const float*)(sb + off); };
    auto Hb = [&](size_t off) { return (const void*)(sb + off); };
    char* wb = (char*)ws;
    auto W = [&](size_t off) { return (float*)(wb + off); };
    auto Wh = [&](size_t off) { return (void*)(wb + off); };
    void* scr = wb + w.scr;
    const size_t scrb = w.scr_bytes;
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->seq, D = h->D, BT = B * h->T, heads = h->cfg.vit_heads;
    const void* ZERO = wb + w.zero;
    EGO_HIP(zero_fill(wb + w.zero, 4096, s));
    int bucket = 0;
    auto bucket_done = [&]() -> int {
        if (n_events) EGO_HIP(hipEventRecord((hipEvent_t)bucket_events[bucket], s));
        ++bucket;
        return EGOTAP_OK;
    };
    const float *posz = S(t.pos[2].y), *rotz = S(t.rot[2].y), *hs1 = (const float*)(sb + t.pu + t.pu_hs1);
    EGO_RC(egotap_train_pose_head_bwd(h, posz, hs1, dpose, B, W(w.dposz), W(w.dhs1), G(g.pose_w), G(g.pose_b), G(g.glob_w), G(g.glob_b), 0, stream));
    float* pug[14] = {G(g.x2f0_w), G(g.x2f0_b), G(g.x2h0_w), G(g.x2h0_b), G(g.b2h0_w), G(g.b2h0_b), G(g.h2h0_w), G(g.h2h0_b),
                      G(g.x2f1_w), G(g.x2f1_b), G(g.x2h1_w), G(g.x2h1_b), G(g.h2h1_w), G(g.h2h1_b)};
    EGO_RC(egotap_train_pu_bwd(h, posz, rotz, B, sb + t.pu, W(w.dhs1), W(w.dposz), W(w.drotz), pug, 0, wb + w.pu, w.pu_bytes, stream));
    // FC encoders: fc3, fc2 in fp32 (small), fc1 on the bf16 kernels; the position encoder returns the token gradient (bf16, token order)
    auto encoder_bwd = [&](int e, const float* dy, void* dtok) -> int {
        for (int j = 2; j >= 0; --j) {
            const int n = j < 2 ? FC_OUT[j] : h->hid;
            const LiftParams::Fc& fc = e == 0 ? p.pos_fc[j] : p.rot_fc[j];
            const LiftParams::Fc& gc = e == 0 ? g.pos_fc[j] : g.rot_fc[j];
            const LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            float* dz = W(w.E[0]);
            EGO_RC(egotap_train_bn_lrelu_bwd(S(f.z), S(f.y), dy, fc.g, S(f.mean), S(f.rstd), dz, G(gc.g), G(gc.beta), BT, n, 0, scr, scrb, stream));
            EGO_RC(egotap_train_colsum(dz, 0, G(gc.b), BT, n, 0, scr, scrb, stream));
            if (j > 0) {
                const int K = FC_OUT[j - 1];
                const float* a_in = S((e == 0 ? t.pos[j - 1] : t.rot[j - 1]).y);
                EGO_RC(egotap_train_gemm_tn(h, 0, dz, 0, a_in, nullptr, G(gc.w), BT, n, K, 0, 0, scr, scrb, stream));
                EGO_RC(egotap_train_transpose(fc.w, W(w.WT), n, K, 0, stream));
                EGO_RC(egotap_train_gemm_nt(h, 0, dz, 0, nullptr, W(w.WT), nullptr, W(w.E[1]), BT, K, n, 0, nullptr, nullptr, 0, stream));
                dy = W(w.E[1]);
            } else {
                EGO_RC(egotap_bf16_from_f32(dz, Wh(w.dzb), (int64_t)BT * 2048, stream));
                EGO_RC(egotap_bf16_fc1_wgrad(h, e, Wh(w.dzb), e == 0 ? Hb(t.tokens) : Hb(t.hmb), G(gc.w), B, ZERO, scr, scrb, stream));
                if (e == 0) EGO_RC(egotap_bf16_fc1_dgrad_tokens(h, Wh(w.dzb), Hb(t.w_fc1p_t), dtok, B, stream));
            }
        }
        return EGOTAP_OK;
    };
    EGO_RC(encoder_bwd(1, W(w.drotz), nullptr));
    EGO_RC(encoder_bwd(0, W(w.dposz), Wh(w.Rb[0])));
    float *F0 = W(w.F[0]), *F1 = W(w.F[1]);
    void *R0 = Wh(w.Rb[0]), *R1 = Wh(w.Rb[1]), *A4 = Wh(w.A4b), *A3 = Wh(w.A3b);
    // final LayerNorm: dtok (R0) -> dx (F0, bf16 copy R1); column sums of dx = the last layer's output.dense.bias gradient
    EGO_RC(egotap_bf16_layernorm_bwd(S(t.X[L]), R0, p.lnf_g, S(t.mf), S(t.rf), nullptr, F0, R1, G(g.lnf_g), G(g.lnf_b), G(g.layer[L - 1].dn_b), M, 0,
                                     scr, scrb, stream));
    for (int i = L - 1; i >= 0; --i) {
        EGO_RC(bucket_done());                                                   // everything above layer i is final
        const auto& P_ = p.layer[i];
        const auto& G_ = g.layer[i];
        const auto& l = t.layer[i];
        // MLP (dx = F0 fp32, R1 bf16)
        EGO_RC(egotap_bf16_gemm_tn(R1, D, Hb(l.hid), 4 * D, G(G_.dn_w), M, D, 4 * D, 0, ZERO, scr, scrb, stream));
        // dz; its column sums (= the intermediate.dense bias gradient) leave the epilogue as per-wave partial sums in R0 (free until dy2)
        EGO_RC(egotap_bf16_gemm_nt(R1, D, Hb(l.w_dn_t), nullptr, M, 4 * D, D, 3, Hb(l.z), A4, R0, 4 * D, stream));
        EGO_RC(egotap_train_colsum((const float*)R0, 0, G(G_.up_b), 2 * ((M + 255) / 256), 4 * D, 0, scr, scrb, stream));
        EGO_RC(egotap_bf16_gemm_tn(A4, 4 * D, Hb(l.y2), D, G(G_.up_w), M, 4 * D, D, 0, ZERO, scr, scrb, stream));
        EGO_RC(egotap_bf16_gemm_nt(A4, 4 * D, Hb(l.w_up_t), nullptr, M, D, 4 * D, 0, nullptr, R0, nullptr, D, stream));             // dy2
        EGO_RC(egotap_bf16_layernorm_bwd(S(l.xm), R0, P_.ln2_g, S(l.m2), S(l.r2), F0, F1, R1, G(G_.ln2_g), G(G_.ln2_b), G(G_.o_b), M, 0, scr, scrb,
                                         stream));                                                                                   // dxm = F1, R1
        // attention
        EGO_RC(egotap_bf16_gemm_tn(R1, D, Hb(l.ctx), D, G(G_.o_w), M, D, D, 0, ZERO, scr, scrb, stream));
        EGO_RC(egotap_bf16_gemm_nt(R1, D, Hb(l.w_o_t), nullptr, M, D, D, 0, nullptr, R0, nullptr, D, stream));                       // dctx
        // dqkv, and the q | k | v bias gradients from the kernels' epilogues (no pass over dqkv)
        EGO_RC(egotap_bf16_attention_bwd_bias(Hb(l.qkv), Hb(l.ctx), R0, S(l.lse), W(w.delta), A3, G(G_.q_b), G(G_.k_b), G(G_.v_b), B, h->seq, heads, scr, scrb,
                                              stream));
        float* gw[3] = {G(G_.q_w), G(G_.k_w), G(G_.v_w)};
        for (int q = 0; q < 3; ++q)
            EGO_RC(egotap_bf16_gemm_tn((const __bf16*)A3 + (size_t)q * D, 3 * D, Hb(l.y1), D, gw[q], M, D, D, 0, ZERO, scr, scrb, stream));
        EGO_RC(egotap_bf16_gemm_nt(A3, 3 * D, Hb(l.w_qkv_t), nullptr, M, D, 3 * D, 0, nullptr, R0, nullptr, D, stream));             // dy1
        EGO_RC(egotap_bf16_layernorm_bwd(S(t.X[i]), R0, P_.ln1_g, S(l.m1), S(l.r1), F1, F0, i > 0 ? R1 : nullptr, G(G_.ln1_g), G(G_.ln1_b),
                                         i > 0 ? G(g.layer[i - 1].dn_b) : nullptr, M, 0, scr, scrb, stream));                        // dx = F0, R1
    }
    // patch embedding (fp32 operands: the input heatmaps)
    EGO_RC(egotap_train_gemm_tn(h, 1, F0, 0, hm, nullptr, G(g.patch_w), M, D, 256, 0, 0, scr, scrb, stream));
    EGO_RC(egotap_train_colsum(F0, 0, G(g.pos_emb), B, h->seq * D, 0, scr, scrb, stream));
    EGO_RC(egotap_train_patch_split(h, g.pos_emb, G(g.patch_b), G(g.mask_tok), 0, stream));
    EGO_RC(bucket_done());
    EGO_RC(bucket_done());
    return EGOTAP_OK;
}

extern "C" int egotap_lift_forward_train(egotap_handle h, const float* hm, int B, float* pose, void* saved, size_t saved_bytes, void* ws,
                                         size_t ws_bytes, void* stream) {
    EGO_CHECK(h && hm && pose && saved && ws && B > 0, "egotap_lift_forward_train: bad argument");
    EGO_CHECK(h->seq % 32 == 0, "egotap_lift_forward_train: training needs a ViT sequence that is a multiple of 32 (heatmap sides 64, 128, ...: every shipped configuration); "
              "this head has %d tokens (heatmap side %d) -- evaluation runs at any side that is a multiple of 16", h->seq, h->cfg.hm_size);
    EGO_RC(lift_resolve(h));
    if (lift_train_bf16s(h)) return lift_forward_train16(h, hm, B, pose, saved, saved_bytes, ws, ws_bytes, stream);
    LiftTrainPlan t; LiftBwdPlan w;
    EGO_RC(lift_train_plan(h, B, t, w));
    EGO_CHECK(saved_bytes >= t.total, "egotap_lift_forward_train: saved buffer too small (%zu < %zu)", saved_bytes, t.total);
    EGO_CHECK(ws_bytes >= w.total, "egotap_lift_forward_train: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const LiftParams& p = h->lp;
    char* sb = (char*)saved;
    auto S = [&](size_t off) { return (float*)(sb + off); };
    void* scr = (char*)ws + w.scr;
    const int M = B * h->seq, D = h->D, BT = B * h->T, heads = h->cfg.vit_heads, L = h->cfg.vit_layers;
    const int prec = h->precision == EGOTAP_PREC_BF16 ? EGOTAP_PREC_BF16 : EGOTAP_PREC_F32;   // attention: exact unless the whole step is bf16
    EGO_RC(egotap_train_patch_fwd(h, hm, B, p.patch_w, p.patch_b, p.mask_tok, p.pos_emb, S(t.X[0]), stream));
    for (int i = 0; i < L; ++i) {
        const auto& P_ = p.layer[i];
        const auto& l = t.layer[i];
        float* x = S(t.X[i]);
        EGO_RC(egotap_train_layernorm_fwd(x, S(l.y1), P_.ln1_g, P_.ln1_b, S(l.m1), S(l.r1), M, 1e-12f, stream));
        EGO_RC(egotap_train_qkv_fwd(h, S(l.y1), P_.q_w, P_.q_b, P_.k_w, P_.k_b, P_.v_w, P_.v_b, S(l.qkv), M, D, stream));
        EGO_RC(egotap_train_attention_fwd(S(l.qkv), S(l.ctx), S(l.lse), B, h->seq, heads, prec, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, S(l.ctx), 0, nullptr, P_.o_w, P_.o_b, S(l.xm), M, D, D, 2, x, nullptr, 0, stream));
        EGO_RC(egotap_train_layernorm_fwd(S(l.xm), S(l.y2), P_.ln2_g, P_.ln2_b, S(l.m2), S(l.r2), M, 1e-12f, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, S(l.y2), 0, nullptr, P_.up_w, P_.up_b, S(l.hid), M, 4 * D, D, 3, nullptr, S(l.z), 0, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, S(l.hid), 0, nullptr, P_.dn_w, P_.dn_b, S(t.X[i + 1]), M, D, 4 * D, 2, S(l.xm), nullptr, 0, stream));
    }
    EGO_RC(egotap_train_layernorm_fwd(S(t.X[L]), S(t.tokens), p.lnf_g, p.lnf_b, S(t.mf), S(t.rf), M, 1e-12f, stream));
    for (int e = 0; e < 2; ++e) {
        const float* a_in = e == 0 ? S(t.tokens) : hm;
        int loader = e == 0 ? 2 : 3, K = e == 0 ? h->ppd * h->ppd * D : 2 * h->cfg.hm_size * h->cfg.hm_size;
        for (int j = 0; j < 3; ++j) {
            const int n = j < 2 ? FC_OUT[j] : h->hid;
            const LiftParams::Fc& fc = e == 0 ? p.pos_fc[j] : p.rot_fc[j];
            const LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            EGO_RC(egotap_train_gemm_nt(h, loader, a_in, 0, nullptr, fc.w, fc.b, S(f.z), BT, n, K, 1, nullptr, nullptr, 0, stream));
            // train-mode BatchNorm1d: batch statistics, running statistics updated in the bound buffers (momentum 0.1, unbiased variance)
            EGO_RC(egotap_train_bn_lrelu_fwd(S(f.z), S(f.y), fc.g, fc.beta, S(f.mean), S(f.rstd), (float*)fc.mean, (float*)fc.var, BT, n, 1e-5f,
                                             0.1f, scr, w.scr_bytes, stream));
            a_in = S(f.y); loader = 0; K = n;
        }
    }
    EGO_RC(egotap_train_pu_fwd(h, S(t.pos[2].y), S(t.rot[2].y), B, sb + t.pu, t.pu_bytes, stream));
    EGO_RC(egotap_train_pose_head_fwd(h, S(t.pos[2].y), (const float*)(sb + t.pu + t.pu_hs1), B, pose, stream));
    return EGOTAP_OK;
}

extern "C" int egotap_lift_backward(egotap_handle h, const float* hm, const float* dpose, int B, const void* saved, size_t saved_bytes,
                                    void* ws, size_t ws_bytes, void* const* bucket_events, int n_events, void* stream) {
    EGO_CHECK(h && hm && dpose && saved && ws && B > 0, "egotap_lift_backward: bad argument");
    EGO_RC(lift_resolve(h));
    if (!h->grad_resolved) {
        EGO_RC(lift_resolve_into(h, h->bound_grad, h->lg, false));
        h->grad_resolved = true;
    }
    const int L = h->cfg.vit_layers;
    EGO_CHECK(n_events == 0 || (bucket_events && n_events == L + 2), "egotap_lift_backward: %d bucket events, the arena has %d buckets", n_events, L + 2);
    if (lift_train_bf16s(h)) return lift_backward16(h, hm, dpose, B, saved, saved_bytes, ws, ws_bytes, bucket_events, n_events, stream);
    LiftTrainPlan t; LiftBwdPlan w;
    EGO_RC(lift_train_plan(h, B, t, w));
    EGO_CHECK(saved_bytes >= t.total, "egotap_lift_backward: saved buffer too small (%zu < %zu)", saved_bytes, t.total);
    EGO_CHECK(ws_bytes >= w.total, "egotap_lift_backward: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const LiftParams& p = h->lp;
    const LiftParams& g = h->lg;
    auto G = [](const float* q) { return (float*)q; };
    const char* sb = (const char*)saved;
    auto S = [&](size_t off) { return (const float*)(sb + off); };
    char* wb = (char*)ws;
    auto W = [&](size_t off) { return (float*)(wb + off); };
    void* scr = wb + w.scr;
    const size_t scrb = w.scr_bytes;
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->seq, D = h->D, BT = B * h->T, heads = h->cfg.vit_heads;
    const int prec = h->precision == EGOTAP_PREC_BF16 ? EGOTAP_PREC_BF16 : EGOTAP_PREC_F32;
    int bucket = 0;
    auto bucket_done = [&]() -> int {     // every gradient of the next bucket (arena order, egotap_amd/training.py _arena_layout) is final
        if (n_events) EGO_HIP(hipEventRecord((hipEvent_t)bucket_events[bucket], s));
        ++bucket;
        return EGOTAP_OK;
    };
    const float *posz = S(t.pos[2].y), *rotz = S(t.rot[2].y), *hs1 = (const float*)(sb + t.pu + t.pu_hs1);
    // pose head + propagation units
    EGO_RC(egotap_train_pose_head_bwd(h, posz, hs1, dpose, B, W(w.dposz), W(w.dhs1), G(g.pose_w), G(g.pose_b), G(g.glob_w), G(g.glob_b), 0, stream));
    float* pug[14] = {G(g.x2f0_w), G(g.x2f0_b), G(g.x2h0_w), G(g.x2h0_b), G(g.b2h0_w), G(g.b2h0_b), G(g.h2h0_w), G(g.h2h0_b),
                      G(g.x2f1_w), G(g.x2f1_b), G(g.x2h1_w), G(g.x2h1_b), G(g.h2h1_w), G(g.h2h1_b)};
    EGO_RC(egotap_train_pu_bwd(h, posz, rotz, B, sb + t.pu, W(w.dhs1), W(w.dposz), W(w.drotz), pug, 0, wb + w.pu, w.pu_bytes, stream));
    // the two FC encoders, last block first; returns in *dA the gradient w.r.t. the gathered fc1 rows (position encoder only)
    auto encoder_bwd = [&](int e, const float* dy, float** dA) -> int {
        for (int j = 2; j >= 0; --j) {
            const int n = j < 2 ? FC_OUT[j] : h->hid;
            const int K = j > 0 ? FC_OUT[j - 1] : (e == 0 ? h->ppd * h->ppd * D : 2 * h->cfg.hm_size * h->cfg.hm_size);
            const int loader = j > 0 ? 0 : (e == 0 ? 2 : 3);
            const LiftParams::Fc& fc = e == 0 ? p.pos_fc[j] : p.rot_fc[j];
            const LiftParams::Fc& gc = e == 0 ? g.pos_fc[j] : g.rot_fc[j];
            const LiftTrainPlan::Fc& f = e == 0 ? t.pos[j] : t.rot[j];
            const float* a_in = j > 0 ? S((e == 0 ? t.pos[j - 1] : t.rot[j - 1]).y) : (e == 0 ? S(t.tokens) : hm);
            float* dz = W(w.E[0]);
            EGO_RC(egotap_train_bn_lrelu_bwd(S(f.z), S(f.y), dy, fc.g, S(f.mean), S(f.rstd), dz, G(gc.g), G(gc.beta), BT, n, 0, scr, scrb, stream));
            EGO_RC(egotap_train_gemm_tn(h, loader, dz, 0, a_in, nullptr, G(gc.w), BT, n, K, 0, 0, scr, scrb, stream));
            EGO_RC(egotap_train_colsum(dz, 0, G(gc.b), BT, n, 0, scr, scrb, stream));
            if (j == 0 && e == 1) return EGOTAP_OK;                          // the rotation encoder's input is data
            EGO_RC(egotap_train_transpose(fc.w, W(w.WT), n, K, 0, stream));  // [K, n]
            float* dnext = j == 0 ? W(w.R[1]) : W(w.E[1]);
            EGO_RC(egotap_train_gemm_nt(h, 0, dz, 0, nullptr, W(w.WT), nullptr, dnext, BT, K, n, 0, nullptr, nullptr, 0, stream));
            dy = dnext;
            if (j == 0) *dA = dnext;
        }
        return EGOTAP_OK;
    };
    float* dA = nullptr;
    EGO_RC(encoder_bwd(1, W(w.drotz), nullptr));
    EGO_RC(encoder_bwd(0, W(w.dposz), &dA));
    float *R0 = W(w.R[0]), *R1 = W(w.R[1]), *R2 = W(w.R[2]), *A4 = W(w.A4), *A3 = W(w.A3), *WT = W(w.WT);
    EGO_RC(egotap_train_tokens_scatter(h, dA, R0, B, stream));                                                   // dtok
    EGO_RC(egotap_train_layernorm_bwd(S(t.X[L]), R0, p.lnf_g, S(t.mf), S(t.rf), nullptr, R1, G(g.lnf_g), G(g.lnf_b), M, 0, scr, scrb, stream));
    float* dx = R1;
    for (int i = L - 1; i >= 0; --i) {
        const auto& P_ = p.layer[i];
        const auto& G_ = g.layer[i];
        const auto& l = t.layer[i];
        // output.dense.bias of layer i closes the bucket of layer i + 1 (or of the head, for the last layer)
        // [r3] every bias gradient of the layer comes out of the weight-gradient GEMM that stages the same dY (egotap_train_gemm_tn_bias)
        EGO_RC(egotap_train_gemm_tn_bias(h, dx, 0, S(l.hid), G(G_.dn_w), G(G_.dn_b), M, D, 4 * D, 0, scr, scrb, stream));
        EGO_RC(bucket_done());
        // MLP
        EGO_RC(egotap_train_transpose(P_.dn_w, WT, D, 4 * D, 0, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, dx, 0, nullptr, WT, nullptr, A4, M, 4 * D, D, 5, S(l.z), nullptr, 0, stream));          // dz
        EGO_RC(egotap_train_gemm_tn_bias(h, A4, 0, S(l.y2), G(G_.up_w), G(G_.up_b), M, 4 * D, D, 0, scr, scrb, stream));
        EGO_RC(egotap_train_transpose(P_.up_w, WT, 4 * D, D, 0, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, A4, 0, nullptr, WT, nullptr, R0, M, D, 4 * D, 0, nullptr, nullptr, 0, stream));          // dy2
        EGO_RC(egotap_train_layernorm_bwd(S(l.xm), R0, P_.ln2_g, S(l.m2), S(l.r2), dx, R2, G(G_.ln2_g), G(G_.ln2_b), M, 0, scr, scrb, stream));
        float* dxm = R2;
        // attention
        EGO_RC(egotap_train_gemm_tn_bias(h, dxm, 0, S(l.ctx), G(G_.o_w), G(G_.o_b), M, D, D, 0, scr, scrb, stream));
        EGO_RC(egotap_train_transpose(P_.o_w, WT, D, D, 0, stream));
        EGO_RC(egotap_train_gemm_nt(h, 0, dxm, 0, nullptr, WT, nullptr, R0, M, D, D, 0, nullptr, nullptr, 0, stream));             // dctx
        EGO_RC(egotap_train_attention_bwd(S(l.qkv), S(l.ctx), R0, S(l.lse), W(w.delta), A3, B, h->seq, heads, prec, stream));   // dqkv
        const float* pw[3] = {P_.q_w, P_.k_w, P_.v_w};
        float* gw[3] = {G(G_.q_w), G(G_.k_w), G(G_.v_w)};
        float* gb[3] = {G(G_.q_b), G(G_.k_b), G(G_.v_b)};
        for (int q = 0; q < 3; ++q) {
            EGO_RC(egotap_train_gemm_tn_bias(h, A3 + (size_t)q * D, 3 * D, S(l.y1), gw[q], gb[q], M, D, D, 0, scr, scrb, stream));
            EGO_RC(egotap_train_transpose(pw[q], WT + (size_t)q * D, D, D, 3 * D, stream));                                         // [Wq^T | Wk^T | Wv^T]
        }
        EGO_RC(egotap_train_gemm_nt(h, 0, A3, 0, nullptr, WT, nullptr, R0, M, D, 3 * D, 0, nullptr, nullptr, 0, stream));          // dy1
        EGO_RC(egotap_train_layernorm_bwd(S(t.X[i]), R0, P_.ln1_g, S(l.m1), S(l.r1), dxm, R1, G(G_.ln1_g), G(G_.ln1_b), M, 0, scr, scrb, stream));
        dx = R1;
    }
    // patch embedding: weight, position embeddings (sum over the batch: dx viewed as [B, seq * D]), bias / mask token
    EGO_RC(egotap_train_gemm_tn(h, 1, dx, 0, hm, nullptr, G(g.patch_w), M, D, 256, 0, 0, scr, scrb, stream));
    EGO_RC(egotap_train_colsum(dx, 0, G(g.pos_emb), B, h->seq * D, 0, scr, scrb, stream));
    EGO_RC(egotap_train_patch_split(h, g.pos_emb, G(g.patch_b), G(g.mask_tok), 0, stream));
    EGO_RC(bucket_done());
    EGO_RC(bucket_done());
    return EGOTAP_OK;
}
#endif
