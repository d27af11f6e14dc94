// PointDSC handle: parameter loading by reference state-dict name, eval-mode BatchNorm folding, device upload,
// workspace carving and the C-ABI entry points (include/oryon_hip.h) that chain the K3-K10 kernels.
// Mirrors get_pointdsc_solver / get_pointdsc_pose (utils/pointdsc/init.py:10-57) and PointDSC.forward's
// inference branch (models/pointdsc/PointDSC.py:128-197).
#include <math.h>
#include "common.h"
#include "pdsc.h"

using namespace oryon;

struct oryon_pointdsc {
    oryon_pointdsc_config_t cfg;
    std::map<std::string, std::vector<float>> host;   // raw tensors by reference name
    float *dev_blob = nullptr;
    char *dev_mlp = nullptr;            // [num_layers][PDSC_MLP_IMG_BYTES] (C == 128)
    char *dev_pq = nullptr;             // [num_layers][PDSC_PQ_IMG_BYTES] (C == 128)
    float *seed_scratch = nullptr;      // oryon_pointdsc_seeds (stage API, no workspace argument): grown on demand
    size_t seed_scratch_floats = 0;
    PdscModel model;
    bool finalized = false;
};

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T> T *take(size_t count)
    {
        off = align_up(off, 256);
        T *r = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
};

size_t carve(const oryon_pointdsc_config_t &cfg, int B, int n_cap, void *ws_ptr, PdscWorkspace *ws)
{
    const int C = cfg.num_channels, Hh = (C / 2 > 32) ? C / 2 : 32;
    const int S_cap = pdsc_seed_cap(cfg, n_cap), k = cfg.k;
    Carver c(ws_ptr);
    PdscWorkspace w;
    const size_t rows = (size_t)B * n_cap;
    w.corr_pos = c.take<float>(rows * 8);
    w.feat = c.take<float>(rows * C);
    w.feat1 = c.take<float>(rows * C);
    w.qkv = c.take<float>(rows * 3 * C);
    w.msg = c.take<float>(rows * C);
    w.kv_img = c.take<char>(C == 128 ? rows / 64 * PDSC_KV_TILE_BYTES : 0);
    w.kv_img2 = c.take<char>(C == 128 ? rows / 64 * PDSC_KV_TILE_BYTES : 0);
    w.sc = c.take<float>(rows * n_cap);
    w.att_splits = pdsc_attention_splits(B, n_cap);
    w.att_o = c.take<float>(w.att_splits > 1 ? rows * C * w.att_splits : 0);
    w.att_ml = c.take<float>(w.att_splits > 1 ? rows * 2 * w.att_splits : 0);
    w.h1 = c.take<float>(rows * Hh);
    w.h2 = c.take<float>(rows * Hh);
    w.feat_n = c.take<float>(rows * C);
    w.conf = c.take<float>(rows);
    w.seed_key = c.take<float>(rows);
    w.seeds = c.take<int32_t>((size_t)B * S_cap);
    w.n_seeds = c.take<int32_t>(B);
    w.knn = c.take<int32_t>((size_t)B * S_cap * k);
    w.Mmat = c.take<float>((size_t)B * S_cap * k * k);
    w.seed_w = c.take<float>((size_t)B * S_cap * k);
    w.seed_dist = c.take<float>((size_t)B * S_cap * n_cap);
    w.v_hist = c.take<float>((size_t)B * S_cap * 16 * 64);
    w.close_hist = c.take<int32_t>((size_t)B * S_cap * 16);
    w.seed_T = c.take<float>((size_t)B * S_cap * 16);
    w.fitness = c.take<float>((size_t)B * S_cap);
    w.best = c.take<int32_t>(B);
    w.T0 = c.take<float>((size_t)B * 16);
    w.S_cap = S_cap;
    w.k = k;
    if (ws) *ws = w;
    return align_up(c.off, 256);
}

bool expected_numel(const oryon_pointdsc_config_t &cfg, const std::string &name, int64_t *numel)
{
    const int C = cfg.num_channels, H = C / 2;
    auto ends = [&](const char *s) { const size_t n = strlen(s); return name.size() >= n && name.compare(name.size() - n, n, s) == 0; };
    if (name == "sigma" || name == "sigma_spat") { *numel = 1; return true; }
    if (name == "encoder.layer0.weight") { *numel = (int64_t)C * cfg.in_dim; return true; }
    if (name == "encoder.layer0.bias") { *numel = C; return true; }
    if (name.rfind("classification.", 0) == 0) {
        if (name == "classification.0.weight") *numel = 32 * C;
        else if (name == "classification.2.weight") *numel = 32 * 32;
        else if (name == "classification.4.weight") *numel = 32;
        else if (name == "classification.0.bias" || name == "classification.2.bias") *numel = 32;
        else if (name == "classification.4.bias") *numel = 1;
        else return false;
        return true;
    }
    if (name.rfind("encoder.blocks.", 0) != 0) return false;
    if (ends("num_batches_tracked")) { *numel = 1; return true; }
    const bool pcn = name.find("PointCN_layer_") != std::string::npos;
    const bool nl = name.find("NonLocal_layer_") != std::string::npos;
    if (pcn) {
        if (ends(".0.weight")) *numel = (int64_t)C * C;
        else *numel = C;   // .0.bias, .1.{weight,bias,running_mean,running_var}
        return true;
    }
    if (nl) {
        if (name.find("projection_") != std::string::npos) { *numel = ends(".weight") ? (int64_t)C * C : C; return true; }
        if (name.find("fc_message.0.") != std::string::npos) { *numel = ends(".weight") ? (int64_t)H * C : H; return true; }
        if (name.find("fc_message.3.") != std::string::npos) { *numel = ends(".weight") ? (int64_t)H * H : H; return true; }
        if (name.find("fc_message.6.") != std::string::npos) { *numel = ends(".weight") ? (int64_t)C * H : C; return true; }
        if (name.find("fc_message.1.") != std::string::npos || name.find("fc_message.4.") != std::string::npos) { *numel = H; return true; }
    }
    return false;
}

// y = gamma * (W x + b - mean) / sqrt(var + eps) + beta   ->   W' = s W,  b' = s (b - mean) + beta,  s = gamma / sqrt(var+eps)
void fold_bn(const std::vector<float> &W, const std::vector<float> &b, const std::vector<float> &g, const std::vector<float> &beta,
             const std::vector<float> &mean, const std::vector<float> &var, int out, int in, float *Wo, float *bo)
{
    for (int o = 0; o < out; ++o) {
        const double s = (double)g[o] / sqrt((double)var[o] + 1e-5);
        for (int i = 0; i < in; ++i) Wo[(size_t)o * in + i] = (float)(s * (double)W[(size_t)o * in + i]);
        bo[o] = (float)(s * ((double)b[o] - (double)mean[o]) + (double)beta[o]);
    }
}

}  // namespace

extern "C" int oryon_pointdsc_create(oryon_pointdsc_t **handle, const oryon_pointdsc_config_t *cfg)
{
    ORYON_CHECK_ARG(handle && cfg);
    ORYON_CHECK_ARG(cfg->in_dim == 6 && cfg->num_layers >= 1 && cfg->num_layers <= 64);
    ORYON_CHECK_ARG(cfg->num_channels == 32 || cfg->num_channels == 64 || cfg->num_channels == 128);
    ORYON_CHECK_ARG(cfg->k >= 1 && cfg->k <= 64 && cfg->num_iterations >= 1 && cfg->ratio > 0.0f && cfg->ratio <= 1.0f);
    ORYON_CHECK_ARG(cfg->sigma_d > 0.0f && cfg->inlier_threshold > 0.0f && cfg->nms_radius >= 0.0f);
    if (cfg->num_iterations > 16) {      // the power-iteration history (v_hist / close_hist) holds 16 iterates per seed: refuse, never truncate
        set_error("oryon_pointdsc_create: num_iterations = %d exceeds the 16 iterates the device power iteration keeps per seed "
                  "(models/pointdsc/PointDSC.py:338-358 would run all of them)", cfg->num_iterations);
        return ORYON_ERR_INVALID_ARG;
    }
    auto *h = new oryon_pointdsc();
    h->cfg = *cfg;
    *handle = h;
    return ORYON_OK;
}

extern "C" void oryon_pointdsc_destroy(oryon_pointdsc_t *h)
{
    if (!h) return;
    if (h->dev_blob) (void)hipFree(h->dev_blob);
    if (h->dev_mlp) (void)hipFree(h->dev_mlp);
    if (h->dev_pq) (void)hipFree(h->dev_pq);
    if (h->seed_scratch) (void)hipFree(h->seed_scratch);
    delete h;
}

extern "C" int oryon_pointdsc_load_param(oryon_pointdsc_t *h, const char *name, const float *data_host, int64_t numel)
{
    ORYON_CHECK_ARG(h && name && data_host && numel >= 0);
    const std::string nm(name);
    int64_t want = 0;
    if (!expected_numel(h->cfg, nm, &want)) {
        set_error("oryon_pointdsc_load_param: unknown parameter '%s'", name);
        return ORYON_ERR_INVALID_ARG;
    }
    if (nm.size() > 19 && nm.compare(nm.size() - 19, 19, "num_batches_tracked") == 0) return ORYON_OK;
    if (want != numel) {
        set_error("oryon_pointdsc_load_param: '%s' has %lld elements, expected %lld", name, (long long)numel, (long long)want);
        return ORYON_ERR_INVALID_ARG;
    }
    h->host[nm].assign(data_host, data_host + numel);
    h->finalized = false;
    return ORYON_OK;
}

extern "C" int oryon_pointdsc_finalize(oryon_pointdsc_t *h, void *stream)
{
    ORYON_CHECK_ARG(h);
    const int C = h->cfg.num_channels, H = C / 2, L = h->cfg.num_layers, D = h->cfg.in_dim;
    auto get = [&](const std::string &n, const std::vector<float> **out) -> bool {
        auto it = h->host.find(n);
        if (it == h->host.end()) { set_error("oryon_pointdsc_finalize: parameter '%s' was never loaded", n.c_str()); return false; }
        *out = &it->second;
        return true;
    };
    // blob layout
    const size_t per_layer = (size_t)C * C + C + (size_t)3 * C * C + 3 * C + (size_t)H * C + H + (size_t)H * H + H + (size_t)C * H + C;
    const size_t total = (size_t)C * D + C + per_layer * L + 32 * C + 32 + 32 * 32 + 32 + 32 + 1;
    std::vector<float> blob(total);
    size_t off = 0;
    auto put = [&](size_t n) { float *p = blob.data() + off; off += n; return p; };
    const std::vector<float> *w, *b, *g, *be, *mu, *va;
    if (!get("sigma", &w)) return ORYON_ERR_STATE;
    h->model.sigma = (*w)[0];
    if (!get("sigma_spat", &w)) return ORYON_ERR_STATE;
    h->model.sigma_d = (*w)[0];
    std::vector<size_t> offs;
    if (!get("encoder.layer0.weight", &w) || !get("encoder.layer0.bias", &b)) return ORYON_ERR_STATE;
    offs.push_back(off); memcpy(put((size_t)C * D), w->data(), sizeof(float) * C * D);
    offs.push_back(off); memcpy(put(C), b->data(), sizeof(float) * C);
    for (int l = 0; l < L; ++l) {
        const std::string pc = "encoder.blocks.PointCN_layer_" + std::to_string(l);
        const std::string nl = "encoder.blocks.NonLocal_layer_" + std::to_string(l);
        if (!get(pc + ".0.weight", &w) || !get(pc + ".0.bias", &b) || !get(pc + ".1.weight", &g) || !get(pc + ".1.bias", &be) ||
            !get(pc + ".1.running_mean", &mu) || !get(pc + ".1.running_var", &va)) return ORYON_ERR_STATE;
        offs.push_back(off); float *Wo = put((size_t)C * C);
        offs.push_back(off); float *bo = put(C);
        fold_bn(*w, *b, *g, *be, *mu, *va, C, C, Wo, bo);
        offs.push_back(off); float *Wq = put((size_t)3 * C * C);
        offs.push_back(off); float *bq = put(3 * C);
        const char *proj[3] = {".projection_q", ".projection_k", ".projection_v"};
        for (int p = 0; p < 3; ++p) {
            if (!get(nl + proj[p] + ".weight", &w) || !get(nl + proj[p] + ".bias", &b)) return ORYON_ERR_STATE;
            memcpy(Wq + (size_t)p * C * C, w->data(), sizeof(float) * C * C);
            memcpy(bq + (size_t)p * C, b->data(), sizeof(float) * C);
        }
        if (!get(nl + ".fc_message.0.weight", &w) || !get(nl + ".fc_message.0.bias", &b) || !get(nl + ".fc_message.1.weight", &g) ||
            !get(nl + ".fc_message.1.bias", &be) || !get(nl + ".fc_message.1.running_mean", &mu) ||
            !get(nl + ".fc_message.1.running_var", &va)) return ORYON_ERR_STATE;
        offs.push_back(off); Wo = put((size_t)H * C);
        offs.push_back(off); bo = put(H);
        fold_bn(*w, *b, *g, *be, *mu, *va, H, C, Wo, bo);
        if (!get(nl + ".fc_message.3.weight", &w) || !get(nl + ".fc_message.3.bias", &b) || !get(nl + ".fc_message.4.weight", &g) ||
            !get(nl + ".fc_message.4.bias", &be) || !get(nl + ".fc_message.4.running_mean", &mu) ||
            !get(nl + ".fc_message.4.running_var", &va)) return ORYON_ERR_STATE;
        offs.push_back(off); Wo = put((size_t)H * H);
        offs.push_back(off); bo = put(H);
        fold_bn(*w, *b, *g, *be, *mu, *va, H, H, Wo, bo);
        if (!get(nl + ".fc_message.6.weight", &w) || !get(nl + ".fc_message.6.bias", &b)) return ORYON_ERR_STATE;
        offs.push_back(off); memcpy(put((size_t)C * H), w->data(), sizeof(float) * C * H);
        offs.push_back(off); memcpy(put(C), b->data(), sizeof(float) * C);
    }
    const char *cls[3] = {"classification.0", "classification.2", "classification.4"};
    const size_t cls_w[3] = {(size_t)32 * C, 32 * 32, 32}, cls_b[3] = {32, 32, 1};
    for (int i = 0; i < 3; ++i) {
        if (!get(std::string(cls[i]) + ".weight", &w) || !get(std::string(cls[i]) + ".bias", &b)) return ORYON_ERR_STATE;
        offs.push_back(off); memcpy(put(cls_w[i]), w->data(), sizeof(float) * cls_w[i]);
        offs.push_back(off); memcpy(put(cls_b[i]), b->data(), sizeof(float) * cls_b[i]);
    }
    if (h->dev_blob) { (void)hipFree(h->dev_blob); h->dev_blob = nullptr; }
    ORYON_CHECK_HIP(hipMalloc(&h->dev_blob, total * sizeof(float)));
    ORYON_CHECK_HIP(hipMemcpyAsync(h->dev_blob, blob.data(), total * sizeof(float), hipMemcpyHostToDevice, as_stream(stream)));
    ORYON_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    // fc_message weights once more, pre-split into fp16 hi / lo and laid out as the fused kernel's LDS image
    if (h->dev_mlp) { (void)hipFree(h->dev_mlp); h->dev_mlp = nullptr; }
    if (h->dev_pq) { (void)hipFree(h->dev_pq); h->dev_pq = nullptr; }
    if (C == 128) {
        std::vector<char> img((size_t)2 * L * PDSC_MLP_IMG_BYTES, 0);       // [L] natural W1 | [L] W1 with the permuted K axis
        auto put_half = [](char *dst_hi, char *dst_lo, size_t byte, float x) {
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            memcpy(dst_hi + byte, &hi, 2);
            memcpy(dst_lo + byte, &lo, 2);
        };
        // k-slot (16-byte unit) q = 2 * step + half, element e of a PERMUTED K axis: the channel an accumulator register holds
        auto perm_src = [](int q, int e) { const int s2 = q >> 1, hi = q & 1, rb = s2 >> 1, j = s2 & 1; return rb * 32 + (e & 3) + 8 * (2 * j + (e >> 2)) + 4 * hi; };
        for (int l = 0; l < L; ++l) {
            char *im = img.data() + (size_t)l * PDSC_MLP_IMG_BYTES;
            // the BN-folded matrices sit in the fp32 blob: w_m1 [H,C], w_m2 [H,H], w_m3 [C,H]
            const size_t base = 2 + (size_t)l * 10;                  // offs index of this layer's w_pcn
            const float *w1 = blob.data() + offs[base + 4], *w2 = blob.data() + offs[base + 6], *w3 = blob.data() + offs[base + 8];
            for (int o = 0; o < H; ++o)
                for (int k = 0; k < C; ++k)                          // natural K order, 256-byte rows, slot ^ (row & 15)
                    put_half(im + PDSC_MLP_W1H, im + PDSC_MLP_W1L, (size_t)o * 256 + (size_t)(((k >> 3) ^ (o & 15)) << 4) + (k & 7) * 2, w1[(size_t)o * C + k]);
            for (int o = 0; o < H; ++o)
                for (int q = 0; q < 8; ++q)
                    for (int e = 0; e < 8; ++e)                      // permuted K order, 128-byte rows, slot ^ ((row >> 1) & 7)
                        put_half(im + PDSC_MLP_W2H, im + PDSC_MLP_W2L, (size_t)o * 128 + (size_t)((q ^ ((o >> 1) & 7)) << 4) + e * 2, w2[(size_t)o * H + perm_src(q, e)]);
            for (int o = 0; o < C; ++o)
                for (int q = 0; q < 8; ++q)
                    for (int e = 0; e < 8; ++e)
                        put_half(im + PDSC_MLP_W3H, im + PDSC_MLP_W3L, (size_t)o * 128 + (size_t)((q ^ ((o >> 1) & 7)) << 4) + e * 2, w3[(size_t)o * H + perm_src(q, e)]);
            // second image: W2 / W3 as above, W1 with K in accumulator-register order (pdsc_att_chain_x3_kernel feeds it the merged
            // attention output straight from the registers)
            char *ip = img.data() + (size_t)(L + l) * PDSC_MLP_IMG_BYTES;
            memcpy(ip, im, PDSC_MLP_IMG_BYTES);
            for (int o = 0; o < H; ++o)
                for (int q = 0; q < 16; ++q)
                    for (int e = 0; e < 8; ++e)
                        put_half(ip + PDSC_MLP_W1H, ip + PDSC_MLP_W1L, (size_t)o * 256 + (size_t)((q ^ (o & 15)) << 4) + e * 2, w1[(size_t)o * C + perm_src(q, e)]);
        }
        ORYON_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&h->dev_mlp), img.size()));
        ORYON_CHECK_HIP(hipMemcpyAsync(h->dev_mlp, img.data(), img.size(), hipMemcpyHostToDevice, as_stream(stream)));
        ORYON_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
        // PointCN (natural K: its input comes from memory) and q | k | v (K in accumulator-register order) for pdsc_pcn_qkv_x3_kernel
        std::vector<char> pq((size_t)L * PDSC_PQ_IMG_BYTES, 0);
        for (int l = 0; l < L; ++l) {
            char *im = pq.data() + (size_t)l * PDSC_PQ_IMG_BYTES;
            const size_t base = 2 + (size_t)l * 10;
            const float *wp = blob.data() + offs[base + 0], *wq = blob.data() + offs[base + 2];
            for (int o = 0; o < C; ++o)
                for (int k = 0; k < C; ++k)
                    put_half(im, im + PDSC_PQ_CHUNK_BYTES / 2, (size_t)o * 256 + (size_t)(((k >> 3) ^ (o & 15)) << 4) + (k & 7) * 2, wp[(size_t)o * C + k]);
            for (int part = 0; part < 3; ++part) {
                char *ch = im + (size_t)(1 + part) * PDSC_PQ_CHUNK_BYTES;
                for (int o = 0; o < C; ++o)
                    for (int q = 0; q < 16; ++q)
                        for (int e = 0; e < 8; ++e)
                            put_half(ch, ch + PDSC_PQ_CHUNK_BYTES / 2, (size_t)o * 256 + (size_t)((q ^ (o & 15)) << 4) + e * 2,
                                     wq[((size_t)part * C + o) * C + perm_src(q, e)]);
            }
            // chunk 4: PointCN once more with K in accumulator-register order, for pdsc_mlp3_pcn_qkv_x3_kernel (its input is the previous
            // layer's output still in registers)
            char *c4 = im + (size_t)4 * PDSC_PQ_CHUNK_BYTES;
            for (int o = 0; o < C; ++o)
                for (int q = 0; q < 16; ++q)
                    for (int e = 0; e < 8; ++e)
                        put_half(c4, c4 + PDSC_PQ_CHUNK_BYTES / 2, (size_t)o * 256 + (size_t)((q ^ (o & 15)) << 4) + e * 2, wp[(size_t)o * C + perm_src(q, e)]);
        }
        if (h->dev_pq) { (void)hipFree(h->dev_pq); h->dev_pq = nullptr; }
        ORYON_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&h->dev_pq), pq.size()));
        ORYON_CHECK_HIP(hipMemcpyAsync(h->dev_pq, pq.data(), pq.size(), hipMemcpyHostToDevice, as_stream(stream)));
        ORYON_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    }
    const float *d = h->dev_blob;
    size_t i = 0;
    PdscModel &M = h->model;
    M.cfg = h->cfg;
    M.w0 = d + offs[i++]; M.b0 = d + offs[i++];
    M.layers.resize(L);
    for (int l = 0; l < L; ++l) {
        PdscLayer &Ly = M.layers[l];
        Ly.w_pcn = d + offs[i++]; Ly.b_pcn = d + offs[i++];
        Ly.w_qkv = d + offs[i++]; Ly.b_qkv = d + offs[i++];
        Ly.w_m1 = d + offs[i++]; Ly.b_m1 = d + offs[i++];
        Ly.w_m2 = d + offs[i++]; Ly.b_m2 = d + offs[i++];
        Ly.w_m3 = d + offs[i++]; Ly.b_m3 = d + offs[i++];
        Ly.mlp_img = h->dev_mlp ? h->dev_mlp + (size_t)l * PDSC_MLP_IMG_BYTES : nullptr;
        Ly.mlp_img_p = h->dev_mlp ? h->dev_mlp + (size_t)(L + l) * PDSC_MLP_IMG_BYTES : nullptr;
        Ly.pq_img = h->dev_pq ? h->dev_pq + (size_t)l * PDSC_PQ_IMG_BYTES : nullptr;
    }
    M.w_c1 = d + offs[i++]; M.b_c1 = d + offs[i++];
    M.w_c2 = d + offs[i++]; M.b_c2 = d + offs[i++];
    M.w_c3 = d + offs[i++]; M.b_c3 = d + offs[i++];
    h->finalized = true;
    return ORYON_OK;
}

extern "C" size_t oryon_pointdsc_workspace_bytes(const oryon_pointdsc_t *h, int B, int n_cap)
{
    if (!h || B <= 0 || n_cap <= 0) return 0;
    return carve(h->cfg, B, n_cap, nullptr, nullptr);
}

// n_cap <= 4096: pdsc_knn_matrix_kernel sorts one seed's n_cap distances in dynamic LDS (2 * pow2(n_cap) floats + ~35 KB)
#define PDSC_COMMON_CHECKS()                                                                          \
    ORYON_CHECK_ARG(h && B >= 0 && n_cap > 0 && n_cap % 128 == 0);                                      \
    if (n_cap > 4096) { set_error("%s: n_cap = %d exceeds 4096 rows per pair (LDS budget of the kNN kernel)", __func__, n_cap); return ORYON_ERR_INVALID_ARG; } \
    if (!h->finalized) { set_error("%s: handle not finalized", __func__); return ORYON_ERR_STATE; }   \
    if (B == 0) return ORYON_OK;

static int get_ws(oryon_pointdsc_t *h, int B, int n_cap, void *workspace, size_t workspace_bytes, PdscWorkspace *ws)
{
    const size_t need = carve(h->cfg, B, n_cap, nullptr, nullptr);
    if (!workspace || workspace_bytes < need) {
        set_error("pointdsc workspace too small (%zu < %zu)", workspace_bytes, need);
        return ORYON_ERR_WORKSPACE;
    }
    carve(h->cfg, B, n_cap, workspace, ws);
    return ORYON_OK;
}

extern "C" int oryon_pointdsc_register(oryon_pointdsc_t *h, const float *src, const float *tgt, const int32_t *n, int B,
                                       int n_cap, const int32_t *status_in, void *workspace, size_t workspace_bytes, float *T,
                                       uint8_t *labels, int32_t *status_out, void *stream)
{
    PDSC_COMMON_CHECKS();
    ORYON_CHECK_ARG(src && tgt && n && T);
    PdscWorkspace ws;
    int rc = get_ws(h, B, n_cap, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if ((rc = pdsc_run_encoder(h->model, ws, src, tgt, n, B, n_cap, st))) { set_error("pointdsc encoder launch failed"); return rc; }
    if ((rc = pdsc_run_seeds(h->model, src, ws.conf, n, B, n_cap, ws.S_cap, ws.seeds, ws.n_seeds, ws.seed_key, st))) { set_error("pointdsc seeds launch failed"); return rc; }
    if ((rc = pdsc_run_hypotheses(h->model, ws, src, tgt, ws.feat_n, n, ws.seeds, ws.n_seeds, B, n_cap, ws.seed_T, ws.fitness,
                                  ws.best, ws.T0, labels, st))) { set_error("pointdsc hypotheses launch failed"); return rc; }
    if ((rc = pdsc_run_refine(h->model, src, tgt, n, B, n_cap, ws.T0, status_in, ws.n_seeds, T, status_out, st))) { set_error("pointdsc refine launch failed"); return rc; }
    return ORYON_OK;
}

extern "C" int oryon_pointdsc_encode(oryon_pointdsc_t *h, const float *src, const float *tgt, const int32_t *n, int B, int n_cap,
                                     void *workspace, size_t workspace_bytes, float *feat, float *confidence, void *stream)
{
    PDSC_COMMON_CHECKS();
    ORYON_CHECK_ARG(src && tgt && n && feat && confidence);
    PdscWorkspace ws;
    int rc = get_ws(h, B, n_cap, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if ((rc = pdsc_run_encoder(h->model, ws, src, tgt, n, B, n_cap, st))) { set_error("pointdsc encoder launch failed"); return rc; }
    const size_t rows = (size_t)B * n_cap;
    ORYON_CHECK_HIP(hipMemcpyAsync(feat, ws.feat, rows * h->cfg.num_channels * sizeof(float), hipMemcpyDeviceToDevice, st));
    ORYON_CHECK_HIP(hipMemcpyAsync(confidence, ws.conf, rows * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ORYON_OK;
}

extern "C" int oryon_pointdsc_seeds(oryon_pointdsc_t *h, const float *src, const float *confidence, const int32_t *n, int B,
                                    int n_cap, int S_cap, int32_t *seeds, int32_t *n_seeds, void *stream)
{
    PDSC_COMMON_CHECKS();
    ORYON_CHECK_ARG(src && confidence && n && seeds && n_seeds && S_cap >= pdsc_seed_cap(h->cfg, n_cap));
    const size_t need = (size_t)B * n_cap;
    if (need > h->seed_scratch_floats) {
        if (h->seed_scratch) (void)hipFree(h->seed_scratch);
        h->seed_scratch = nullptr;
        h->seed_scratch_floats = 0;
        ORYON_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&h->seed_scratch), need * sizeof(float)));
        h->seed_scratch_floats = need;
    }
    int rc = pdsc_run_seeds(h->model, src, confidence, n, B, n_cap, S_cap, seeds, n_seeds, h->seed_scratch, as_stream(stream));
    if (rc) set_error("pointdsc seeds launch failed");
    return rc;
}

extern "C" int oryon_pointdsc_hypotheses(oryon_pointdsc_t *h, const float *src, const float *tgt, const float *feat, const int32_t *n,
                                         const int32_t *seeds, const int32_t *n_seeds, int B, int n_cap, int S_cap, void *workspace,
                                         size_t workspace_bytes, float *seed_T, float *fitness, int32_t *best, void *stream)
{
    PDSC_COMMON_CHECKS();
    ORYON_CHECK_ARG(src && tgt && feat && n && seeds && n_seeds && seed_T && fitness && best);
    PdscWorkspace ws;
    int rc = get_ws(h, B, n_cap, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    ORYON_CHECK_ARG(S_cap == ws.S_cap);
    hipStream_t st = as_stream(stream);
    // feat is the un-normalised encoder output [B,n_cap,C]; normalise it exactly as the full path does
    pdsc_launch_normalise(feat, h->cfg.num_channels, n_cap, B, n, ws.feat_n, st);
    rc = pdsc_run_hypotheses(h->model, ws, src, tgt, ws.feat_n, n, seeds, n_seeds, B, n_cap, seed_T, fitness, best, ws.T0, nullptr, st);
    if (rc) set_error("pointdsc hypotheses launch failed");
    return rc;
}

extern "C" int oryon_pointdsc_refine(oryon_pointdsc_t *h, const float *src, const float *tgt, const int32_t *n, int B, int n_cap,
                                     const float *T_in, float *T_out, uint8_t *labels, void *stream)
{
    PDSC_COMMON_CHECKS();
    ORYON_CHECK_ARG(src && tgt && n && T_in && T_out);
    (void)labels;
    int rc = pdsc_run_refine(h->model, src, tgt, n, B, n_cap, T_in, nullptr, nullptr, T_out, nullptr, as_stream(stream));
    if (rc) set_error("pointdsc refine launch failed");
    return rc;
}
