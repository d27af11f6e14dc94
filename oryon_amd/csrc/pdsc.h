// Internal structures shared by the PointDSC translation units.
#pragma once
#include <stdlib.h>
#include <map>
#include <string>
#include <vector>
#include "common.h"

namespace oryon {

struct PdscLayer {
    const float *w_pcn, *b_pcn;   // [C,C] BN-folded, [C]
    const float *w_qkv, *b_qkv;   // [3C,C], [3C]
    const float *w_m1, *b_m1;     // [C/2,C] BN-folded
    const float *w_m2, *b_m2;     // [C/2,C/2] BN-folded
    const float *w_m3, *b_m3;     // [C,C/2]
    const char *mlp_img;          // fc_message weights as the LDS image of pdsc_mlp3_x3_kernel (C == 128 only, else nullptr)
    const char *mlp_img_p;        // the same with W1's K axis in accumulator-register order (input = the attention output still in registers)
    const char *pq_img;           // PointCN + q|k|v weights as the four 64 KB LDS chunks of pdsc_pcn_qkv_x3_kernel (C == 128 only)
};

// LDS image of one layer's fc_message weights for pdsc_mlp3_x3_kernel (C = 128, H = 64): six fp16 matrices, rows swizzled for
// conflict-free ds_read_b128, the K axis of the 2nd / 3rd matrix permuted into MFMA accumulator-register order (pdsc_encoder.hip).
constexpr int PDSC_MLP_W1H = 0, PDSC_MLP_W1L = 16384, PDSC_MLP_W2H = 32768, PDSC_MLP_W2L = 40960, PDSC_MLP_W3H = 49152,
              PDSC_MLP_W3L = 65536, PDSC_MLP_IMG_BYTES = 81920;
// pdsc_pcn_qkv_x3_kernel: four chunks (PointCN, q, k, v) of [hi: 128 rows x 256 B | lo: the same], slot ^ (row & 15); q|k|v K axis permuted
constexpr int PDSC_PQ_CHUNK_BYTES = 65536, PDSC_PQ_IMG_BYTES = 5 * PDSC_PQ_CHUNK_BYTES;   // PointCN | q | k | v | PointCN with the permuted K axis
// K / V image of one 64-key tile (C = 128): Kh | Kl as [16 channel octets][64 keys][8 halves], Vh | Vl as [8 key octets][128 channels][8 keys].
// Round 6 (K): the 16 bytes a lane reads (key, channel octet) sit next to the neighbouring KEYS' 16 bytes, not next to the key's other
// channels ([64 keys][136 halves] before): a wave's read is 2 x 512 contiguous bytes (conflict-free without the row pad), and the writers'
// 8-byte pieces (lane = key: 4 channels each, two lanes per octet) fill 512 contiguous bytes = four whole 128-byte lines per store
// instruction instead of touching 32 lines - the store path takes ~4 cycles per line it touches (DESIGN.md "PointDSC encoder: what round 6 found").
constexpr int PDSC_KV_KL = 16384, PDSC_KV_VH = 32768, PDSC_KV_VL = 49152, PDSC_KV_TILE_BYTES = 65536;
// pdsc_att_chain_x3_kernel's LDS: [0, 128 KB) two K / V tiles, later [0, 80 KB) the fc_message image + [80 KB, 146 KB) the key-half merge area /
// weight areas; behind them the layer's biases (768 floats)
constexpr int PDSC_AC_BIAS_OFF = PDSC_MLP_IMG_BYTES + 4 * 64 * (128 / 32 * 16 + 2) * 4, PDSC_AC_BIAS_BYTES = 768 * 4;
// element (half) index inside a K plane of the tile image: key 0..63, channel octet 0..15
__host__ __device__ constexpr int pdsc_k_img_elem(int key, int octet) { return (octet * 64 + key) * 8; }
// "G4" row-fragment layout of the one-launch-per-layer path's q and PointCN-output arrays (round 6): [32 channel quads][n_cap rows][4 floats]
// per pair instead of [n_cap rows][C] - a lane owns (row, quad) pieces (lane = row, accumulator registers 4 g .. 4 g + 3 = one quad), so a
// float4 access of 32 lanes covers 512 contiguous bytes.  float4 index of (row, quad):
__host__ __device__ constexpr size_t pdsc_g4_index(int n_cap, int row, int quad) { return (size_t)quad * n_cap + row; }

struct PdscModel {
    oryon_pointdsc_config_t cfg;
    float sigma;      // feature-consistency sigma (learnable scalar, PointDSC.py:97)
    float sigma_d;    // sigma_spat (PointDSC.py:98)
    const float *w0, *b0;             // layer0 [C,in_dim]
    std::vector<PdscLayer> layers;
    const float *w_c1, *b_c1, *w_c2, *b_c2, *w_c3, *b_c3;
};

struct PdscWorkspace {
    float *corr_pos;  // [B,n_cap,8]
    float *feat;      // [B,n_cap,C]
    float *feat1;     // [B,n_cap,C]
    float *qkv;       // [B,n_cap,3C]
    float *msg;       // [B,n_cap,C]
    char *kv_img;     // [B,n_cap/64,PDSC_KV_TILE_BYTES] K / V of every 64-key tile as the attention kernel's LDS image (C == 128; else unused)
    char *kv_img2;    // second image: the one-launch-per-layer kernel reads layer l's K / V while its workgroups write layer l + 1's
    float *sc;        // [B,n_cap/32,n_cap/64,8,64,4] spatial-consistency tiles in attention-register layout
    float *att_o;     // [att_splits,B,n_cap,C]   key-split attention partials (att_splits > 1 only)
    float *att_ml;    // [att_splits,B,n_cap,2]   running max, exp-sum
    int att_splits;
    float *h1, *h2;   // [B,n_cap,max(C/2,32)]
    float *feat_n;    // [B,n_cap,C]
    float *conf;      // [B,n_cap]
    float *seed_key;  // [B,n_cap] NMS keys
    int32_t *seeds;   // [B,S_cap]
    int32_t *n_seeds; // [B]
    int32_t *knn;     // [B,S_cap,k]
    float *Mmat;      // [B,S_cap,k,k]
    float *seed_w;    // [B,S_cap,k]
    float *seed_dist; // [B,S_cap,n_cap] feature distances seed -> every row
    float *v_hist;    // [B,S_cap,16,64] power-iteration iterates
    int32_t *close_hist;  // [B,S_cap,16] per-iteration closeness flags
    float *seed_T;    // [B,S_cap,16]
    float *fitness;   // [B,S_cap]
    int32_t *best;    // [B]
    float *T0;        // [B,16]
    int S_cap, k;
};

// key splits of the attention launch: aim at >= 2 workgroups per CU (512), never more splits than 64-key tiles
inline int pdsc_attention_splits(int B, int n_cap)
{
    static const int forced = dev_env_int("ORYON_PDSC_ATT_SPLITS", 0);
    const int blocks = B * (n_cap / 128);
    int ks = forced ? forced : (256 + blocks - 1) / blocks;     // one workgroup per CU is enough once tile t+1 is prefetched
    const int tiles = n_cap / 64;
    if (ks > 4) ks = 4;
    if (ks > tiles) ks = tiles;
    return ks < 1 ? 1 : ks;
}

inline int pdsc_seed_cap(const oryon_pointdsc_config_t &cfg, int n_cap)
{
    return (int)((double)n_cap * (double)cfg.ratio) + 1;
}

void pdsc_launch_normalise(const float *feat, int C, int n_cap, int B, const int32_t *n_rows, float *out, hipStream_t st);
int pdsc_run_encoder(const PdscModel &M, const PdscWorkspace &ws, const float *src, const float *tgt, const int32_t *n_rows,
                     int B, int n_cap, hipStream_t st);
int pdsc_run_seeds(const PdscModel &M, const float *src, const float *conf, const int32_t *n_rows, int B, int n_cap, int S_cap,
                   int32_t *seeds, int32_t *n_seeds, float *key_scratch, hipStream_t st);
int pdsc_run_hypotheses(const PdscModel &M, const PdscWorkspace &ws, const float *src, const float *tgt, const float *feat_n,
                        const int32_t *n_rows, const int32_t *seeds, const int32_t *n_seeds, int B, int n_cap, float *seed_T,
                        float *fitness, int32_t *best, float *T_best, uint8_t *labels, hipStream_t st);
int pdsc_run_refine(const PdscModel &M, const float *src, const float *tgt, const int32_t *n_rows, int B, int n_cap,
                    const float *T_in, const int32_t *status_in, const int32_t *n_seeds, float *T_out, int32_t *status_out,
                    hipStream_t st);

}  // namespace oryon
