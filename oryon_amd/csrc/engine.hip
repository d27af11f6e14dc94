// oryon_engine: the whole batched match -> lift -> registration step enqueued from C++ (round 3).
//
// Replaces the body of the per-sample loop of FPM_Pipeline.test_step (pipeline.py:313-355) for B pairs, like
// oryon_amd/engine.py did with ~40 torch allocations, ~15 ctypes calls and per-step torch stream / event objects.  Here one call
// enqueues everything:
//
//     gather stream G : K0   roi_compact x2 -> roi_subsample -> gather_q8 (queries) -> gather_q8 (anchors)
//     match  stream M : K1s8 oryon_match_corrs_i8 (int8 screen, lazy tail, sampling)  -> K2 oryon_lift_pairs
//     reg    stream R : K3-K10 oryon_pointdsc_register                                  (one stream per slot)
//
// on streams and events the engine owns, over a persistent arena carved once (no allocation, no torch object, no Python per
// launch).  The K0 outputs alternate between two buffer sets and everything a step hands on or hands back (ROI lists, matcher
// outputs, lifted points, the registration workspace, poses) between n_slots (4) result slots, so K0 of step k+1 and the
// registrations of steps k-1, k-2 (two registration streams) run beside the matching of step k and no matcher ever waits for a
// registration to release its buffers - the same software pipeline the Python engine built out of torch streams and fresh tensors.  Results are bit for bit those of the Python engine (same
// entry points, same arguments): tests/test_gpu_native_engine.py.
#include <stdlib.h>
#include <chrono>
#include <new>
#include <vector>
#include "common.h"

using namespace oryon;

namespace {
constexpr int MAX_SLOTS = 8;
constexpr int MAX_G_SLOTS = 4;         // K0 output sets (the rows the screening kernel reads): cfg.gather_sets of them
constexpr int MAX_REG_STREAMS = 4;
constexpr int TIMING_RING = 64;        // timing-event sets: the sections of the last 64 steps can be read back
constexpr int ROW_PAD = 256;
constexpr int POOL_STREAMS = 8;        // HIP streams per device in the process-wide pool (see acquire_pool)
constexpr int DEFAULT_ROLES = 2345;    // cfg.stream_roles == 0 (round 6; round 5: 2301 - see oryon_engine_create)

// cfg.stream_roles: four decimal digits, each a pool index 0..7 (match / gather / registration 0 / registration 1); 0 = DEFAULT_ROLES
inline bool roles_valid(int roles)
{
    if (roles < 0 || roles > 7777) return false;
    for (int d = 0, r = roles; d < 4; ++d, r /= 10)
        if (r % 10 >= POOL_STREAMS) return false;
    return true;
}

inline size_t up(size_t x, size_t m) { return (x + m - 1) / m * m; }

struct GatherBuf {
    int8_t *a8, *q8;
    float *a_sc, *q_sc, *a_eps, *q_eps, *a_norm, *q_norm, *a_hat;
    void *q_hilo;                // cfg.x3_prefetch: [2][B, cap_q, 256] halves (hi rows | lo rows of the query rows) + q_lo_max [B]
    float *q_lo_max;
    // cfg.sample_first: the first-stage anchor subset's operands (cap_a1 rows); a8 / a_hat then serve the gated second stage
    int8_t *a8_1;
    float *a_sc_1, *a_eps_1, *a_norm_1, *a_hat_1;
};

struct SlotBuf {
    int32_t *roi_a, *roi_q, *n_a, *n_q;
    float *min_dist;
    int32_t *argmin;
    uint8_t *valid;
    int32_t *corrs, *n_valid, *n_sel, *status, *n_und;
    int32_t *roi_a1, *n_a1, *n_a2, *corrs2, *n_valid2, *n_sel2, *status2;      // cfg.sample_first only
    float *pcd_a, *pcd_q;
    int32_t *n_lift;
    float *pose;
    int32_t *status_out;
    int32_t *n_valid_out, *n_lift_out;      // copies of n_valid / n_lift made on the registration stream: part of the slot's protected block
    void *pdsc_ws;
    size_t base, bytes;                        // offset of the slot in the arena
};

struct Named {
    const char *name;
    size_t off, bytes;
};

struct Layout {
    GatherBuf gbuf[MAX_G_SLOTS];
    SlotBuf slot[MAX_SLOTS];
    std::vector<Named> names[MAX_SLOTS];
    void *match_ws;
    size_t match_ws_bytes, pdsc_ws_bytes, bytes;
    int cap_a, cap_q, c_pad, n_cap, cap_a1;
};
}  // namespace

struct oryon_engine {
    oryon_engine_config_t cfg;
    oryon_pointdsc_t *solver;
    char *arena;
    Layout L;
    int device;
    hipStream_t sg, sm, sr[MAX_REG_STREAMS];
    int n_reg_streams;
    int roles;                                // pool positions of match / gather / registration 0 / registration 1 (four digits)
    // ordering events (timing disabled), one set per result slot
    hipEvent_t ev_inputs[MAX_SLOTS], ev_gathered[MAX_SLOTS], ev_matched[MAX_SLOTS], ev_done[MAX_SLOTS];
    // timing events: gather section, match section, screening kernel, registration section
    hipEvent_t tev[TIMING_RING][9];           // per step: gather begin / end, match begin / end, screen begin / end, registration begin / end, [8] = first gather launch (behind the ROI kernels)
    bool timed[TIMING_RING];
    bool used[MAX_SLOTS];
    bool timing;
    // feedback for cfg.x3_prefetch: per slot the step's n_und / n_a arrays in pinned host memory + an event behind the copies
    int32_t *fb_host;                         // [MAX_SLOTS][2][B]
    hipEvent_t ev_fb[MAX_SLOTS];
    int64_t fb_step[MAX_SLOTS];               // submit number whose counts the slot's pinned block will hold (-1: none)
    int64_t fb_seen;                          // newest submit whose counts have been read
    bool hard_mode;
    int64_t n_x3;
    int64_t n_submit;
    double host_ns_total, host_ns_last;
};

namespace {
// carve the arena (base == nullptr: sizes only)
int carve_engine(const oryon_engine_config_t &c, const oryon_pointdsc_t *solver, char *base, Layout &L)
{
    const int HW = c.FH * c.FW;
    // narrow descriptors (the reference's own C = 32 @ 192^2, configs/config.yaml:34-35) run the same screened route zero-padded to the
    // 256-channel operand rows: zero columns change neither the canonical fmaf chain nor any bound, K0v3 reads C planes and writes
    // 256-byte rows, and 8x padding on the MX-fp6 pipe (64 channels per MFMA) is still several times faster than the exact fp32-MFMA
    // scan the per-call schedule runs at these widths
    L.c_pad = c.C <= 256 ? 256 : 512;
    const int keep = c.src_sampling > 0 ? (c.src_sampling < HW ? c.src_sampling : HW) : HW;
    L.cap_a = (int)up((size_t)keep, ROW_PAD);
    L.cap_q = (int)up((size_t)HW, ROW_PAD);
    L.n_cap = (int)up((size_t)c.n_corrs, 128);
    L.cap_a1 = c.sample_first > 0 ? (int)up((size_t)(c.sample_first < HW ? c.sample_first : HW), ROW_PAD) : 0;
    const size_t B = (size_t)c.B;
    L.match_ws_bytes = oryon_match_corrs_i8_workspace_bytes(c.B, L.c_pad, L.cap_a, L.cap_q, L.n_cap);
    if (L.cap_a1) {
        // the first stage of the sample-first schedule calls the matcher with cap_a1 anchors: fewer rows, but its query split (and with it
        // the per-split partial arrays) can be LARGER than the full call's - the one workspace serves both calls
        const size_t w1 = oryon_match_corrs_i8_workspace_bytes(c.B, L.c_pad, L.cap_a1, L.cap_q, L.n_cap);
        if (!w1) return ORYON_ERR_INVALID_ARG;
        if (w1 > L.match_ws_bytes) L.match_ws_bytes = w1;
    }
    L.pdsc_ws_bytes = oryon_pointdsc_workspace_bytes(solver, c.B, L.n_cap);
    if (!L.match_ws_bytes || !L.pdsc_ws_bytes) return ORYON_ERR_INVALID_ARG;
    size_t off = 0;
    for (int g = 0; g < c.gather_sets; ++g) {
        GatherBuf &b = L.gbuf[g];
        auto take = [&](size_t n) {
            const size_t o = off;
            off = up(off + n, 256);
            return base ? base + o : nullptr;
        };
#define TAKE(field, type, count) b.field = reinterpret_cast<type *>(take((size_t)(count) * sizeof(type)))
        TAKE(a8, int8_t, B * L.cap_a * L.c_pad);
        TAKE(q8, int8_t, B * L.cap_q * L.c_pad);
        TAKE(a_sc, float, B * (L.cap_a / 16));
        TAKE(q_sc, float, B * (L.cap_q / 16));
        TAKE(a_eps, float, B);
        TAKE(q_eps, float, B);
        TAKE(a_norm, float, B * L.cap_a);
        TAKE(q_norm, float, B * L.cap_q);
        TAKE(a_hat, float, B * L.cap_a * L.c_pad);
        b.q_hilo = nullptr;
        b.q_lo_max = nullptr;
        if (c.x3_prefetch && L.c_pad == 256) {
            b.q_hilo = take((size_t)2 * B * L.cap_q * L.c_pad * 2);
            TAKE(q_lo_max, float, B);
        }
        b.a8_1 = nullptr;
        b.a_sc_1 = b.a_eps_1 = b.a_norm_1 = b.a_hat_1 = nullptr;
        if (L.cap_a1) {
            TAKE(a8_1, int8_t, B * L.cap_a1 * L.c_pad);
            TAKE(a_sc_1, float, B * (L.cap_a1 / 16));
            TAKE(a_eps_1, float, B);
            TAKE(a_norm_1, float, B * L.cap_a1);
            TAKE(a_hat_1, float, B * L.cap_a1 * L.c_pad);
        }
#undef TAKE
    }
    for (int s = 0; s < c.n_slots; ++s) {
        SlotBuf &b = L.slot[s];
        L.names[s].clear();
        b.base = off;
        auto take = [&](const char *name, size_t n) {
            const size_t o = off;
            off = up(off + n, 256);
            L.names[s].push_back({name, o, n});
            return base ? base + o : nullptr;
        };
#define TAKE(field, type, count) b.field = reinterpret_cast<type *>(take(#field, (size_t)(count) * sizeof(type)))
        TAKE(roi_a, int32_t, B * HW);
        TAKE(roi_q, int32_t, B * HW);
        TAKE(n_a, int32_t, B);
        TAKE(n_q, int32_t, B);
        TAKE(min_dist, float, B * L.cap_a);
        TAKE(argmin, int32_t, B * L.cap_a);
        TAKE(valid, uint8_t, B * L.cap_a);
        TAKE(corrs, int32_t, B * L.n_cap * 4);
        TAKE(n_valid, int32_t, B);
        TAKE(n_sel, int32_t, B);
        TAKE(status, int32_t, B);
        TAKE(n_und, int32_t, B);
        b.roi_a1 = b.n_a1 = b.n_a2 = b.corrs2 = b.n_valid2 = b.n_sel2 = b.status2 = nullptr;
        if (L.cap_a1) {
            TAKE(roi_a1, int32_t, B * HW);
            TAKE(n_a1, int32_t, B);
            TAKE(n_a2, int32_t, B);
            TAKE(corrs2, int32_t, B * L.n_cap * 4);
            TAKE(n_valid2, int32_t, B);
            TAKE(n_sel2, int32_t, B);
            TAKE(status2, int32_t, B);
        }
        TAKE(pcd_a, float, B * L.n_cap * 3);
        TAKE(pcd_q, float, B * L.n_cap * 3);
        TAKE(n_lift, int32_t, B);
        TAKE(pose, float, B * 16);
        TAKE(status_out, int32_t, B);
        TAKE(n_valid_out, int32_t, B);
        TAKE(n_lift_out, int32_t, B);
#undef TAKE
        b.pdsc_ws = take("pdsc_ws", L.pdsc_ws_bytes);
        b.bytes = off - b.base;
    }
    L.match_ws = base ? base + off : nullptr;
    off = up(off + L.match_ws_bytes, 256);
    L.bytes = off;
    return ORYON_OK;
}

int check_cfg(const oryon_engine_config_t *c)
{
    ORYON_CHECK_ARG(c && c->B > 0 && c->C > 0 && c->C <= 512 && c->FH > 0 && c->FW > 0);
    ORYON_CHECK_ARG(c->HA > 0 && c->WA > 0 && c->HQ > 0 && c->WQ > 0 && c->n_corrs > 0 && c->src_sampling >= 0);
    ORYON_CHECK_ARG(c->dist_th > 0.0f && c->dist_th <= 0.5f && c->n_slots >= 1 && c->n_slots <= MAX_SLOTS);
    ORYON_CHECK_ARG(c->gather_sets >= 1 && c->gather_sets <= MAX_G_SLOTS && c->gather_sets <= c->n_slots);
    ORYON_CHECK_ARG(c->reg_streams >= 1 && c->reg_streams <= MAX_REG_STREAMS && c->reg_lag >= 0 && c->reg_lag < c->n_slots);
    ORYON_CHECK_ARG((c->layout == ORYON_LAYOUT_NCHW || c->layout == ORYON_LAYOUT_NHWC) && (c->screen == 0 || c->screen == 1));
    ORYON_CHECK_ARG(c->overlap >= 0 && c->overlap <= 2 && (c->overlap == 0 || c->n_slots >= 2));      // results of step k live until submit k + n_slots
    ORYON_CHECK_ARG((size_t)c->C * (size_t)c->FH * (size_t)c->FW * 4u < (1ull << 32));
    ORYON_CHECK_ARG(c->sample_first >= 0 && (c->x3_prefetch == 0 || c->x3_prefetch == 1));
    ORYON_CHECK_ARG(roles_valid(c->stream_roles));
    return ORYON_OK;
}
}  // namespace

// sizeof(oryon_engine_config_t) as this library was built: bindings in other languages check their mirror of the struct against it
namespace oryon {
__global__ void engine_counts_kernel(const int32_t *__restrict__ n_valid, const int32_t *__restrict__ n_lift, int B,
                                     int32_t *__restrict__ n_valid_out, int32_t *__restrict__ n_lift_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        n_valid_out[i] = n_valid[i];
        n_lift_out[i] = n_lift[i];
    }
}
}  // namespace oryon

extern "C" size_t oryon_engine_config_bytes(void) { return sizeof(oryon_engine_config_t); }

extern "C" size_t oryon_engine_arena_bytes(const oryon_engine_config_t *cfg, const oryon_pointdsc_t *solver)
{
    if (check_cfg(cfg) || !solver) return 0;
    Layout L;
    if (carve_engine(*cfg, solver, nullptr, L)) return 0;
    return L.bytes;
}

namespace oryon {
// The engine's HIP streams come from a per-device pool that lives as long as the process (round 5).  The runtime places a stream on one
// of its GPU_MAX_HW_QUEUES hardware queues when the stream is created, and which queues a NEW set lands on differs from build to
// build: the same hard cfg2 step ran 5.45 ms on the first engine of a process and 6.4 ms on the third (streams destroyed and re-created
// in between: two of the new ones shared a queue).  Engines of one process therefore share one set of streams - they do not run
// concurrently (a second engine's steps simply queue behind the first's), and none of them destroys a stream.
// Round 6: the pool's mutex is held while the streams are created (two threads creating engines used to race on the empty slots), the
// pool is keyed by the device the ARENA lives on, and oryon_engine_warm_streams() lets a host create the pool before anything else of
// the process creates HIP streams (RCCL's communicator does): the roles below are positions in the process's creation order.
__global__ void pool_touch_kernel() {}
struct StreamPool { hipStream_t all[POOL_STREAMS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; };
static std::mutex g_pool_mu;
static StreamPool g_pools[64];
// copies the device's eight pooled streams into out[], creating the missing ones under the lock
static hipError_t acquire_pool(int dev, hipStream_t out[POOL_STREAMS])
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    StreamPool &pool = g_pools[(dev >= 0 && dev < 64) ? dev : 0];
    for (int i = 0; i < POOL_STREAMS; ++i) {
        if (!pool.all[i]) {
            hipError_t err = hipStreamCreateWithFlags(&pool.all[i], hipStreamNonBlocking);
            if (err != hipSuccess) { pool.all[i] = nullptr; return err; }
            // An empty kernel binds the stream's hardware queue here, in pool order (the runtime may create it lazily, at the first
            // command): pool position i then means "queue i mod 4 counted from the pool's first", whatever is used first later.
            hipLaunchKernelGGL(pool_touch_kernel, dim3(1), dim3(64), 0, pool.all[i]);
            err = hipStreamSynchronize(pool.all[i]);
            if (err != hipSuccess) return err;
        }
        out[i] = pool.all[i];
    }
    return hipSuccess;
}
}  // namespace oryon

extern "C" int oryon_engine_warm_streams(void)
{
    int dev = 0;
    ORYON_CHECK_HIP(hipGetDevice(&dev));
    hipStream_t tmp[POOL_STREAMS];
    ORYON_CHECK_HIP(acquire_pool(dev, tmp));
    return ORYON_OK;
}

namespace {
// (re)assign the engine's streams from the pool by role digits; the caller has drained the engine's streams when they were in use
int assign_roles(oryon_engine *e, int roles)
{
    hipStream_t pool[POOL_STREAMS];
    ORYON_CHECK_HIP(acquire_pool(e->device, pool));
    const int r_m = roles / 1000 % 10, r_g = roles / 100 % 10, r_r0 = roles / 10 % 10, r_r1 = roles % 10;
    e->sm = e->cfg.overlap >= 1 ? pool[r_m] : nullptr;
    e->sg = e->cfg.overlap >= 2 ? pool[r_g] : nullptr;
    for (int s = 0; s < MAX_REG_STREAMS; ++s) e->sr[s] = nullptr;
    for (int s = 0; s < e->n_reg_streams; ++s)
        if (e->cfg.overlap >= 1) e->sr[s] = pool[(s == 0 ? r_r0 : s == 1 ? r_r1 : 2 + s) & 7];      // a third / fourth registration stream: pool streams 4, 5
    e->roles = roles;
    return ORYON_OK;
}
}  // namespace

extern "C" int oryon_engine_create(oryon_engine_t **handle, const oryon_engine_config_t *cfg, oryon_pointdsc_t *solver, void *arena,
                                   size_t arena_bytes)
{
    ORYON_CHECK_ARG(handle && solver && arena);
    int rc = check_cfg(cfg);
    if (rc) return rc;
    oryon_engine *e = new (std::nothrow) oryon_engine();
    ORYON_CHECK_ARG(e != nullptr);
    e->cfg = *cfg;
    e->solver = solver;
    e->arena = static_cast<char *>(arena);
    if ((rc = carve_engine(*cfg, solver, e->arena, e->L))) { delete e; set_error("oryon_engine_create: cannot size the workspaces (solver finalized?)"); return rc; }
    if (arena_bytes < e->L.bytes) {
        set_error("oryon_engine_create: arena too small (%zu < %zu)", arena_bytes, e->L.bytes);
        delete e;
        return ORYON_ERR_WORKSPACE;
    }
    // the arena decides the device: the streams come from THAT device's pool, and the calling thread must be on it (every launch of a
    // submit goes to the current device)
    (void)hipGetDevice(&e->device);
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, arena) == hipSuccess && attr.device != e->device) {
            set_error("oryon_engine_create: the arena lives on device %d, the calling thread is on device %d", attr.device, e->device);
            delete e;
            return ORYON_ERR_INVALID_ARG;
        }
        (void)hipGetLastError();
    }
    e->sg = e->sm = nullptr;
    e->timing = false;
    e->n_submit = 0;
    e->fb_host = nullptr;
    e->fb_seen = -1;
    e->hard_mode = false;
    e->n_x3 = 0;
    for (int s = 0; s < MAX_SLOTS; ++s) { e->ev_fb[s] = nullptr; e->fb_step[s] = -1; }
    e->host_ns_total = e->host_ns_last = 0.0;
    hipError_t err = hipSuccess;
    auto ok = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    // Which of the pool's 8 streams serve as match / gather / registration 0 / registration 1.  The runtime multiplexes a process's streams
    // onto a few hardware queues (the arbitration between them is not documented), and the placement matters far more
    // than one would think (cfg2 step, round 5, same box, two runs each: 0123 3.46 ms; 2301 3.33-3.34; 2345 3.34-3.38; 5670 3.7-3.8; 1357
    // 3.9; 0246 4.07; 3210 / 3456 4.1-4.2 ms).  Round 6 (tools/pg_probe4.py; 16 placements, 16 pairs per step): what ELSE the process
    // created before its first tensor shifts the picture - with a one-rank RCCL communicator created first (every rank of an N > 1 run
    // has one) registration 0 on pool stream 0 loses the pipeline (2301: 1.47 -> 2.42 ms), 2345 / 2341 / 6345 / 6341 stay at 1.44-1.47 in
    // both cases; in a plain process the gather on pool stream 7 costs 13 %.  Hence the default 2345.  cfg.stream_roles names the placement
    // (0 = that default); oryon_engine_set_stream_roles changes it on a live engine, which is how a host measures the candidates on ITS
    // process (oryon_amd.engine.MatchPoseEngine.tune_stream_roles).  (dev build: ORYON_ENGINE_ROLES overrides the default)
    static const int default_roles = dev_env_int("ORYON_ENGINE_ROLES", DEFAULT_ROLES);
    e->n_reg_streams = cfg->reg_streams;
    for (int s = 0; s < MAX_REG_STREAMS; ++s) e->sr[s] = nullptr;
    if (assign_roles(e, cfg->stream_roles > 0 ? cfg->stream_roles : (roles_valid(default_roles) ? default_roles : DEFAULT_ROLES)) != ORYON_OK)
        err = hipErrorUnknown;
    for (int s = 0; s < MAX_SLOTS; ++s) {
        e->used[s] = false;
        e->ev_inputs[s] = e->ev_gathered[s] = e->ev_matched[s] = e->ev_done[s] = nullptr;
    }
    for (int r = 0; r < TIMING_RING; ++r) {
        e->timed[r] = false;
        for (int i = 0; i < 9; ++i) e->tev[r][i] = nullptr;
    }
    for (int s = 0; s < cfg->n_slots; ++s) {
        for (hipEvent_t *ev : {&e->ev_inputs[s], &e->ev_gathered[s], &e->ev_matched[s], &e->ev_done[s]})
            ok(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
    for (int r = 0; r < TIMING_RING; ++r)
        for (int i = 0; i < 9; ++i) ok(hipEventCreate(&e->tev[r][i]));
    if (cfg->x3_prefetch && e->L.c_pad == 256) {
        ok(hipHostMalloc(reinterpret_cast<void **>(&e->fb_host), (size_t)MAX_SLOTS * 2 * cfg->B * sizeof(int32_t), hipHostMallocDefault));
        for (int s = 0; s < cfg->n_slots; ++s) ok(hipEventCreateWithFlags(&e->ev_fb[s], hipEventDisableTiming));
    }
    // rows of `corrs` beyond n_corrs are never written by the sampler: zero them once, so that a slot's first use reads like the freshly
    // zeroed tensor the per-call schedule hands out.  (When a slot is re-used, the rows of a pair that selects nothing - n_sel == 0 - keep
    // what the slot's previous step left there: rows >= n_sel are undefined, K2 reads n_sel rows only; the header says so.)
    // The engine's streams are non-blocking (not ordered against the null stream these memsets run on): wait for them here, once.
    for (int s = 0; s < cfg->n_slots; ++s)
        ok(hipMemset(e->L.slot[s].corrs, 0, (size_t)cfg->B * e->L.n_cap * 4 * sizeof(int32_t)));
    ok(hipDeviceSynchronize());
    if (err != hipSuccess) {
        set_error("oryon_engine_create: stream / event creation failed: %s", hipGetErrorString(err));
        oryon_engine_destroy(e);
        return ORYON_ERR_HIP;
    }
    *handle = e;
    return ORYON_OK;
}

extern "C" void oryon_engine_destroy(oryon_engine_t *e)
{
    if (!e) return;
    // the streams belong to the process-wide pool: drained here, never destroyed
    for (int s = 0; s < MAX_REG_STREAMS; ++s)
        if (e->sr[s]) (void)hipStreamSynchronize(e->sr[s]);
    for (int s = 0; s < MAX_SLOTS; ++s) {
        for (hipEvent_t ev : {e->ev_inputs[s], e->ev_gathered[s], e->ev_matched[s], e->ev_done[s]})
            if (ev) (void)hipEventDestroy(ev);
    }
    for (int r = 0; r < TIMING_RING; ++r)
        for (int i = 0; i < 9; ++i)
            if (e->tev[r][i]) (void)hipEventDestroy(e->tev[r][i]);
    if (e->sm) (void)hipStreamSynchronize(e->sm);
    if (e->sg) (void)hipStreamSynchronize(e->sg);
    for (int s = 0; s < MAX_SLOTS; ++s)
        if (e->ev_fb[s]) (void)hipEventDestroy(e->ev_fb[s]);
    if (e->fb_host) (void)hipHostFree(e->fb_host);
    delete e;
}

extern "C" int oryon_engine_set_timing(oryon_engine_t *e, int enable)
{
    ORYON_CHECK_ARG(e);
    e->timing = enable != 0;
    return ORYON_OK;
}

extern "C" int oryon_engine_set_stream_roles(oryon_engine_t *e, int roles)
{
    ORYON_CHECK_ARG(e && roles_valid(roles));
    if (roles == 0) roles = DEFAULT_ROLES;
    if (e->cfg.overlap == 0 || roles == e->roles) return ORYON_OK;
    // steps in flight were ordered with events between the OLD streams: drain them, then every later submit runs on the new ones (the
    // slot events stay valid - an event recorded on one stream may be waited for on any other)
    for (hipStream_t st : {e->sm, e->sg, e->sr[0], e->sr[1], e->sr[2], e->sr[3]})
        if (st) ORYON_CHECK_HIP(hipStreamSynchronize(st));
    return assign_roles(e, roles);
}

extern "C" int oryon_engine_stream_roles(const oryon_engine_t *e, int *roles)
{
    ORYON_CHECK_ARG(e && roles);
    *roles = e->roles;
    return ORYON_OK;
}

extern "C" int oryon_engine_buffer(const oryon_engine_t *e, int slot, const char *name, size_t *offset, size_t *bytes)
{
    ORYON_CHECK_ARG(e && name && offset && bytes && slot >= 0 && slot < e->cfg.n_slots);
    for (const Named &n : e->L.names[slot])
        if (strcmp(n.name, name) == 0) {
            *offset = n.off;
            *bytes = n.bytes;
            return ORYON_OK;
        }
    set_error("oryon_engine_buffer: no buffer named '%s'", name);
    return ORYON_ERR_INVALID_ARG;
}

extern "C" int oryon_engine_geometry(const oryon_engine_t *e, int *cap_a, int *cap_q, int *c_pad, int *n_cap)
{
    ORYON_CHECK_ARG(e);
    if (cap_a) *cap_a = e->L.cap_a;
    if (cap_q) *cap_q = e->L.cap_q;
    if (c_pad) *c_pad = e->L.c_pad;
    if (n_cap) *n_cap = e->L.n_cap;
    return ORYON_OK;
}

extern "C" int oryon_engine_submit(oryon_engine_t *e, const float *feat_a, const float *feat_q, const int32_t *mask_a, const int32_t *mask_q,
                                   const float *depth_a, const float *depth_q, const float *cam_a, const float *cam_q,
                                   const int64_t *pair_key, int force_eager, int inputs_resident, void *caller_stream)
{
    ORYON_CHECK_ARG(e && feat_a && feat_q && mask_a && mask_q && depth_a && depth_q && cam_a && cam_q);
    const auto t_host0 = std::chrono::steady_clock::now();
    const oryon_engine_config_t &c = e->cfg;
    const int slot = (int)(e->n_submit % c.n_slots);
    SlotBuf &b = e->L.slot[slot];
    GatherBuf &g = e->L.gbuf[e->n_submit % c.gather_sets];
    hipStream_t caller = as_stream(caller_stream);
    hipStream_t sm = c.overlap >= 1 ? e->sm : caller;
    hipStream_t sg = c.overlap >= 2 ? e->sg : sm;
    hipStream_t sr = c.overlap >= 1 ? e->sr[e->n_submit % e->n_reg_streams] : caller;
    const int HW = c.FH * c.FW, B = c.B;
    const bool timing = e->timing;
    hipEvent_t *tev = e->tev[e->n_submit % TIMING_RING];
    e->timed[e->n_submit % TIMING_RING] = timing;
    int rc;
    // ---- ordering in: the engine's streams start after the caller's inputs exist; a result slot is re-used only after the step
    // that last used it has completed (its registration wrote the slot's pose: everything before it is done as well); a K0 buffer
    // set only after the matcher of the step before last has read it
    if (c.overlap >= 1) {
        // the registration of this step overwrites the slot's pose / status_out: it is ordered after whatever the caller had queued
        // before this call (its reads of the slot's previous results).  The gather / match streams wait for the caller's stream only
        // when the inputs are not known to be complete already (inputs_resident): that wait is what keeps K0 of step k+1 from running
        // under the matching of step k when the caller's stream is itself waiting for step k-1.
        // The first n_slots submits always wait: the arena itself was handed over on the caller's stream (an allocator may have given
        // it memory whose previous owner still has work queued there), and the engine's streams are not otherwise ordered after it.
        // A caller that still has reads of the slot's OTHER buffers (everything but pose / status_out) queued on its stream when the
        // slot comes round again must pass inputs_resident = 0 for that submit (header: "slot lifetime"); oryon_amd/engine.py does.
        ORYON_CHECK_HIP(hipEventRecord(e->ev_inputs[slot], caller));
        ORYON_CHECK_HIP(hipStreamWaitEvent(sr, e->ev_inputs[slot], 0));
        if (!inputs_resident || e->n_submit < c.n_slots) {
            ORYON_CHECK_HIP(hipStreamWaitEvent(sg, e->ev_inputs[slot], 0));
            if (sm != sg) ORYON_CHECK_HIP(hipStreamWaitEvent(sm, e->ev_inputs[slot], 0));
        }
        if (e->used[slot]) {
            // the step that used this result slot (k - n_slots): K0 overwrites its ROI lists, the matcher what its registration read
            ORYON_CHECK_HIP(hipStreamWaitEvent(sg, e->ev_done[slot], 0));
            if (sm != sg) ORYON_CHECK_HIP(hipStreamWaitEvent(sm, e->ev_done[slot], 0));
        }
        // optional throttle (off by default: measured slower, DESIGN.md "step engine"): the matcher of step k starts only after the
        // registration of step k - reg_lag has finished
        if (c.reg_lag > 0 && e->n_submit >= c.reg_lag)
            ORYON_CHECK_HIP(hipStreamWaitEvent(sm, e->ev_done[(e->n_submit - c.reg_lag) % c.n_slots], 0));
        if (sg != sm && e->n_submit >= c.gather_sets)       // K0 overwrites the rows the matcher of step k - gather_sets read
            ORYON_CHECK_HIP(hipStreamWaitEvent(sg, e->ev_matched[(e->n_submit - c.gather_sets) % c.n_slots], 0));
    }
    // development aid (tools/ablate_native.sh): ORYON_ENGINE_ABLATE bit 0 / 1 / 2 leaves out K0's gathers / the matcher / the registration
    // once every buffer set has been filled by a complete step - the following steps then time the REST of the pipeline on stale buffers
    // (identical inputs every step, as in bench.py).  Never set in production: results are those of an earlier step.
    static const int ablate_env = dev_env_int("ORYON_ENGINE_ABLATE", 0);
    const int ablate = e->n_submit >= (int64_t)(c.n_slots + c.gather_sets) ? ablate_env : 0;
    // ---- K0 on the gather stream
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[0], sg));
    if ((rc = oryon_roi_compact(mask_a, B, HW, b.roi_a, b.n_a, sg))) return rc;
    if ((rc = oryon_roi_compact(mask_q, B, HW, b.roi_q, b.n_q, sg))) return rc;
    if (c.src_sampling > 0 && (rc = oryon_roi_subsample(b.roi_a, b.n_a, B, HW, c.src_sampling, c.seed, pair_key, sg))) return rc;
    const bool mx6 = c.screen == 1 && !force_eager;           // the eager route (complete min_dist / argmin arrays) keeps the int8 operands
    // "sample first" (cfg.sample_first = N, oryon_sample_first_gate): the matcher first sees a uniformly random N-anchor subset of every pair
    // (second-level device-RNG subsample, row order kept); only pairs whose subset holds fewer than n_corrs valid rows are redone on all
    // anchors - gated on the device, so the second K0 pass and the second matcher call see zero anchors for every other pair
    const bool sf = e->L.cap_a1 > 0 && !force_eager;
    const int32_t *roi_a_k0 = b.roi_a, *n_a_k0 = b.n_a;
    int cap_a_k0 = e->L.cap_a;
    int8_t *a8_k0 = g.a8;
    float *a_sc_k0 = g.a_sc, *a_eps_k0 = g.a_eps, *a_norm_k0 = g.a_norm, *a_hat_k0 = g.a_hat;
    if (sf) {
        ORYON_CHECK_HIP(hipMemcpyAsync(b.roi_a1, b.roi_a, (size_t)B * HW * sizeof(int32_t), hipMemcpyDeviceToDevice, sg));
        ORYON_CHECK_HIP(hipMemcpyAsync(b.n_a1, b.n_a, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, sg));
        if ((rc = oryon_roi_subsample(b.roi_a1, b.n_a1, B, HW, c.sample_first, c.seed ^ 0x5A17F125ull, pair_key, sg))) return rc;
        roi_a_k0 = b.roi_a1; n_a_k0 = b.n_a1; cap_a_k0 = e->L.cap_a1;
        a8_k0 = g.a8_1; a_sc_k0 = g.a_sc_1; a_eps_k0 = g.a_eps_1; a_norm_k0 = g.a_norm_1; a_hat_k0 = g.a_hat_1;
    }
    // cfg.x3_prefetch: what the most recently COMPLETED steps report (their n_und / n_a arrays in pinned memory, behind an event that is
    // only queried, never waited for) decides whether this step's K0 pass also writes the hi / lo rows of the second level
    bool x3_pre = false;
    if (e->fb_host && mx6) {
        for (int s = 0; s < c.n_slots; ++s) {
            if (e->fb_step[s] > e->fb_seen && hipEventQuery(e->ev_fb[s]) == hipSuccess) {
                const int32_t *h = e->fb_host + (size_t)s * 2 * B;
                long und = 0, all = 0;
                for (int i = 0; i < B; ++i) { und += h[i]; all += h[B + i]; }
                e->hard_mode = all > 0 && 4 * und > all;
                e->fb_seen = e->fb_step[s];
            }
        }
        x3_pre = e->hard_mode && !sf && g.q_hilo != nullptr;
    }
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[8], sg));          // the ROI kernels are behind us: K0's gather launches start here
    if (ablate & 1) {
    } else if (mx6) {
        // the row buffers hold 32-byte mx6 slots instead of int8 rows (same size); the per-map error norms go where eps_max went
        if (x3_pre) {
            if ((rc = oryon_gather_mx6_x3(feat_q, B, c.C, HW, c.layout, b.roi_q, HW, b.n_q, e->L.cap_q, e->L.c_pad, reinterpret_cast<uint8_t *>(g.q8),
                                          g.q_eps, g.q_norm, g.q_hilo, g.q_lo_max, c.round_f16, sg))) return rc;
            e->n_x3 += 1;
        } else
        if ((rc = oryon_gather_mx6(feat_q, B, c.C, HW, c.layout, b.roi_q, HW, b.n_q, e->L.cap_q, e->L.c_pad, reinterpret_cast<uint8_t *>(g.q8),
                                   g.q_eps, g.q_norm, nullptr, c.round_f16, sg))) return rc;
        if ((rc = oryon_gather_mx6(feat_a, B, c.C, HW, c.layout, roi_a_k0, HW, n_a_k0, cap_a_k0, e->L.c_pad, reinterpret_cast<uint8_t *>(a8_k0),
                                   a_eps_k0, a_norm_k0, a_hat_k0, c.round_f16, sg))) return rc;
    } else {
    if ((rc = oryon_gather_q8(feat_q, B, c.C, HW, c.layout, b.roi_q, HW, b.n_q, e->L.cap_q, e->L.c_pad, g.q8, g.q_sc, g.q_eps, g.q_norm,
                              nullptr, c.round_f16, sg))) return rc;
    if ((rc = oryon_gather_q8(feat_a, B, c.C, HW, c.layout, roi_a_k0, HW, n_a_k0, cap_a_k0, e->L.c_pad, a8_k0, a_sc_k0, a_eps_k0, a_norm_k0,
                              a_hat_k0, c.round_f16, sg))) return rc;
    }
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[1], sg));
    if (sg != sm) {
        ORYON_CHECK_HIP(hipEventRecord(e->ev_gathered[slot], sg));
        ORYON_CHECK_HIP(hipStreamWaitEvent(sm, e->ev_gathered[slot], 0));
    }
    // ---- K1s8 + K1b + K2 on the match stream (one shared workspace: matcher calls are serial on this stream)
    // the profile events are armed for the screening launch of THIS call only: whatever path leaves the function disarms them, so a
    // failed sub-call (or an ablated matcher) cannot leave them to be recorded by an unrelated matcher call of this thread later
    struct Disarm {
        bool armed = false;
        ~Disarm() { if (armed) (void)oryon_profile_events(nullptr, nullptr); }
    } disarm;
    if (timing) {
        ORYON_CHECK_HIP(hipEventRecord(tev[2], sm));
        (void)oryon_profile_events(tev[4], tev[5]);
        disarm.armed = true;
    }
    // corrs rows beyond max_corrs are never written by the sampler and K2 only reads n_sel rows: no zero-fill needed
    // one matcher call (the engine's screen setting) on the given anchor operands
    auto match_call = [&](const float *a_hat_, const int8_t *a8_, const float *a_sc_, const float *a_eps_, const int32_t *roi_a_, int cap_a_,
                          const int32_t *n_a_, int32_t *corrs_, int32_t *n_valid_, int32_t *n_sel_, int32_t *status_, int32_t *n_und_) -> int {
        if (mx6 && x3_pre)
            return oryon_match_corrs_mx6_x3(a_hat_, reinterpret_cast<const uint8_t *>(a8_), a_eps_, feat_q, c.C, HW, c.layout, roi_a_, HW, b.roi_q, HW,
                                            g.q_norm, reinterpret_cast<const uint8_t *>(g.q8), g.q_eps, g.q_hilo, g.q_lo_max, B, e->L.c_pad, cap_a_,
                                            e->L.cap_q, n_a_, b.n_q, c.dist_th, c.FW, c.n_corrs, e->L.n_cap, c.seed, pair_key, b.min_dist, b.argmin,
                                            b.valid, corrs_, n_valid_, n_sel_, status_, n_und_, c.round_f16, e->L.match_ws, e->L.match_ws_bytes, sm);
        if (mx6)
            return oryon_match_corrs_mx6(a_hat_, reinterpret_cast<const uint8_t *>(a8_), a_eps_, feat_q, c.C, HW, c.layout, roi_a_, HW, b.roi_q, HW,
                                         g.q_norm, reinterpret_cast<const uint8_t *>(g.q8), g.q_eps, B, e->L.c_pad, cap_a_, e->L.cap_q, n_a_, b.n_q,
                                         c.dist_th, c.FW, c.n_corrs, e->L.n_cap, c.seed, pair_key, b.min_dist, b.argmin, b.valid, corrs_, n_valid_,
                                         n_sel_, status_, n_und_, c.round_f16, e->L.match_ws, e->L.match_ws_bytes, sm);
        return oryon_match_corrs_i8(a_hat_, a8_, a_sc_, feat_q, c.C, HW, c.layout, roi_a_, HW, b.roi_q, HW, g.q_norm, g.q8, g.q_sc, g.q_eps, B,
                                    e->L.c_pad, cap_a_, e->L.cap_q, n_a_, b.n_q, c.dist_th, c.FW, c.n_corrs, e->L.n_cap, c.seed, pair_key, force_eager,
                                    b.min_dist, b.argmin, b.valid, corrs_, n_valid_, n_sel_, status_, n_und_, c.round_f16, e->L.match_ws,
                                    e->L.match_ws_bytes, sm);
    };
    if (ablate & 2) {
    } else if (sf) {
        if ((rc = match_call(g.a_hat_1, g.a8_1, g.a_sc_1, g.a_eps_1, b.roi_a1, e->L.cap_a1, b.n_a1, b.corrs, b.n_valid, b.n_sel, b.status, b.n_und)))
            return rc;
        if ((rc = oryon_sample_first_gate(b.n_valid, b.n_a1, b.n_a, B, c.n_corrs, b.n_a2, sm))) return rc;
        // second stage for the pairs that came up short: their anchors' operands (every other pair has a zero count: nothing is gathered or
        // matched for it), the matcher on all anchors, rows / counts / status of those pairs copied over the first stage's
        if (mx6) {
            if ((rc = oryon_gather_mx6(feat_a, B, c.C, HW, c.layout, b.roi_a, HW, b.n_a2, e->L.cap_a, e->L.c_pad, reinterpret_cast<uint8_t *>(g.a8),
                                       g.a_eps, g.a_norm, g.a_hat, c.round_f16, sm))) return rc;
        } else if ((rc = oryon_gather_q8(feat_a, B, c.C, HW, c.layout, b.roi_a, HW, b.n_a2, e->L.cap_a, e->L.c_pad, g.a8, g.a_sc, g.a_eps, g.a_norm,
                                         g.a_hat, c.round_f16, sm))) return rc;
        if ((rc = match_call(g.a_hat, g.a8, g.a_sc, g.a_eps, b.roi_a, e->L.cap_a, b.n_a2, b.corrs2, b.n_valid2, b.n_sel2, b.status2, nullptr)))
            return rc;
        if ((rc = oryon_sample_first_merge(b.n_a2, b.corrs2, b.n_valid2, b.n_sel2, b.status2, B, e->L.n_cap, b.corrs, b.n_valid, b.n_sel, b.status,
                                           sm))) return rc;
    } else if ((rc = match_call(g.a_hat, g.a8, g.a_sc, g.a_eps, b.roi_a, e->L.cap_a, b.n_a, b.corrs, b.n_valid, b.n_sel, b.status, b.n_und)))
        return rc;
    if (e->fb_host && mx6 && !(ablate & 2)) {
        int32_t *h = e->fb_host + (size_t)slot * 2 * B;
        ORYON_CHECK_HIP(hipMemcpyAsync(h, b.n_und, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
        ORYON_CHECK_HIP(hipMemcpyAsync(h + B, b.n_a, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
        ORYON_CHECK_HIP(hipEventRecord(e->ev_fb[slot], sm));
        e->fb_step[slot] = e->n_submit;
    }
    if (!(ablate & 2) && (rc = oryon_lift_pairs(b.corrs, b.n_sel, B, e->L.n_cap, c.FH, c.FW, depth_a, c.HA, c.WA, depth_q, c.HQ, c.WQ, cam_a, cam_q,
                                                b.status, b.pcd_a, b.pcd_q, b.n_lift, sm))) return rc;
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[3], sm));
    if (c.overlap >= 1) {
        ORYON_CHECK_HIP(hipEventRecord(e->ev_matched[slot], sm));
        ORYON_CHECK_HIP(hipStreamWaitEvent(sr, e->ev_matched[slot], 0));
    }
    // ---- K3-K10 on the slot's registration stream
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[6], sr));
    if (!(ablate & 4) && (rc = oryon_pointdsc_register(e->solver, b.pcd_a, b.pcd_q, b.n_lift, B, e->L.n_cap, b.status, b.pdsc_ws, e->L.pdsc_ws_bytes,
                                                       b.pose, nullptr, b.status_out, sr))) return rc;
    // the two per-pair counters a caller reads with the pose: copied on the registration stream (which is always ordered after the
    // caller's stream) into the slot's protected block, so that result views of them stay valid for the slot's whole lifetime
    hipLaunchKernelGGL(engine_counts_kernel, dim3((B + 255) / 256), dim3(256), 0, sr, b.n_valid, b.n_lift, B, b.n_valid_out, b.n_lift_out);
    ORYON_CHECK_LAUNCH();
    if (timing) ORYON_CHECK_HIP(hipEventRecord(tev[7], sr));
    if (c.overlap >= 1) ORYON_CHECK_HIP(hipEventRecord(e->ev_done[slot], sr));
    e->used[slot] = true;
    e->n_submit += 1;
    e->host_ns_last = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t_host0).count();
    e->host_ns_total += e->host_ns_last;
    return slot;
}

extern "C" int oryon_engine_wait(oryon_engine_t *e, int slot, void *caller_stream)
{
    ORYON_CHECK_ARG(e && slot >= 0 && slot < e->cfg.n_slots && e->used[slot]);
    if (e->cfg.overlap >= 1) ORYON_CHECK_HIP(hipStreamWaitEvent(as_stream(caller_stream), e->ev_done[slot], 0));
    return ORYON_OK;
}

extern "C" int oryon_engine_host_stats(const oryon_engine_t *e, int64_t *n_submit, double *submit_ms_total, double *submit_ms_last)
{
    ORYON_CHECK_ARG(e);
    if (n_submit) *n_submit = e->n_submit;
    if (submit_ms_total) *submit_ms_total = e->host_ns_total * 1e-6;
    if (submit_ms_last) *submit_ms_last = e->host_ns_last * 1e-6;
    return ORYON_OK;
}

extern "C" int oryon_engine_feedback(oryon_engine_t *e, int64_t *step, int64_t *n_undecided, int64_t *n_anchors)
{
    ORYON_CHECK_ARG(e && step && n_undecided && n_anchors);
    *step = -1;
    *n_undecided = *n_anchors = 0;
    if (!e->fb_host) return ORYON_OK;
    const int B = e->cfg.B;
    int best = -1;
    for (int s = 0; s < e->cfg.n_slots; ++s)
        if (e->fb_step[s] >= 0 && (best < 0 || e->fb_step[s] > e->fb_step[best]) && hipEventQuery(e->ev_fb[s]) == hipSuccess) best = s;
    if (best < 0) return ORYON_OK;
    const int32_t *h = e->fb_host + (size_t)best * 2 * B;
    int64_t und = 0, all = 0;
    for (int i = 0; i < B; ++i) { und += h[i]; all += h[B + i]; }
    *step = e->fb_step[best];
    *n_undecided = und;
    *n_anchors = all;
    return ORYON_OK;
}

extern "C" int oryon_engine_x3_steps(const oryon_engine_t *e, int64_t *n_steps)
{
    ORYON_CHECK_ARG(e && n_steps);
    *n_steps = e->n_x3;
    return ORYON_OK;
}

extern "C" int oryon_engine_timing(oryon_engine_t *e, int64_t step, float *out8)
{
    ORYON_CHECK_ARG(e && out8 && step >= 0 && step < e->n_submit && step >= e->n_submit - TIMING_RING);
    if (!e->timed[step % TIMING_RING]) { set_error("oryon_engine_timing: step %lld was submitted with timing off (oryon_engine_set_timing)", (long long)step); return ORYON_ERR_STATE; }
    hipEvent_t *t = e->tev[step % TIMING_RING];
    // durations of the three sections + the screening kernel, and the sections' start / end relative to the start of the gather
    const int q[8][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {0, 3}, {0, 6}, {0, 7}};
    for (int i = 0; i < 8; ++i) {
        const hipError_t err = hipEventElapsedTime(&out8[i], t[q[i][0]], t[q[i][1]]);
        if (err != hipSuccess) {
            set_error("oryon_engine_timing: %s (step still running?)", hipGetErrorString(err));
            return ORYON_ERR_STATE;
        }
    }
    return ORYON_OK;
}

extern "C" int oryon_engine_gather_ms(oryon_engine_t *e, int64_t step, float *ms)
{
    ORYON_CHECK_ARG(e && ms && step >= 0 && step < e->n_submit && step >= e->n_submit - TIMING_RING);
    if (!e->timed[step % TIMING_RING]) { set_error("oryon_engine_gather_ms: step %lld was submitted with timing off", (long long)step); return ORYON_ERR_STATE; }
    const hipError_t err = hipEventElapsedTime(ms, e->tev[step % TIMING_RING][8], e->tev[step % TIMING_RING][1]);
    if (err != hipSuccess) { set_error("oryon_engine_gather_ms: %s (step still running?)", hipGetErrorString(err)); return ORYON_ERR_STATE; }
    return ORYON_OK;
}

extern "C" int oryon_engine_elapsed(oryon_engine_t *e, int64_t step_a, int event_a, int64_t step_b, int event_b, float *ms)
{
    ORYON_CHECK_ARG(e && ms && event_a >= 0 && event_a < 8 && event_b >= 0 && event_b < 8);
    for (int64_t st : {step_a, step_b}) {
        ORYON_CHECK_ARG(st >= 0 && st < e->n_submit && st >= e->n_submit - TIMING_RING);
        if (!e->timed[st % TIMING_RING]) { set_error("oryon_engine_elapsed: step %lld was submitted with timing off", (long long)st); return ORYON_ERR_STATE; }
    }
    const hipError_t err = hipEventElapsedTime(ms, e->tev[step_a % TIMING_RING][event_a], e->tev[step_b % TIMING_RING][event_b]);
    if (err != hipSuccess) { set_error("oryon_engine_elapsed: %s (step still running?)", hipGetErrorString(err)); return ORYON_ERR_STATE; }
    return ORYON_OK;
}
