"""The C ABI's step engine (oryon_engine_*, csrc/engine.hip) against the per-call schedule of oryon_amd/engine.py: the same entry
points with the same arguments, so every result must be identical BIT FOR BIT - whatever the overlap mode, however many steps
are in flight, whatever the caller does with the slot views.  Replaces the per-sample loop of pipeline.py:313-355."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _solver(layers=2, ch=32):
    from oracle import oryon_oracle as orc
    from oryon_amd.pointdsc import PointDSC
    m = PointDSC(in_dim=6, num_layers=layers, num_channels=ch, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
    m.load_state_dict(orc.analytic_pointdsc_params(layers, ch), strict=True)
    return m.cuda().eval()


def _inputs(first, n, H, C, dev="cuda", nhwc=False):
    from oryon_amd.synth import make_pair
    pairs = [make_pair(first + i, H, H, C, device=dev) for i in range(n)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    fa, fq = st("feat_a"), st("feat_q")
    if nhwc:
        fa, fq = fa.contiguous(memory_format=torch.channels_last), fq.contiguous(memory_format=torch.channels_last)
    cam = st("camera").to(dev)
    return (fa, fq, st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"), cam, cam)


KEEP_KEYS = ("pose", "status", "n_valid", "n_lifted", "n_a", "n_q", "roi_a", "roi_q", "corrs", "pcd_a", "pcd_q", "valid")


@pytest.mark.parametrize("H,C,nhwc", [(56, 256, False), (48, 160, False), (40, 512, False), (56, 256, True),
                                      (192, 32, False), (96, 64, False), (64, 128, False), (64, 32, True)])
def test_native_step_equals_python_schedule(H, C, nhwc):
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    solver = _solver()
    ins = _inputs(100, 3, H, C, nhwc=nhwc)
    key = torch.arange(7, 10, device="cuda")
    for keep in (False, True):
        a = MatchPoseEngine(solver, MatchPoseConfig(), native=False).run(*ins, key, keep=keep)
        b = MatchPoseEngine(solver, MatchPoseConfig(), native=True).run(*ins, key, keep=keep)
        torch.cuda.synchronize()
        assert b["status"].tolist() == [0, 0, 0]
        for k in (KEEP_KEYS if keep else KEEP_KEYS[:4]):
            if k in ("roi_a", "roi_q"):          # entries beyond the count are scratch
                n = a["n_a" if k == "roi_a" else "n_q"].tolist()
                for i, ni in enumerate(n):
                    assert torch.equal(a[k][i, :ni], b[k][i, :ni]), k
            elif k in ("corrs", "pcd_a", "pcd_q"):
                assert torch.equal(a[k][:, :500], b[k][:, :500]), k
            elif k == "valid":
                for i, ni in enumerate(a["n_a"].tolist()):
                    assert torch.equal(a[k][i, :ni], b[k][i, :ni]), k
            else:
                assert torch.equal(a[k], b[k]), k


def test_native_half_descriptors_and_failures():
    """round_f16 route + pairs without a mask / without correspondences: status codes and identity poses as pipeline.py:335-350."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    solver = _solver()
    fa, fq, ma, mq, da, dq, cam, _ = _inputs(200, 4, 48, 256)
    ma = ma.clone(); fq = fq.clone()
    ma[1] = 0                                         # no anchor mask -> NO_MASK
    fq[2] = torch.randn_like(fq[2])                   # unrelated query descriptors -> NO_CORR
    cfg = MatchPoseConfig(half_descriptors=True)
    a = MatchPoseEngine(solver, cfg, native=False).run(fa, fq, ma, mq, da, dq, cam, cam)
    b = MatchPoseEngine(solver, cfg, native=True).run(fa, fq, ma, mq, da, dq, cam, cam)
    torch.cuda.synchronize()
    assert b["status"].tolist() == [0, 1, 2, 0]
    eye = torch.eye(4, device="cuda")
    assert torch.equal(b["pose"][1], eye) and torch.equal(b["pose"][2], eye)
    for k in ("pose", "status", "n_valid", "n_lifted"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("overlap", [(False, False), (True, False), (True, True)])
def test_native_pipelined_many_steps_in_flight(overlap):
    """12 steps submitted back to back before anything is collected (more than the two slots hold: the engine collects a slot's
    results itself before re-using it), inputs alternately resident / behind an event / plain: all equal the serial python schedule."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    solver = _solver()
    sets = [_inputs(300 + 2 * i, 2, 64, 256) for i in range(4)]
    keys = [torch.arange(8 * i, 8 * i + 2, device="cuda") for i in range(12)]     # inputs_resident covers the keys too: made up front
    ref = MatchPoseEngine(solver, MatchPoseConfig(), native=False)
    want = [ref.run(*sets[i % 4], keys[i]) for i in range(12)]
    torch.cuda.synchronize()
    eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=overlap[0], overlap_gather=overlap[1], native=True)
    outs = []
    for i in range(12):
        kw = {}
        if i % 3 == 0:
            kw["inputs_resident"] = True
        elif i % 3 == 1:
            ev = torch.cuda.Event()
            ev.record()
            kw["inputs_event"] = ev
        outs.append(eng.run(*sets[i % 4], keys[i] if i % 3 == 0 else keys[i].clone(), **kw))
    for o in outs:
        eng.finish(o)
    torch.cuda.synchronize()
    for i, (o, w) in enumerate(zip(outs, want)):
        for k in ("pose", "status", "n_valid", "n_lifted"):
            assert torch.equal(o[k], w[k]), (i, k)


def test_native_result_views_and_timing():
    """result_views=True hands out views of the slot buffers (no allocation); the engine's HIP-event timing and host statistics."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    solver = _solver()
    ins = list(_inputs(400, 2, 64, 256))
    ins[6] = ins[7] = ins[6].reshape(2, 9).float().contiguous()       # the C ABI's own input types: nothing to convert per step
    ref = MatchPoseEngine(solver, MatchPoseConfig(), native=False).run(*ins)
    eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=True, overlap_gather=True, native=True, result_views=True)
    eng.native_timing = True
    o0 = eng.finish(eng.run(*ins, inputs_resident=True))
    o1 = eng.finish(eng.run(*ins, inputs_resident=True))
    torch.cuda.synchronize()
    nat = eng._native
    assert o0["pose"].data_ptr() == nat.view(0, "pose").data_ptr() and o1["pose"].data_ptr() == nat.view(1, "pose").data_ptr()
    assert torch.equal(o0["pose"], ref["pose"]) and torch.equal(o1["pose"], ref["pose"])
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(6):
        o = eng.finish(eng.run(*ins, inputs_resident=True))
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before, "the native step allocated through torch"
    assert torch.equal(o["pose"], ref["pose"])
    t = nat.timing(nat.steps - 1)
    assert 0 < t["screen_kernel_ms"] <= t["match_ms"] and t["gather_ms"] > 0 and t["registration_ms"] > 0
    assert t["registration_end"] >= t["match_end"] >= t["match_start"] >= 0
    n, tot, last = nat.host_stats()
    assert n == 8 and 0 < last < 50 and tot >= last
    from oryon_amd._lib import lib
    assert lib().oryon_dominant_kernel().decode().startswith("match_mx6_screen")


def test_engine_c_abi_argument_checks():
    """Bad configurations / arenas are refused with an error code and a message, never a crash."""
    from oryon_amd import _lib
    from oryon_amd._lib import lib
    solver = _solver()
    solver._ensure_handle(torch.device("cuda", 0))
    cfg = _lib.EngineConfig(B=2, C=32, FH=16, FW=16, HA=16, WA=16, HQ=16, WQ=16, layout=0, dist_th=0.25, n_corrs=500, src_sampling=5000,
                            seed=1, round_f16=0, n_slots=2, overlap=2, gather_sets=2, reg_streams=2, reg_lag=0, screen=0)
    assert lib().oryon_engine_arena_bytes(ctypes.byref(cfg), solver._handle) > 0           # the reference's own width: zero-padded rows
    cfg.C = 513
    assert lib().oryon_engine_arena_bytes(ctypes.byref(cfg), solver._handle) == 0          # wider than the screening kernels
    cfg.C = 256
    need = lib().oryon_engine_arena_bytes(ctypes.byref(cfg), solver._handle)
    assert need > 0
    arena = torch.empty((need,), dtype=torch.uint8, device="cuda")
    h = ctypes.c_void_p()
    assert lib().oryon_engine_create(ctypes.byref(h), ctypes.byref(cfg), solver._handle, arena.data_ptr(), need - 1) == -3
    assert b"arena too small" in lib().oryon_last_error()
    cfg.n_slots = 1
    assert lib().oryon_engine_create(ctypes.byref(h), ctypes.byref(cfg), solver._handle, arena.data_ptr(), need) == -1
    cfg.n_slots = 2
    assert lib().oryon_engine_create(ctypes.byref(h), ctypes.byref(cfg), solver._handle, arena.data_ptr(), need) == 0
    off, nb = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib().oryon_engine_buffer(h, 0, b"pose", ctypes.byref(off), ctypes.byref(nb)) == 0 and nb.value == 2 * 64
    assert lib().oryon_engine_buffer(h, 0, b"nope", ctypes.byref(off), ctypes.byref(nb)) == -1
    assert lib().oryon_engine_wait(h, 0, None) == -1                                          # nothing submitted to that slot yet
    lib().oryon_engine_destroy(h)


def test_native_sample_first_schedule_equals_python_schedule():
    """oryon_engine_config_t.sample_first: the same two-stage schedule as the Python engine's (random N-anchor subset first, pairs that come
    up short redone on all anchors - one pair here has a mask that is too thin for the subset), same RNG keys: identical poses, statuses
    and correspondences; `keep` steps (complete matcher outputs wanted) ignore it in both."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    solver = _solver()
    ins = list(_inputs(400, 4, 128, 256))
    ins[2] = ins[2].clone()
    live = ins[2][3].view(-1).nonzero().flatten()
    ins[2][3].view(-1)[live[1400:]] = 0               # pair 3: 1400 anchors, of which a 1024-subset holds < 500 valid rows at times -> second stage
    key = torch.arange(40, 44, device="cuda")
    cfg = MatchPoseConfig(sample_first=1024)
    a = MatchPoseEngine(solver, cfg, native=False).run(*ins, key)
    b = MatchPoseEngine(solver, cfg, native=True).run(*ins, key)
    torch.cuda.synchronize()
    for k in ("pose", "status", "n_valid", "n_lifted"):
        assert torch.equal(a[k], b[k]), k
    plain = MatchPoseEngine(solver, MatchPoseConfig(), native=True).run(*ins, key)
    torch.cuda.synchronize()
    assert b["status"].tolist() == [0, 0, 0, 0] and torch.equal(plain["status"], b["status"])
    assert int(b["n_a"].min()) > 1024 if "n_a" in b else True
    assert not torch.equal(plain["pose"], b["pose"])                           # another (equally valid) sample
    assert float((plain["pose"] - b["pose"]).abs().max()) < 5e-2
    ak = MatchPoseEngine(solver, cfg, native=False).run(*ins, key, keep=True)
    bk = MatchPoseEngine(solver, cfg, native=True).run(*ins, key, keep=True)
    torch.cuda.synchronize()
    assert torch.equal(ak["pose"], bk["pose"]) and torch.equal(bk["pose"], plain["pose"])


def test_native_x3_prefetch_gives_identical_results_on_smooth_fields():
    """oryon_engine_config_t.x3_prefetch: on smooth descriptor fields (every sampled anchor goes to the fp16x3 second level) the engine's
    K0 pass starts writing the hi / lo rows itself once the first steps' counts have come back (oryon_gather_mx6_x3 +
    oryon_match_corrs_mx6_x3) - poses, statuses, correspondences and lifted points must stay bit for bit those of an engine that never
    prefetches, and of the first step (which could not have prefetched)."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    solver = _solver()
    pairs = [make_pair(500 + i, 96, 96, 256, device="cuda", smooth=0.02) for i in range(3)]
    st = lambda k: torch.stack([p[k] for p in pairs]).contiguous()
    cam = st("camera").reshape(3, 9).float().cuda().contiguous()
    ins = (st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"), cam, cam)
    key = torch.arange(3, device="cuda")
    outs = {}
    for pre in (0, 1):
        eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=True, overlap_gather=True, native=True)
        eng.native_geometry["x3_prefetch"] = pre
        res = []
        for _ in range(8):
            o = eng.finish(eng.run(*ins, key, keep=False))
            torch.cuda.synchronize()                      # every step's counts are back before the next submit looks at them
            res.append({k: o[k].clone() for k in ("pose", "status", "n_valid", "n_lifted")})
        outs[pre] = res
        n_x3 = eng._native.x3_steps()
        assert (n_x3 >= 6) if pre else (n_x3 == 0), n_x3
        assert res[0]["status"].tolist() == [0, 0, 0]
    for i in range(8):
        for k in ("pose", "status", "n_valid", "n_lifted"):
            assert torch.equal(outs[0][i][k], outs[1][i][k]), (i, k)
            assert torch.equal(outs[1][i][k], outs[1][0][k]), (i, k)


@pytest.mark.gpu
def test_validity_cascade_with_unsettled_panels_gives_identical_results():
    """Round 6: on the route the engine takes once its feedback says "hard" (oryon_match_corrs_mx6_x3) the screen runs as a cascade - a
    windowed first launch that settles the validity of whole panels, the complete scan only for panels with an open anchor, a complete
    second pass for the sampled anchors.  A batch of two smooth pairs (every panel settles in the window) and one pair of the generator's
    Gaussian descriptors (anchors without a counterpart: its panels never settle and take the gated complete scan) must give, bit for
    bit, what the engine gives without the cascade (x3_prefetch off: the plain full screen) - poses, statuses, counts."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    solver = _solver()
    pairs = [make_pair(700, 96, 96, 256, device="cuda", smooth=0.02), make_pair(701, 96, 96, 256, device="cuda", smooth=0.02),
             make_pair(702, 96, 96, 256, device="cuda")]
    st = lambda k: torch.stack([p[k] for p in pairs]).contiguous()
    cam = st("camera").reshape(3, 9).float().cuda().contiguous()
    ins = (st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"), cam, cam)
    key = torch.arange(3, device="cuda")
    outs = {}
    for pre in (0, 1):
        eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=True, overlap_gather=True, native=True)
        eng.native_geometry["x3_prefetch"] = pre
        res = []
        for _ in range(6):
            o = eng.finish(eng.run(*ins, key, keep=False))
            torch.cuda.synchronize()
            res.append({k: o[k].clone() for k in ("pose", "status", "n_valid", "n_lifted")})
        outs[pre] = res
        if pre:
            assert eng._native.x3_steps() >= 4             # two thirds of the anchors are "undecided": the feedback switched the route
        assert res[0]["status"].tolist() == [0, 0, 0]
    for i in range(6):
        for k in ("pose", "status", "n_valid", "n_lifted"):
            assert torch.equal(outs[0][i][k], outs[1][i][k]), (i, k)
    # the Gaussian pair has anchors without a counterpart: fewer valid rows than anchors, i.e. its panels could not settle in the window
    assert int(outs[1][5]["n_valid"][2]) < int(outs[1][5]["n_valid"][0])
