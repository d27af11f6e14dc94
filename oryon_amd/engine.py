"""Batched match -> lift -> registration engine: the whole per-sample loop of FPM_Pipeline.test_step
(pipeline.py:313-355) for B pairs at once, with no host synchronisation between stages.

    masks -> ROI (K0) -> subsample <=5000 (K0, device RNG) -> gather+normalise (K0) -> cosine NN (K1)
          -> sample 500 correspondences (K1b, device RNG) -> scale/validate/lift (K2) -> PointDSC (K3-K10)

Differences from the drop-in per-sample facade (oryon_amd.pcd / oryon_amd.pointdsc), by design:
  * the two random draws use a counter-based device RNG keyed by (seed, global pair index), so results do
    not depend on how pairs are sharded over GPUs; the reference's host torch.multinomial stream is kept
    only in the per-sample facade;
  * failures are reported as per-pair status codes (ORYON_PAIR_*) with an identity pose, exactly the values
    pipeline.py:335-350 writes.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import check, lib
from .pointdsc import PointDSC

PAIR_OK, PAIR_NO_MASK, PAIR_NO_CORR = 0, 1, 2


@dataclass
class MatchPoseConfig:
    """Flag names follow configs/config.yaml of the reference (test.* section)."""
    dist_th: float = 0.25          # test.dist_th
    n_corrs: int = 500             # test.n_corrs (= dataset.max_corrs)
    src_sampling: Optional[int] = 5000   # test.src_sampling
    seed: int = 1                  # seed (pipeline.py:296-299)
    # "screened":   int8 pre-screen (K1s8, C_pad 256 / 512) -> fp16 screening (K1s) -> exact fp32 re-scoring; identical valid
    #               set / argmin / min_dist on valid rows for every input, used when 64 < C <= 512
    # "screened16": the same without the int8 stage
    # "exact":      full fp32-MFMA scan (K1) for every row
    match_mode: str = "screened"
    # True = the reference's half-descriptor branch (utils/pcd.py:195-197, corrs_device='cuda'; BASELINE configs[4] "fp16 descriptors"):
    # every descriptor value is rounded to float16 before anything else; the contraction itself is unchanged
    half_descriptors: bool = False
    # > 0: "sample first" schedule of the int8 route - the matcher runs on a uniformly random subset of this many anchors per pair first;
    # since only n_corrs correspondences leave it, drawn uniformly from the valid rows (utils/pcd.py:205-214), a subset that already holds
    # >= n_corrs valid rows gives an identically distributed sample at a fraction of the contraction.  Pairs whose subset came up short
    # are redone on all of their anchors (device-gated, no host round trip).  Off (0) by default: the default route computes the
    # validity of every one of the <= src_sampling anchors, as the reference does.
    sample_first: int = 0


class NativeStep:
    """`oryon_engine_*` of the C ABI (csrc/engine.hip): the whole step - K0 on a gather stream, K1s8 + K1b + K2 on a match stream,
    K3-K10 on a registration stream per slot - enqueued by ONE C call over a persistent arena.  Nothing is allocated per step and no
    torch stream / event object is created; the arena is one torch uint8 tensor (torch is the memory plumbing), the slot buffers the
    results come back in are views of it.  A slot's views stay valid until the `n_slots`-th (6th) next submit."""

    _SLOT_VIEWS = {  # name -> (dtype, shape builder)
        "pose": (torch.float32, lambda g: (g["B"], 4, 4)), "status_out": (torch.int32, lambda g: (g["B"],)),
        "n_valid": (torch.int32, lambda g: (g["B"],)), "n_lift": (torch.int32, lambda g: (g["B"],)),
        "n_valid_out": (torch.int32, lambda g: (g["B"],)), "n_lift_out": (torch.int32, lambda g: (g["B"],)),
        "n_a": (torch.int32, lambda g: (g["B"],)), "n_q": (torch.int32, lambda g: (g["B"],)), "n_und": (torch.int32, lambda g: (g["B"],)),
        "n_sel": (torch.int32, lambda g: (g["B"],)), "status": (torch.int32, lambda g: (g["B"],)),
        "roi_a": (torch.int32, lambda g: (g["B"], g["HW"])), "roi_q": (torch.int32, lambda g: (g["B"], g["HW"])),
        "min_dist": (torch.float32, lambda g: (g["B"], g["cap_a"])), "argmin": (torch.int32, lambda g: (g["B"], g["cap_a"])),
        "valid": (torch.uint8, lambda g: (g["B"], g["cap_a"])), "corrs": (torch.int32, lambda g: (g["B"], g["n_cap"], 4)),
        "pcd_a": (torch.float32, lambda g: (g["B"], g["n_cap"], 3)), "pcd_q": (torch.float32, lambda g: (g["B"], g["n_cap"], 3)),
    }

    def __init__(self, solver: PointDSC, cfg: "MatchPoseConfig", key: Tuple, dev: torch.device, overlap: int, n_slots: int = 6,
                 gather_sets: int = 2, reg_streams: int = 2, reg_lag: int = 0, screen: int = 1, x3_prefetch: int = 1,
                 stream_roles: int = 0):
        B, C, FH, FW, HA, WA, HQ, WQ, layout = key
        self.key, self.dev = key, dev
        if overlap >= 2:
            from . import configure
            configure()                 # no-op when the host program configured the process; warns when HIP started with too few queues
        solver._ensure_handle(dev)
        self._solver = solver                                   # keeps the C handle alive
        self.ecfg = _lib.EngineConfig(B=B, C=C, FH=FH, FW=FW, HA=HA, WA=WA, HQ=HQ, WQ=WQ, layout=layout, dist_th=cfg.dist_th,
                                      n_corrs=cfg.n_corrs, src_sampling=int(cfg.src_sampling or 0), seed=int(cfg.seed) & (2**64 - 1),
                                      round_f16=int(cfg.half_descriptors), n_slots=n_slots, overlap=overlap,
                                      gather_sets=min(gather_sets, n_slots), reg_streams=reg_streams,
                                      reg_lag=reg_lag if overlap else 0, screen=screen,
                                      sample_first=int(cfg.sample_first) if cfg.sample_first and cfg.sample_first > 0 else 0,
                                      x3_prefetch=int(bool(x3_prefetch)) if (C <= 256 and screen == 1) else 0,
                                      stream_roles=int(stream_roles))
        self.cfg_sig = (cfg.dist_th, cfg.n_corrs, cfg.src_sampling, cfg.seed, cfg.half_descriptors, overlap, int(cfg.sample_first))
        need = lib().oryon_engine_arena_bytes(ctypes.byref(self.ecfg), solver._handle)
        if need == 0:
            raise _lib.OryonError(f"oryon_engine_arena_bytes: {lib().oryon_last_error().decode()}")
        with torch.cuda.device(dev):
            self.arena = torch.empty((need,), dtype=torch.uint8, device=dev)
            self._h = ctypes.c_void_p()
            check(lib().oryon_engine_create(ctypes.byref(self._h), ctypes.byref(self.ecfg), solver._handle, self.arena.data_ptr(), need),
                  "oryon_engine_create")
        ca, cq, cp, nc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().oryon_engine_geometry(self._h, ctypes.byref(ca), ctypes.byref(cq), ctypes.byref(cp), ctypes.byref(nc)))
        self.geo = dict(B=B, HW=FH * FW, cap_a=ca.value, cap_q=cq.value, c_pad=cp.value, n_cap=nc.value)
        self.n_slots = n_slots
        self.steps = 0                                          # submits so far; step k used slot k % n_slots
        self._views = [dict() for _ in range(n_slots)]
        # slots whose buffers OTHER than pose / status_out the caller may still read asynchronously (views handed out by a `keep` step,
        # copies / statistics queued on the caller's stream): their next submit must order K0 / the matcher after the caller's
        # stream (include/oryon_hip.h, "slot lifetime") - i.e. it cannot claim inputs_resident
        self.reads_pending = [None] * n_slots          # None | True | torch.cuda.Event

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                torch.cuda.synchronize(self.dev)
                lib().oryon_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def view(self, slot: int, name: str) -> Tensor:
        v = self._views[slot].get(name)
        if v is None:
            off, nbytes = ctypes.c_size_t(), ctypes.c_size_t()
            check(lib().oryon_engine_buffer(self._h, slot, name.encode(), ctypes.byref(off), ctypes.byref(nbytes)), "oryon_engine_buffer")
            dtype, shape = self._SLOT_VIEWS[name]
            v = self.arena[off.value: off.value + nbytes.value].view(dtype).view(shape(self.geo))
            self._views[slot][name] = v
        return v

    def set_timing(self, on: bool) -> None:
        check(lib().oryon_engine_set_timing(self._h, int(on)))

    def stream_roles(self) -> int:
        """Pool positions (creation order) of the match / gather / registration 0 / registration 1 streams, four digits."""
        r = ctypes.c_int()
        check(lib().oryon_engine_stream_roles(self._h, ctypes.byref(r)))
        return r.value

    def set_stream_roles(self, roles: int) -> None:
        """Move the live engine to another stream placement (drains the steps in flight: warm-up only)."""
        with torch.cuda.device(self.dev):
            check(lib().oryon_engine_set_stream_roles(self._h, int(roles)), "oryon_engine_set_stream_roles")

    def timing(self, step: int) -> Dict[str, float]:
        """ms: durations of the gather / match+lift / screening-kernel / registration sections of submit number `step` (0-based, one
        of the last 64) and the sections' positions relative to the start of its gather (the step must have completed)."""
        out = (ctypes.c_float * 8)()
        check(lib().oryon_engine_timing(self._h, int(step), out), "oryon_engine_timing")
        names = ("gather_ms", "match_ms", "screen_kernel_ms", "registration_ms", "match_start", "match_end", "registration_start",
                 "registration_end")
        return dict(zip(names, (float(x) for x in out)))

    def elapsed(self, step_a: int, event_a: int, step_b: int, event_b: int) -> float:
        """ms between two timing events (0/1 gather, 2/3 match + lift, 4/5 screening kernel, 6/7 registration: begin / end) of two steps."""
        ms = ctypes.c_float()
        check(lib().oryon_engine_elapsed(self._h, int(step_a), event_a, int(step_b), event_b, ctypes.byref(ms)), "oryon_engine_elapsed")
        return float(ms.value)

    def gather_ms(self, step: int) -> float:
        """ms of the two K0 gather launches of submit `step` (the gather section without the ROI kernels)."""
        ms = ctypes.c_float()
        check(lib().oryon_engine_gather_ms(self._h, int(step), ctypes.byref(ms)), "oryon_engine_gather_ms")
        return float(ms.value)

    def x3_steps(self) -> int:
        """Submits so far whose K0 pass also wrote the second level's hi / lo rows (oryon_engine_config_t.x3_prefetch)."""
        n = ctypes.c_int64()
        check(lib().oryon_engine_x3_steps(self._h, ctypes.byref(n)))
        return n.value

    def feedback(self) -> Optional[Tuple[int, int, int]]:
        """(step, undecided anchors, anchors) of the newest completed step from the engine's own pinned-memory feedback (MX-fp6 screen),
        or None - no device copy, no read of a slot buffer, never waits."""
        st, und, na = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(lib().oryon_engine_feedback(self._h, ctypes.byref(st), ctypes.byref(und), ctypes.byref(na)))
        return None if st.value < 0 else (st.value, und.value, na.value)

    def host_stats(self) -> Tuple[int, float, float]:
        n, tot, last = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        check(lib().oryon_engine_host_stats(self._h, ctypes.byref(n), ctypes.byref(tot), ctypes.byref(last)))
        return n.value, tot.value, last.value

    def submit(self, feat_a, feat_q, mask_a, mask_q, depth_a, depth_q, cam_a, cam_q, pair_key, force_eager: bool, inputs_resident: bool) -> int:
        with torch.cuda.device(self.dev):
            slot = lib().oryon_engine_submit(self._h, feat_a.data_ptr(), feat_q.data_ptr(), mask_a.data_ptr(), mask_q.data_ptr(),
                                             depth_a.data_ptr(), depth_q.data_ptr(), cam_a.data_ptr(), cam_q.data_ptr(),
                                             None if pair_key is None else pair_key.data_ptr(), int(force_eager), int(inputs_resident),
                                             torch.cuda.current_stream(self.dev).cuda_stream)
        if slot < 0:
            check(slot, "oryon_engine_submit")
        self.steps += 1
        return slot

    def next_slot(self) -> int:
        return self.steps % self.n_slots

    def wait(self, slot: int) -> None:
        check(lib().oryon_engine_wait(self._h, slot, torch.cuda.current_stream(self.dev).cuda_stream), "oryon_engine_wait")


class MatchPoseEngine:
    def __init__(self, solver: PointDSC, cfg: Optional[MatchPoseConfig] = None, overlap_registration: bool = False,
                 overlap_gather: bool = False, native: bool = True, result_views: bool = False):
        """result_views: `finish` leaves the native step's results as views of its slot buffers
        instead of copying pose / status / counts (four tiny tensors) out of the arena: the allocation-free mode of bench.py
        (valid until the sixth-next `run`).
        native: on the screened route (match_mode "screened", C <= 512) the whole step is enqueued by ONE call of the C ABI's
        step engine (`oryon_engine_submit`, csrc/engine.hip) over a persistent arena - no torch allocation, stream or event per step;
        the returned tensors are views of the slot buffers and stay valid until the sixth-next `run`.  False keeps the per-call
        schedule below (same entry points, same results bit for bit); other routes always take it.
        overlap_registration: run the registration stage (K3-K10: many small, latency-bound launches) on a second HIP
        stream so that it overlaps with the matching stage of the NEXT batch submitted by the caller; `run` then returns
        immediately after queueing and `finish(out)` makes the caller's stream wait for the poses.
        overlap_gather: additionally run K0 (ROI + gather/normalise: HBM-bound) on its own stream, so that the gather of
        the next batch overlaps the MFMA-bound screening of the current one."""
        self.solver = solver
        self.cfg = cfg or MatchPoseConfig()
        self.n_cap = ops.round_up(self.cfg.n_corrs, 128)
        self.overlap = overlap_registration
        self.overlap_gather = overlap_gather
        self.native = native
        self.result_views = result_views
        self._native: Optional[NativeStep] = None
        self._inflight: Dict[int, Dict[str, Tensor]] = {}       # slot -> result dict of the native step that last used it
        self.native_geometry = dict(n_slots=6, gather_sets=2, reg_streams=2, reg_lag=0, screen=1, x3_prefetch=1, stream_roles=0)      # NativeStep's pipeline depth / placement (see oryon_engine_config_t)
        self.native_timing = False          # bracket the sections of every native step with HIP events (NativeStep.timing)
        self._reg_stream = None
        self.reg_streams = 2
        self._gather_stream = None
        # adaptive use of the int8 pre-screen: the fraction of anchors it had to hand to the fp16 stage is read back
        # asynchronously (pinned buffer + event, never a sync); above `i8_max_undecided` the next batches skip the int8 stage
        # and it is tried again every `i8_retry_every` batches.  Exactness never depends on this - only the run time does.
        # (round 2: the lazy tail resolves ambiguous anchors with an exact scan of the sampled rows only, which is cheaper than the fp16
        # route on every distribution tried, so the back-off is off by default - thresholds >= 1 never trigger)
        self.i8_max_undecided = 2.0
        self.collect_i8_stats = False       # measurement aid: keep `_i8_frac` up to date even though the back-off is off
        self.i8_retry_every = 16
        self._i8_pending = None        # (pinned [2] int64 tensor, event)
        self._i8_host = None           # pinned buffer, allocated once
        self._i8_frac = 0.0
        self._i8_skipped = 0

    # What tune_stream_roles tries (digits: matcher, gather, registration 0, registration 1 = positions in the stream pool).  The runtime
    # multiplexes a process's streams onto a few hardware queues (how the command processor arbitrates between them is not documented), so
    # which pool stream plays which role decides whether the pipeline overlaps at all - and the good choices depend on what ELSE the
    # process created before its first tensor.  Measured at 16 pairs per step over 16 placements each (tools/pg_probe4.py):
    # plain process - gather on pool stream 7 costs 13 %, everything else within 2 %; with a one-rank RCCL communicator created first (what
    # every rank of an N > 1 run has) - registration 0 on pool stream 0 costs 70 % (1.44 -> 2.43 ms; round 5's default 2301 did that).
    # The candidates avoid both; 2345 is the library default since round 6.
    DEFAULT_ROLES = 2345
    ROLE_CANDIDATES = (2345, 6345, 2341, 6341, 2301)

    def tune_stream_roles(self, run_steps, candidates: Optional[Tuple[int, ...]] = None, steps: int = 8, warm: int = 2) -> Dict[str, object]:
        """Pick the engine's stream placement by MEASUREMENT on this process (warm-up only, never inside a timed region).

        `run_steps(n)` must submit and collect n complete steps of the caller's workload through this engine.  Every candidate gets
        `warm` untimed steps and `steps` timed ones between two device synchronisations; the fastest stays.  Needs the native engine to
        exist (run one step first).  Returns {"roles": chosen, "ms_per_step": {candidate: ms}, "default": library default}.
        Results never depend on the placement - only which hardware queue each engine stream sits on does."""
        import time
        nat = self._native
        if nat is None or nat.ecfg.overlap == 0:
            return {"roles": None, "ms_per_step": {}, "default": self.DEFAULT_ROLES}
        cands = tuple(candidates or self.ROLE_CANDIDATES)
        seen: Dict[int, float] = {}
        for r in cands:
            nat.set_stream_roles(r)
            run_steps(warm)
            torch.cuda.synchronize(nat.dev)
            t0 = time.perf_counter()
            run_steps(steps)
            torch.cuda.synchronize(nat.dev)
            seen[r] = (time.perf_counter() - t0) / steps * 1e3
        best = min(seen, key=seen.get)
        # keep the default unless another placement is clearly (> 1 %) faster: the timing noise of 8 steps is about that
        if self.DEFAULT_ROLES in seen and seen[self.DEFAULT_ROLES] <= seen[best] * 1.01:
            best = self.DEFAULT_ROLES
        nat.set_stream_roles(best)
        self.native_geometry["stream_roles"] = best          # a rebuilt NativeStep keeps the choice
        return {"roles": best, "ms_per_step": {str(k): round(v, 4) for k, v in seen.items()}, "default": self.DEFAULT_ROLES}

    def finish(self, out: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Order the caller's current stream after the registration of `out` (no-op without overlap)."""
        slot = out.pop("_native_slot", None)
        if slot is not None:
            self._native.wait(slot)
            out.pop("_inputs", None)
            queued = False
            if self.collect_i8_stats or self.i8_max_undecided < 1.0:
                # (a `keep` step takes the eager int8 route, which the engine's feedback does not cover: read its counters as before -
                # such a step marks its slot as having reads pending anyway)
                fb = self._native.feedback() if len(out) <= 4 else None
                if fb is not None and fb[2] > 0:
                    # the engine's own pinned-memory feedback (MX-fp6 screen): no copy is queued behind the step - queued reads of the
                    # slot's counters made its next submit wait for the caller's stream (a hole in the pipeline: bench.py's hard run
                    # read 5.8 ms per step where the engine alone runs 5.4)
                    self._i8_frac = fb[1] / fb[2]
                elif self._i8_pending is None and (len(out) > 4 or self._native.ecfg.screen != 1):
                    self._queue_i8_stats(self._native.view(slot, "n_und"), self._native.view(slot, "n_a"))
                    queued = True
            if not self.result_views:
                for k, v in list(out.items()):           # the slot buffers are re-used n_slots steps later: hand out copies
                    if isinstance(v, Tensor):
                        out[k] = v.clone()
                # pose / status / n_valid / n_lifted come out of the protected block: only a `keep` step's extra buffers need the event
                queued = queued or len(out) > 4
            elif len(out) > 4:
                # views of a `keep` step (more than pose / status / n_valid / n_lifted): the caller may read them whenever it likes
                self._native.reads_pending[slot] = True
            if queued and self._native.reads_pending[slot] is not True:
                # reads of the slot's unprotected buffers now sit on the caller's stream: remember where they end.  If they have
                # completed by the time the slot comes round again (the usual case, n_slots steps later) nothing needs ordering
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self._native.dev))
                self._native.reads_pending[slot] = ev
            if self._inflight.get(slot) is out:
                del self._inflight[slot]
            return out
        ev = out.pop("_done", None)
        if ev is not None:
            cur = torch.cuda.current_stream(out["pose"].device)
            cur.wait_event(ev)
            for k in ("pose", "status"):
                out[k].record_stream(cur)
        return out

    def _py_mark(self, idx: int, stream) -> None:
        """Dev aid (tools/engine_timeline.py): with `py_timeline` set to a list, the per-call schedule records a timing event at the
        section boundaries of every step (same indices as oryon_engine_elapsed: 0/1 gather, 2/3 match + lift, 6/7 registration)."""
        tl = getattr(self, "py_timeline", None)
        if tl is None:
            return
        if idx == 0:
            tl.append({})
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        tl[-1][idx] = ev

    def _queue_i8_stats(self, n_und: Tensor, n_a: Tensor) -> None:
        """Asynchronous read-back of (undecided anchors, anchors) per pair on the current stream: pinned buffer + event, never a sync."""
        B, dev = n_a.shape[0], n_a.device
        if self._i8_host is None:
            self._i8_host = torch.empty((2, B), dtype=torch.int32, pin_memory=True)
            self._i8_dev = torch.empty((2, B), dtype=torch.int32, device=dev)
        if self._i8_host.shape[1] == B:
            # two device-to-device copies into one staging tensor + one async D2H: no torch arithmetic in the step
            self._i8_dev[0].copy_(n_und)
            self._i8_dev[1].copy_(n_a)
            self._i8_host.copy_(self._i8_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._i8_pending = (self._i8_host, ev)

    def _collect_inflight(self, slot: Optional[int] = None) -> None:
        """Native steps whose results the caller has not collected yet and whose slot (all slots: None) is about to be re-used:
        `finish` them now, handing out copies."""
        for s_ in ([slot] if slot is not None else list(self._inflight)):
            stale = self._inflight.pop(s_, None)
            if stale is not None and "_native_slot" in stale:
                views, self.result_views = self.result_views, False
                try:
                    self.finish(stale)
                finally:
                    self.result_views = views

    def _run_native(self, feat_a, feat_q, mask_a, mask_q, depth_a, depth_q, cam_a, cam_q, pair_key, keep, inputs_event, inputs_resident, dev):
        """The step through `oryon_engine_submit`.  Inputs are handed over as they are when they already have the C ABI's types
        (fp32 maps in NCHW or channels_last storage, int32 masks, fp32 depth, fp32 [B,9] intrinsics); anything else is converted on the
        caller's stream first (a torch allocation: keep the inputs in those types to stay allocation-free)."""
        cfg = self.cfg
        B, C, FH, FW = feat_a.shape
        converted = False

        def as_type(t, dtype, shape=None):
            nonlocal converted
            if shape is not None:
                t = t.reshape(shape)
            if t.dtype != dtype or not t.is_contiguous():
                converted = True
                t = t.to(dtype).contiguous()
            return t

        feat_a, lay_a = ops.map_layout(feat_a)
        feat_q, lay_q = ops.map_layout(feat_q)
        if lay_a != lay_q:
            feat_q, lay_q = feat_q.contiguous(memory_format=torch.channels_last if lay_a == ops.LAYOUT_NHWC else torch.contiguous_format), lay_a
            converted = True
        assert feat_a.dtype == torch.float32 and feat_q.dtype == torch.float32
        mask_a, mask_q = as_type(mask_a, torch.int32, (B, FH * FW)), as_type(mask_q, torch.int32, (B, FH * FW))
        depth_a, depth_q = as_type(depth_a, torch.float32), as_type(depth_q, torch.float32)
        cam_a, cam_q = as_type(cam_a, torch.float32, (B, 9)), as_type(cam_q, torch.float32, (B, 9))
        key = (B, C, FH, FW, depth_a.shape[1], depth_a.shape[2], depth_q.shape[1], depth_q.shape[2], lay_a)
        overlap = 2 if (self.overlap and self.overlap_gather) else (1 if self.overlap else 0)
        sig = (cfg.dist_th, cfg.n_corrs, cfg.src_sampling, cfg.seed, cfg.half_descriptors, overlap, int(cfg.sample_first))
        nat = self._native
        if nat is None or nat.key != key or nat.cfg_sig != sig or nat.dev != dev:
            self._collect_inflight()                  # results still living in the old arena
            # (a rebuilt engine runs on the same HIP streams as the one it replaces: the C side takes them from a per-device pool that lives
            # as long as the process - where the runtime places NEW streams on its hardware queues differs from build to build, and the
            # same hard cfg2 step was measured at 5.45 ms on the first engine of a process and 6.4 ms on the third before the pool)
            self._native = None                       # release the previous arena before sizing the new one
            del nat
            nat = self._native = NativeStep(self.solver, cfg, key, dev, overlap, **self.native_geometry)
            nat.set_timing(self.native_timing)
        if inputs_event is not None:
            torch.cuda.current_stream(dev).wait_event(inputs_event)
            inputs_resident = False
        # the slot this step takes still holds the results of a step the caller has not collected: collect them now (copies)
        self._collect_inflight(nat.next_slot())
        nslot = nat.next_slot()
        pend = nat.reads_pending[nslot]
        if pend is not None and pend is not True and pend.query():
            pend = None                               # the reads queued at `finish` have completed
        resident = inputs_resident and not converted and pend is None
        nat.reads_pending[nslot] = None
        slot = nat.submit(feat_a, feat_q, mask_a, mask_q, depth_a, depth_q, cam_a, cam_q, pair_key, keep, resident)
        # the engine's streams read the inputs asynchronously: the result dict keeps them alive until `finish` has ordered the caller's
        # stream after the step (an input freed earlier could be handed to a new tensor and overwritten while the step still reads it)
        # all four are views of the slot's PROTECTED block (the engine orders their next overwrite after the caller's stream): the two
        # counters are the registration stream's copies of the matcher's n_valid / the lift's n_lift, not the unprotected originals
        out = dict(pose=nat.view(slot, "pose"), status=nat.view(slot, "status_out"), n_valid=nat.view(slot, "n_valid_out"),
                   n_lifted=nat.view(slot, "n_lift_out"), _native_slot=slot,
                   _inputs=(feat_a, feat_q, mask_a, mask_q, depth_a, depth_q, cam_a, cam_q, pair_key))
        if keep:
            for k in ("roi_a", "roi_q", "n_a", "n_q", "min_dist", "argmin", "valid", "corrs", "pcd_a", "pcd_q"):
                out[k] = nat.view(slot, k)
        self._inflight[slot] = out
        if not self.overlap:
            self.finish(out)                          # serial schedule: everything ran on the caller's stream already
        return out

    @torch.no_grad()
    def run(self, feat_a: Tensor, feat_q: Tensor, mask_a: Tensor, mask_q: Tensor, depth_a: Tensor, depth_q: Tensor,
            cam_a: Tensor, cam_q: Tensor, pair_key: Optional[Tensor] = None, keep: bool = False,
            inputs_event: Optional["torch.cuda.Event"] = None, inputs_resident: bool = False) -> Dict[str, Tensor]:
        """feat_* [B,C,FH,FW] fp32, mask_* [B,FH,FW] int (==1 selects), depth_* [B,H,W] fp32 mm, cam_* [B,3,3] or [B,9].
        Returns pose [B,4,4] fp32 (identity on failure), status [B] int32, n_valid [B], n_lifted [B].
        With overlap_gather the gather stream starts after `inputs_event` (the producer's event), or immediately when
        `inputs_resident` says the inputs (ALL of them, `pair_key` included) were complete before this call; by default it waits for
        everything queued on the caller's stream so far (always correct, but then it cannot overlap the previous batch's matching)."""
        dev = _lib.require_gpu(feat_a.device)
        cfg = self.cfg
        B, C, FH, FW = feat_a.shape
        main = torch.cuda.current_stream(dev)
        # the screens' validity cut is 1 - 2*dist_th: thresholds above 0.5 (or non-positive) take the exact scan
        screened = cfg.match_mode in ("screened", "screened16") and 64 < C <= 512 and 0.0 < cfg.dist_th <= 0.5
        use_i8 = screened and cfg.match_mode == "screened" and C > 128
        if use_i8:
            if self._i8_pending is not None and self._i8_pending[1].query():
                h = self._i8_pending[0]
                self._i8_frac = int(h[0].sum()) / max(1, int(h[1].sum()))
                self._i8_pending = None
            if self._i8_frac > self.i8_max_undecided:
                self._i8_skipped += 1
                if self._i8_skipped < self.i8_retry_every:
                    use_i8 = False
                else:
                    self._i8_skipped, self._i8_frac = 0, 0.0
        # the native step engine serves every width up to 512 on the screened route (narrow maps zero-padded to 256 channels by K0);
        # the per-call schedule below keeps its own choice per width (exact scan for C <= 64, fp16 screen for C <= 128)
        native_ok = cfg.match_mode == "screened" and C <= 512 and 0.0 < cfg.dist_th <= 0.5 and (use_i8 or C <= 128)
        if native_ok and self.native:
            return self._run_native(feat_a, feat_q, mask_a, mask_q, depth_a, depth_q, cam_a, cam_q, pair_key, keep, inputs_event,
                                    inputs_resident, dev)
        if pair_key is None:
            pair_key = torch.arange(B, dtype=torch.int64, device=dev)
        if self.overlap_gather:
            if self._gather_stream is None:
                self._gather_stream = torch.cuda.Stream(device=dev)
            if inputs_event is None and not inputs_resident:
                inputs_event = torch.cuda.Event()
                inputs_event.record(main)
            gctx = torch.cuda.stream(self._gather_stream)
            gctx.__enter__()
            self._py_mark(0, self._gather_stream)
            if inputs_event is not None:
                self._gather_stream.wait_event(inputs_event)
        if cfg.half_descriptors and not use_i8:
            # every route but the int8 one takes pre-rounded maps (one extra pass; K0v3 rounds on the way in).  The rounding runs on
            # the stream that consumes it (the gather stream when there is one, after its wait for the inputs)
            feat_a, feat_q = ops.round_to_f16(feat_a), ops.round_to_f16(feat_q)
        # two launches on the callers' tensors instead of a torch.cat + one launch: no torch arithmetic inside the step
        roi_a, n_a = ops.roi_compact(mask_a.reshape(B, FH, FW))
        roi_q, n_q = ops.roi_compact(mask_q.reshape(B, FH, FW))
        if cfg.src_sampling is not None:
            ops.roi_subsample_(roi_a, n_a, cfg.src_sampling, cfg.seed, pair_key)
            cap_a = ops.round_up(min(cfg.src_sampling, FH * FW), ops.ROW_PAD)
        else:
            cap_a = ops.round_up(FH * FW, ops.ROW_PAD)
        cap_q = ops.round_up(FH * FW, ops.ROW_PAD)
        a16 = q16 = a8 = q8 = a_sc = q_sc = q_eps = q_norm = q_hat = roi_a1 = n_a1 = None
        stage1 = use_i8 and cfg.sample_first > 0 and not keep
        if use_i8:
            # K0v3: anchors -> fp32 + int8 rows, queries -> int8 rows + row norms only (the re-scoring pass reads its few candidates
            # from the raw map); contiguous and channels_last maps are both read in place
            c_pad = 256 if C <= 256 else 512
            q8, q_sc, q_eps, q_norm, _ = ops.gather_q8(feat_q, roi_q, n_q, cap_q, c_pad, round_f16=cfg.half_descriptors)
            if stage1:
                # first-stage anchors: a random subset (second-level device-RNG subsample, row-major order kept) of the pair's anchors
                roi_a1, n_a1 = roi_a.clone(), n_a.clone()
                ops.roi_subsample_(roi_a1, n_a1, cfg.sample_first, cfg.seed ^ 0x5A17F125, pair_key)
                cap_a1 = ops.round_up(min(cfg.sample_first, FH * FW), ops.ROW_PAD)
                a8, a_sc, _, _, a_hat = ops.gather_q8(feat_a, roi_a1, n_a1, cap_a1, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
            else:
                a8, a_sc, _, _, a_hat = ops.gather_q8(feat_a, roi_a, n_a, cap_a, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
        elif screened:
            c_pad = 128 if C <= 128 else (256 if C <= 256 else 512)
            a_hat, a16 = ops.gather_normalise(feat_a.contiguous(), roi_a, n_a, cap_a, c_pad=c_pad, want_f16=True)
            q_hat, q16 = ops.gather_normalise(feat_q.contiguous(), roi_q, n_q, cap_q, c_pad=c_pad, want_f16=True)
        else:
            a_hat = ops.gather_normalise(feat_a.contiguous(), roi_a, n_a, cap_a)
            q_hat = ops.gather_normalise(feat_q.contiguous(), roi_q, n_q, cap_q)
        if self.overlap_gather:
            gathered = torch.cuda.Event()
            self._py_mark(1, self._gather_stream)
            gathered.record(self._gather_stream)
            gctx.__exit__(None, None, None)
            main.wait_event(gathered)
            self._py_mark(2, main)
            # EVERY tensor allocated under the gather stream and read on the main stream: without the record the allocator would hand
            # its memory to the next batch's gather (which runs ahead) while this batch's matcher still reads it
            rounded = (feat_a, feat_q) if cfg.half_descriptors and not use_i8 else ()
            for t_ in (roi_a, roi_q, n_a, n_q, a_hat, q_hat, a16, q16, a8, q8, a_sc, q_sc, q_eps, q_norm, roi_a1, n_a1) + rounded:
                if t_ is not None:
                    t_.record_stream(main)
        if use_i8:
            # lazy K1s8 + K1b: validity of every anchor from the int8 bound, exact argmin for the sampled anchors only; `keep` (the
            # caller wants the complete min_dist / argmin arrays) forces the eager route
            n_und = torch.empty((B,), dtype=torch.int32, device=dev)
            if stage1:
                corrs, n_valid, n_sel, status, min_dist, argmin, valid = ops.match_corrs_i8(
                    a_hat, a8, a_sc, feat_q, roi_a1, roi_q, q_norm, q8, q_sc, q_eps, n_a1, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, n_undecided=n_und, round_f16=cfg.half_descriptors)
                # second stage, gated on the device: all anchors of the pairs whose subset held fewer than n_corrs valid rows
                n_a2 = ops.sample_first_gate(n_valid, n_a1, n_a, cfg.n_corrs)
                a8b, a_scb, _, _, a_hatb = ops.gather_q8(feat_a, roi_a, n_a2, cap_a, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
                c2, nv2, ns2, st2, _, _, _ = ops.match_corrs_i8(
                    a_hatb, a8b, a_scb, feat_q, roi_a, roi_q, q_norm, q8, q_sc, q_eps, n_a2, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, round_f16=cfg.half_descriptors)
                ops.sample_first_merge_(n_a2, c2, nv2, ns2, st2, corrs, n_valid, n_sel, status)
            else:
                corrs, n_valid, n_sel, status, min_dist, argmin, valid = ops.match_corrs_i8(
                    a_hat, a8, a_sc, feat_q, roi_a, roi_q, q_norm, q8, q_sc, q_eps, n_a, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, force_eager=keep, n_undecided=n_und, round_f16=cfg.half_descriptors)
            # statistics for the back-off (skip the int8 stage while most anchors come back undecided): only collected when the
            # back-off can trigger at all (i8_max_undecided < 1; off by default - the lazy tail handles such inputs on the device)
            if self._i8_pending is None and (self.i8_max_undecided < 1.0 or self.collect_i8_stats):
                self._queue_i8_stats(n_und, n_a)
        else:
            if screened:
                min_dist, argmin, valid = ops.match_screened(a_hat, q_hat, a16, q16, n_a, n_q, cfg.dist_th)
            else:
                min_dist, argmin, valid = ops.match(a_hat, q_hat, n_a, n_q, cfg.dist_th)
            corrs, n_valid, n_sel, status = ops.select_corrs(roi_a, roi_q, n_a, n_q, argmin, valid, FW, cfg.n_corrs, cfg.seed,
                                                             pair_key, corr_rows=self.n_cap)
        cam_a = cam_a.reshape(B, 9).to(torch.float32).contiguous()
        cam_q = cam_q.reshape(B, 9).to(torch.float32).contiguous()
        pcd_a, pcd_q, n_lift = ops.lift_pairs(corrs, n_sel, (FH, FW), depth_a, depth_q, cam_a, cam_q, status)
        if self.overlap:
            # registrations of consecutive batches alternate between two streams (and two workspaces): their many small,
            # low-occupancy launches then overlap each other as well as the next batch's matching
            if self._reg_stream is None:
                self._reg_stream = [torch.cuda.Stream(device=dev) for _ in range(self.reg_streams)]
                self._reg_turn = 0
            slot = self._reg_turn % len(self._reg_stream)
            self._reg_turn += 1
            rs = self._reg_stream[slot]
            ready = torch.cuda.Event()
            self._py_mark(3, main)
            ready.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(rs):
                rs.wait_event(ready)
                for t_ in (pcd_a, pcd_q, n_lift, status):
                    t_.record_stream(rs)
                self._py_mark(6, rs)
                pose, _, status_out = self.solver.register(pcd_a, pcd_q, n_lift, status, ws_slot=slot)
                self._py_mark(7, rs)
                done = torch.cuda.Event()
                done.record(rs)
        else:
            pose, _, status_out = self.solver.register(pcd_a, pcd_q, n_lift, status)
            done = None
        out = dict(pose=pose, status=status_out, n_valid=n_valid, n_lifted=n_lift)
        if done is not None:
            out["_done"] = done
        if keep:
            out.update(roi_a=roi_a, roi_q=roi_q, n_a=n_a, n_q=n_q, min_dist=min_dist, argmin=argmin, valid=valid, corrs=corrs,
                       pcd_a=pcd_a, pcd_q=pcd_q)
        return out
