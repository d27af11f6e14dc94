"""Host submission time of one engine step vs its GPU time (is the headline loop host-bound?).  GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs, build_solver
import oryon_amd
oryon_amd.configure()
from oryon_amd.engine import MatchPoseEngine, MatchPoseConfig
dev = torch.device("cuda", 0)
B, H, C = 64, 224, 256
inputs = make_inputs(B, H, C, first=0, dev=dev)
main_stream = torch.cuda.Stream(device=dev) if os.environ.get("PROBE_SIDE_MAIN") else torch.cuda.current_stream(dev)
torch.cuda.set_stream(main_stream)
for overlap, lag in ((True, 1), (False, 1)):
    engine = MatchPoseEngine(build_solver(dev), MatchPoseConfig(dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1, match_mode="screened"),
                             overlap_registration=overlap)
    key = torch.arange(B, dtype=torch.int64, device=dev)
    def submit():
        return engine.run(inputs["feat_a"], inputs["feat_q"], inputs["mask_a"], inputs["mask_q"], inputs["depth_a"], inputs["depth_q"],
                          inputs["cam"], inputs["cam"], key, inputs_resident=True)
    for _ in range(3):
        engine.finish(submit())
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    outs = []
    for _ in range(n):
        o = submit()
        outs.append(o)
        if len(outs) > lag:
            engine.finish(outs.pop(0))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"overlap={overlap} lag={lag}: host submit {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
