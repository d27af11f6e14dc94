import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_test
from bench import build_solver
from oryon_amd.pipeline import Pipeline, default_args
dev = "cuda"; H, C, B = 192, 32, 8
args = default_args(**{"test.mask": "oracle", "model.image_encoder.img_size": [H, H], "dataset.img_size": [H, H]})
pipe = Pipeline(args, pointdsc_solver=build_solver(torch.device(dev)))
batch, pairs = run_test.synthetic_batch(0, B, H, C, dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.test_step(batch, 0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pipe.test_step_batched(batch, 0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"it{it}: per-sample loop {1e3*(t1-t0)/B:.2f} ms/pair, batched {1e3*(t2-t1)/B:.2f} ms/pair (B={B}, {H}x{H}, C={C})")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); pipe.test_step(batch, 0); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
