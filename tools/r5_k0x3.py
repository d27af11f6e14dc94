"""Round 5: the fused K0 pass (mx6 slots + hi / lo half rows, oryon_gather_mx6_x3) at cfg2 size - timing and output fingerprints,
one variant per process (ORYON_K0V4=1|0, development library).   python tools/r5_k0x3.py <label> [B]"""
import json, os, sys
import _devlib  # noqa: F401
import torch
from oryon_amd import ops
from oryon_amd.ops import check, lib, ptr, stream_ptr
from oryon_amd.synth import make_pair

label = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, C = 224, 256
dev = torch.device("cuda", 0)
pairs = [make_pair(i, H, H, C, device=dev) for i in range(B)]
feat_q = torch.stack([p["feat_q"] for p in pairs]); mask_q = torch.stack([p["mask_q"] for p in pairs])
del pairs
roi_q, nq = ops.roi_compact(mask_q)
cap_q = ops.round_up(H * H, 256)
out6 = torch.zeros((B, cap_q, 256), dtype=torch.uint8, device=dev)
err = torch.zeros((B,), dtype=torch.float32, device=dev)
norm = torch.zeros((B, cap_q), dtype=torch.float32, device=dev)
hilo = torch.zeros((2, B, cap_q, 256), dtype=torch.float16, device=dev)
losq = torch.zeros((B,), dtype=torch.float32, device=dev)


def run():
    check(lib().oryon_gather_mx6_x3(feat_q.data_ptr(), B, C, H * H, 0, ptr(roi_q), roi_q.shape[1], ptr(nq), cap_q, 256, ptr(out6), ptr(err),
                                    ptr(norm), ptr(hilo), ptr(losq), 0, stream_ptr(dev)), "oryon_gather_mx6_x3")


run(); torch.cuda.synchronize()
fp = {"rows": 0, "hi": 0, "lo": 0, "norm": 0, "err": err.cpu().tolist(), "losq": losq.cpu().tolist()}
for m in range(B):
    k = int(nq[m]); kf = (k + 255) // 256 * 256
    fp["rows"] += int(out6[m, :kf].contiguous().view(torch.int64).sum().item())
    fp["norm"] += int(norm[m, :kf].contiguous().view(torch.int32).to(torch.int64).sum().item())
    # hi / lo rows live in 32-row tiles: whole tiles below k are fully written
    kt = k // 32 * 32
    fp["hi"] += int(hilo[0, m, :kt].contiguous().view(torch.int64).sum().item())
    fp["lo"] += int(hilo[1, m, :kt].contiguous().view(torch.int64).sum().item())
ts = []
for _ in range(3):
    run()
torch.cuda.synchronize()
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
out = {"label": label, "ms": ts[len(ts) // 2], "ms_min": ts[0], "fp": fp}
print(json.dumps({k: v for k, v in out.items() if k != "fp"}), {k: fp[k] for k in ("rows", "hi", "lo", "norm")}, "err0", fp["err"][0], "losq0", fp["losq"][0])
json.dump(out, open(os.path.join(_devlib.ROOT, "gpurun_out", f"r5_k0x3_{label}.json"), "w"))
