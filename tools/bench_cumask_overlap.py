"""Experiment: K0 gather of batch k+1 on a CU-masked stream beside the K1s8 matcher of batch k on the complementary CUs
(hipExtStreamCreateWithCUMask), against the serial schedule.  python tools/bench_cumask_overlap.py [gather_cus_per_32] [layout]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from bench import make_inputs

dev = torch.device("cuda", 0)
B, H, C = 64, 224, 256
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8            # CUs out of every 32 given to the gather stream
layout = sys.argv[2] if len(sys.argv) > 2 else "interleaved"
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

if layout == "interleaved":      # bit b -> (b % 8 = XCD?, b // 8 = CU in XCD) : take the first G of every 32 consecutive bits
    g_bits = [b for b in range(256) if (b % 32) < G]
elif layout == "strided":        # every (32 // G)-th bit
    g_bits = [b for b in range(256) if b % (32 // G) == 0]
else:                            # contiguous low bits
    g_bits = list(range(8 * G))
m_bits = [b for b in range(256) if b not in set(g_bits)]
s_g, s_m = masked_stream(g_bits), masked_stream(m_bits)

inp = make_inputs(B, H, C, 0, dev)
masks = torch.cat((inp["mask_a"], inp["mask_q"]), 0)
roi, cnt = ops.roi_compact(masks)
roi_a, roi_q, n_a, n_q = roi[:B], roi[B:], cnt[:B], cnt[B:]
ops.roi_subsample_(roi_a, n_a, 5000, 1, torch.arange(B, dtype=torch.int64, device=dev))
cap_a, cap_q = ops.round_up(5000, ops.ROW_PAD), ops.round_up(H * H, ops.ROW_PAD)

def gather():
    a = ops.gather_normalise_q8(inp["feat_a"], roi_a, n_a, cap_a, 256)
    q = ops.gather_normalise_q8(inp["feat_q"], roi_q, n_q, cap_q, 256)
    return a, q

def match(a, q):
    return ops.match_screened8(a[0], q[0], a[2], q[2], a[3], q[3], q[4], n_a, n_q, 0.25, C)

def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

torch.cuda.synchronize()
t_g = timeit(lambda: gather())
aq = gather()
t_m = timeit(lambda: match(*aq))
def on(stream, fn):
    def f():
        with torch.cuda.stream(stream): fn()
    return f
t_g_mask = timeit(on(s_g, gather))
t_m_mask = timeit(on(s_m, lambda: match(*aq)))
print(f"default stream: gather {t_g:.3f} ms, match {t_m:.3f} ms, serial {t_g + t_m:.3f} ms")
print(f"masked alone ({layout}, {8 * G} CUs gather / {256 - 8 * G} match): gather {t_g_mask:.3f} ms, match {t_m_mask:.3f} ms")

state = {"aq": gather()}
def pipelined(sg, sm):
    def f():
        cur = state["aq"]
        with torch.cuda.stream(sg):
            nxt = gather()
            ev = torch.cuda.Event(); ev.record(sg)
        with torch.cuda.stream(sm):
            match(*cur)
            sm.wait_event(ev)                  # the next step's matcher needs the gather
        state["aq"] = nxt
    return f
torch.cuda.synchronize()
print(f"pipelined, masked streams: {timeit(pipelined(s_g, s_m)):.3f} ms per step")
p1, p2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print(f"pipelined, plain streams:  {timeit(pipelined(p1, p2)):.3f} ms per step")
