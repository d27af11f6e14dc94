"""CPU-only tests: the C-ABI library loads and exports every symbol include/oryon_hip.h declares; the host
logic (sharding, collation over gloo with world_size 2, config, CSV line, state-dict layout) behaves; the product
path refuses to run without a GPU instead of falling back."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from oryon_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "oryon_hip.h")).read()
    declared = set(re.findall(r"\b(oryon_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"oryon_pointdsc_t", "oryon_pointdsc_config_t"}
    assert declared, "no declarations parsed"
    L = _lib.lib()                      # raises if the .so is missing or lacks a bound symbol
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/oryon_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b"gfx950" in L.oryon_version()


def test_argument_validation_without_gpu():
    from oryon_amd import _lib
    L = _lib.lib()
    assert L.oryon_roi_compact(None, 1, 16, None, None, None) == -1          # ORYON_ERR_INVALID_ARG, no launch attempted
    assert b"invalid argument" in L.oryon_last_error()
    assert L.oryon_match_f32(None, None, 1, 32, 128, 128, None, None, 0.25, None, None, None, None, 0, None) == -1
    assert L.oryon_match_workspace_bytes(512, 5120) == 0                      # batch alone gives >= 16 rounds: no query split
    assert L.oryon_match_workspace_bytes(64, 5120) == 64 * 4 * 5120 * 8       # cfg2: 4 splits x (fp32 + int32) per anchor row
    assert L.oryon_match_workspace_bytes(1, 2048) > 0
    # the newer entry points reject bad arguments the same way (no HIP call is reached)
    assert L.oryon_gather_normalise_q8(None, 1, 256, 16, None, 16, None, 256, 256, None, None, None, None, None, None) == -1
    assert L.oryon_match_screened8(None, None, None, None, None, None, None, 1, 256, 256, 256, 256, None, None, 0.25, None, None, None,
                                   None, None, 0, None) == -1
    assert L.oryon_match_screened(None, None, None, None, 1, 256, 256, 256, None, None, 0.25, None, None, None, None, 0, None) == -1
    assert L.oryon_add_layernorm_bf16(None, None, None, None, 4, 1024, 1e-5, None, None, None) == -1
    assert L.oryon_swin_window_attention_bf16(None, None, None, 1, 7, 7, 128, 4, 0, None, None) == -1
    assert L.oryon_match_screened8_workspace_bytes(64, 256, 5120, 50176) > L.oryon_match_screened_workspace_bytes(64, 256, 5120) > 0
    # round 4: the decoder handle and the fusion window attention
    assert L.oryon_decoder_create(None, None, None) == -1
    assert L.oryon_decoder_forward(None, None, None, None, 2, 24, 24, None, 0, None, None, 0, 0, None) == -1
    assert L.oryon_decoder_workspace_bytes(128, 24, 24) == 3 * 128 * 192 * 192 * 32 * 4 + L.oryon_decoder_workspace_bytes(128, 24, 24) % (128 * 192 * 192 * 32 * 4)
    assert L.oryon_decoder_workspace_bytes(128, 24, 20) == 0 and L.oryon_decoder_workspace_bytes(-1, 24, 24) == 0
    import ctypes
    off = (ctypes.c_int64 * 3)()
    assert L.oryon_decoder_workspace_layout(2, 24, 24, off) == 0 and list(off) == [0, 2 * 192 * 192 * 32 * 4, 2 * 2 * 192 * 192 * 32 * 4]
    assert L.oryon_decoder_workspace_layout(2, 24, 25, off) == -1
    assert L.oryon_fusion_window_attention_f32(None, None, 1, 24, 24, 128, 4, 12, 0, None, None) == -1
    assert L.oryon_conv24_image_bytes(128, 512, 3) == (16 * 9 + 1) * 2 * 4 * 2 * 1024 and L.oryon_conv24_image_bytes(128, 80, 7) == (3 * 49 + 1) * 2 * 4 * 2 * 1024
    assert L.oryon_conv24_image_bytes(96, 80, 7) == 0 and L.oryon_conv24_image_bytes(128, 80, 5) == 0
    assert L.oryon_conv24_f16x3(None, 1, 80, None, None, 128, 7, 0, None, None) == -1


def test_no_cpu_fallback():
    from oryon_amd import pcd
    from oryon_amd._lib import OryonError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = torch.randn(8, 4, 4)
    m = torch.ones(4, 4, dtype=torch.int32)
    with pytest.raises(OryonError):
        pcd.nn_correspondences(f, f, m, m, 0.25, 500, 5000, "cpu")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oryon_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace("oracle/oryon_oracle.c computes", ""), fn


def test_state_dict_layout_matches_reference_names():
    from oracle import oryon_oracle as orc
    from oryon_amd.pointdsc import PointDSC
    m = PointDSC(num_layers=3, num_channels=64)
    want = dict(orc.pointdsc_param_shapes(3, 64))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}


def test_shard_range_and_csv_line():
    from oryon_amd.dist import shard_range
    from oryon_amd.pipeline import Pipeline, default_args
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    cover = []
    for r in range(8):
        s, e = shard_range(1024, r, 8)
        cover += list(range(s, e))
    assert cover == list(range(1024))
    args = default_args(**{"test.mask": "oracle"})
    assert args.test.mask == "oracle" and args.test.dist_th == 0.25 and args.model.image_encoder.img_size == [192, 192]
    p = Pipeline(args)
    line = p.add_pred_pose("1 2 3", "1 5 3", np.float32(0.5), np.float32(0.25), np.eye(4))
    parts = line.strip().split(",")
    assert parts[0] == "1 2 3" and parts[1] == "1 5 3" and len(parts[2].split(" ")) == 12 and parts[3] == "0.5"


def test_synth_pair_is_a_rigid_pair():
    from oracle import oryon_oracle as orc
    from oryon_amd.synth import make_pair
    p = make_pair(5, 40, 40, 8)
    assert p["feat_a"].shape == (8, 40, 40) and p["mask_q"].dtype == torch.int32
    # lifting matched pixels with the generator's own geometry reproduces the ground-truth pose
    H = W = 40
    ys, xs = torch.nonzero(p["mask_q"] == 1, as_tuple=True)
    cam = p["camera"].reshape(9)
    Pq = orc.lift_points(p["depth_q"], cam, xs, ys) / 1000.0
    assert Pq.shape[0] > 100 and float(Pq[:, 2].min()) > 0.5


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from oryon_amd.dist import init_from_env, shard_range, gather_poses
rank, world, _ = init_from_env("cpu")
total = 5
s, e = shard_range(total, rank, world)
pose = torch.eye(4).repeat(e - s, 1, 1)
for i in range(e - s):
    pose[i, 0, 3] = float(s + i)
status = torch.tensor([(s + i) % 3 for i in range(e - s)], dtype=torch.int32)
P, S = gather_poses(pose, status, total)
assert P.shape == (total, 4, 4) and S.tolist() == [i % 3 for i in range(total)], S
assert P[:, 0, 3].tolist() == [float(i) for i in range(total)]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_collation_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


WORKER4 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from oryon_amd.dist import init_from_env, shard_range, gather_poses, gather_pose_windows
rank, world, _ = init_from_env("cpu")
assert world == 4
total = 10                                   # not divisible by 4: blocks of 3, 3, 3, 1
s, e = shard_range(total, rank, world)
assert (e - s) == (3 if rank < 3 else 1)
pose = torch.eye(4).repeat(e - s, 1, 1)
for i in range(e - s):
    pose[i, 0, 3] = float(s + i)
status = torch.tensor([(s + i) % 3 for i in range(e - s)], dtype=torch.int32)
P, S = gather_poses(pose, status, total)
assert P.shape == (total, 4, 4) and S.tolist() == [i % 3 for i in range(total)], S
assert P[:, 0, 3].tolist() == [float(i) for i in range(total)]
# the final collation of a window of k steps: short ranks pad their block to ceil(total / world) rows
k, per = 5, 3
stage = torch.zeros((k, per, 17))
for j in range(k):
    stage[j, : e - s, :16] = pose.reshape(e - s, 16)
    stage[j, : e - s, 7] = float(j)
    stage[j, :, 16] = -1.0
    stage[j, : e - s, 16] = status.to(torch.float32)
allr = gather_pose_windows(stage)
assert allr.shape == (4, k, per, 17)
for j in range(k):
    rows = allr[:, j].reshape(4 * per, 17)
    rows = rows[rows[:, 16] >= 0]            # padding rows cut by their status
    assert rows.shape[0] == total and rows[:, 3].tolist() == [float(i) for i in range(total)] and bool((rows[:, 7] == j).all())
    assert rows[:, 16].to(torch.int32).tolist() == [i % 3 for i in range(total)]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_collation_gloo_world4_total_not_divisible(tmp_path):
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr",
                        "127.0.0.1", "--master-port", "29633", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 4


def test_bench_final_collation_selftest_world2():
    """`bench.py --collate final`: the window's poses are staged and collated by ONE all_gather (the north_star's wording)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "5",
                        "--collation-selftest", "--collate", "final"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["collated_in_order"] is True and rec["collate"] == "final" and rec["collectives"] == 1


def test_stream_pool_is_created_before_the_process_group(monkeypatch):
    """the engine's stream roles are positions in the process's stream-creation order, so `init_from_env("cuda")` must
    create the pool (oryon_engine_warm_streams) BEFORE RCCL's communicator creates its streams - at world 1 as well."""
    import torch.distributed as dist
    from oryon_amd import dist as odist
    order = []
    monkeypatch.setattr(odist, "warm_engine_streams", lambda local: order.append(("warm", local)) or True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: order.append(("set_device", d)))
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: order.append(("init_process_group", a[0], k.get("device_id"))))
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert odist.init_from_env("cuda") == (1, 2, 1)
    kinds = [o[0] for o in order]
    assert kinds.index("warm") < kinds.index("init_process_group") and order[kinds.index("warm")] == ("warm", 1)
    assert order[kinds.index("init_process_group")][1] == "nccl"
    order.clear()
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert odist.init_from_env("cuda") == (0, 1, 0) and [o[0] for o in order] == ["warm"]


def test_stream_role_entry_points_validate_their_arguments():
    import ctypes
    from oryon_amd import _lib
    L = _lib.lib()
    assert L.oryon_engine_set_stream_roles(None, 2301) == -1 and L.oryon_engine_stream_roles(None, None) == -1
    cfg = _lib.EngineConfig(B=1, C=256, FH=8, FW=8, HA=8, WA=8, HQ=8, WQ=8, layout=0, dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1,
                            round_f16=0, n_slots=6, overlap=2, gather_sets=2, reg_streams=2, reg_lag=0, screen=1, sample_first=0,
                            x3_prefetch=1, stream_roles=2381)                # 8 is not a pool position
    assert L.oryon_engine_arena_bytes(ctypes.byref(cfg), ctypes.c_void_p(1)) == 0
    if not torch.cuda.is_available():
        assert L.oryon_engine_warm_streams() < 0                             # no device: an error code, not a crash


def test_bench_self_launches_for_gpus_2():
    """`python bench.py --gpus 2` (no torchrun, exactly how the driver invokes it) must become its own launcher: two ranks rendezvous on
    127.0.0.1, shard the pairs, run the collation and rank 0 prints ONE JSON line.  --collation-selftest swaps the GPU work for CPU
    tensors over gloo, everything else is the real control flow."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5",
                        "--collation-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert [l for l in r.stdout.splitlines() if l.strip()][-1] == lines[0], r.stdout        # ... and it is the LAST thing on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["global_pairs"] == 10 and rec["collated_in_order"] is True


def test_hot_kernels_do_not_spill():
    """The kernels of the cfg2 step keep their working set in registers (tools/check_kernel_resources.py reads the AMDGPU metadata of the
    built objects): a compiler-flag change that silently spilled 52 registers of the MX-fp6 screen cost 12 % in round 3."""
    import glob, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not glob.glob(os.path.join(root, "oryon_amd", "csrc", "*.o")):
        pytest.skip("objects not built in this tree")
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_kernel_resources as ckr
    rows = ckr.report()
    if not rows:
        pytest.skip("LLVM object tools not available")
    hot = ("match_mx6_screen_w4_kernelILi256ELi8E", "match_i8_screen_v2_kernelILi256E", "gather_q8_v3_kernelILi256ELi1ELb0ELi1E",
           "gather_q8_v3_kernelILi256ELi1ELb0ELi0E", "gather_mx6_v4_kernelILb0E", "gather_mx6_v4_kernelILb1E", "pdsc_attention_x3_img_kernel", "pdsc_pcn_qkv_x3_kernel", "pdsc_mlp3_x3_kernel",
           "match_decide_lite_kernel", "match_resolve_selected_kernel", "dec_conv3x3_kernelILi1ELb0ELb1ELb1E", "dec_conv3x3_kernelILi2ELb0ELb0ELb0E",
           "dec_final_kernel", "fusion_window_attention_x3_kernel",
           # round 6: the encoder's layer kernel (its chain's counted vmcnt waits assume no scratch traffic of the compiler's), the towers'
           # attention and stream linears, the second level's sweeps, the cascade's windowed screen
           "pdsc_att_chain_x3_kernelILi128ELb1ELb0E", "pdsc_att_chain_x3_kernelILi128ELb0ELb0E", "mha_x3_kernel",
           "linear_f16x3_stream_kernelILi0ELb0ELb0E", "linear_f16x3_stream_kernelILi0ELb0ELb1E", "linear_f16x3_stream_kernelILi1ELb0ELb0E",
           "linear_f16x3_stream_kernelILi0ELb1ELb0E", "match_x3_scan_kernelILi256ELi8ELb0E", "match_x3_sweep2_kernelILi256ELb0E",
           "match_mx6_screen_w4_win_kernelILi256ELi8ELi4E")
    seen = set()
    for k in rows:
        for h in hot:
            if h in k["name"]:
                seen.add(h)
                assert k["scratch"] == 0, f"{k['name']}: {k['spill']} spilled registers, {k['scratch']} B of scratch per lane"
    assert seen == set(hot), sorted(set(hot) - seen)


def test_shipped_library_reads_no_environment_variable():
    """liboryon_hip.so has no switch that changes (or skips) what it computes - no ORYON_* name among its strings and no
    import of getenv; the development build (make dev), which the variant-vs-variant tests and tools/ load explicitly, is where they live."""
    import re, subprocess
    from oryon_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "oryon_hip.h")).read()
    names = sorted(set(m.decode() for m in re.findall(rb"ORYON_[A-Z0-9_]{3,}", blob)))
    names = [n for n in names if n not in header]            # enum constants of the C ABI appear in argument-check messages
    assert names == [], names
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert not re.search(r"\bgetenv\b", nm.stdout), "liboryon_hip.so imports getenv"
    dev = os.path.join(os.path.dirname(_lib.LIB_PATH), "liboryon_hip_dev.so")
    if os.path.exists(dev):
        assert b"ORYON_AMB_X3" in open(dev, "rb").read()


def test_engine_config_struct_matches_the_library():
    """The ctypes mirror of oryon_engine_config_t has the size the library was built with (a field added on one side only would shift
    every later field silently)."""
    import ctypes
    from oryon_amd import _lib
    assert _lib.lib().oryon_engine_config_bytes() == ctypes.sizeof(_lib.EngineConfig)
    names = [f[0] for f in _lib.EngineConfig._fields_]
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "oryon_hip.h")).read()
    body = header[header.index("typedef struct {", header.index("typedef struct oryon_engine oryon_engine_t;")):header.index("} oryon_engine_config_t;")]
    import re
    decl = []
    for line in body.splitlines()[1:]:
        code = line.split("/*")[0]
        m = re.match(r"\s*(?:int|float|uint64_t)\s+([^;]+);", code)
        if m:
            decl += [x.strip() for x in m.group(1).split(",")]
    assert decl == names, (decl, names)


def test_default_stream_placement_is_the_same_in_python_header_and_library_source():
    """The placement the library falls back to (cfg.stream_roles == 0) is named in three places - they must agree, and the tuner must try it."""
    import re
    from oryon_amd.engine import MatchPoseEngine
    src = open(os.path.join(ROOT, "oryon_amd", "csrc", "engine.hip")).read()
    c_default = int(re.search(r"constexpr int DEFAULT_ROLES = (\d+);", src).group(1))
    header = open(os.path.join(ROOT, "include", "oryon_hip.h")).read()
    h_default = int(re.search(r"0 = the library's default \((\d+)\)", header).group(1))
    assert c_default == h_default == MatchPoseEngine.DEFAULT_ROLES
    assert MatchPoseEngine.DEFAULT_ROLES in MatchPoseEngine.ROLE_CANDIDATES
    for r in MatchPoseEngine.ROLE_CANDIDATES:                      # four digits, each a pool position
        assert 0 < r <= 7777 and all(int(d) < 8 for d in str(r))

