"""CPU-side checks: the C-ABI library loads and exports every symbol include/eadgan_hip.h declares (no compute calls
without a GPU), argument errors surface as RuntimeError, the product path refuses to run without a GPU, and the
data-parallel gradient averaging works across 2 gloo ranks."""
import ctypes
import importlib
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    return importlib.import_module("ead-gan_amd")


def test_library_exports_every_declared_symbol():
    eg = _pkg()
    protos = eg._lib.parse_header()
    assert len(protos) >= 45
    cdll = ctypes.CDLL(eg._lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/eadgan_hip.h but not exported"
    for required in ("eg_conv_fwd", "eg_conv_bwd_data", "eg_conv_wgrad", "eg_wgrad_reduce_sn", "eg_bn_fwd_train", "eg_bn_bwd",
                     "eg_sn_power_iter", "eg_adam_step", "eg_warp_affine", "eg_loss_affine_rpqxy", "eg_im2col_img"):
        assert required in protos
    assert eg._lib.lib().query("eg_version") >= 100


def test_argument_errors_do_not_abort():
    eg = _pkg()
    bad = eg.ops.make_conv(2, 12, 12, 32, 32, 4, 2, 1)            # 12 is not a power of two
    with pytest.raises(RuntimeError, match="powers of two"):
        eg._lib.lib().call("eg_conv_fwd", ctypes.byref(bad), 0, 1, 1, 1, None, None)
    with pytest.raises(RuntimeError, match="null pointer"):
        eg._lib.lib().call("eg_conv_fwd", ctypes.byref(eg.ops.make_conv(2, 16, 16, 32, 32, 4, 2, 1)), 0, None, None, None, None, None)


def test_geometry_helpers_match_conv_arithmetic():
    eg = _pkg()
    c = eg.ops.make_conv(4, 32, 32, 128, 256, 4, 2, 1)
    assert eg.ops.pack_fwd_elems(c, 1) == 256 * 16 * 128
    assert eg.ops.pack_bwd_elems(c, 1) == 4 * 128 * 4 * 256            # 4 sub-pixel phases of 2x2 taps
    assert eg.ops.conv_wgrad_ws_bytes(c, 1) % (256 * 16 * 128 * 4) == 0
    c3 = eg.ops.make_conv(4, 32, 32, 16, 32, 3, 2, 1)                   # 3x3 stride 2: phases have 1,2,2,4 taps
    assert eg.ops.pack_bwd_elems(c3, 0) == 16 * (32 + 64 + 64 + 128)


def test_product_path_refuses_cpu_tensors():
    eg = _pkg()
    G = eg.celeba.Generator()
    assert list(G.state_dict().keys())[0] == "conv_blocks.0.weight"
    with pytest.raises(RuntimeError, match="MI355X only"):
        G(torch.zeros(2, 200), torch.zeros(2, 10), torch.zeros(2, 8))
    with pytest.raises(RuntimeError, match="MI355X only"):
        eg.celeba.get_matrix(torch.zeros(2, 5))


def test_no_oracle_import_in_product_path():
    for fn in os.listdir(os.path.join(ROOT, "ead-gan_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "ead-gan_amd", fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


DP_WORKER = r"""
import importlib, os, sys, torch
sys.path.insert(0, sys.argv[1])
eg = importlib.import_module("ead-gan_amd")
rank, world, local = eg.dp.init_from_env(backend="gloo")
ar = eg.dp.GradAllReduce(world)
g = torch.arange(10, dtype=torch.float32) * (rank + 1)
ar(g)
want = torch.arange(10, dtype=torch.float32) * sum(r + 1 for r in range(world)) / world
assert torch.allclose(g, want), (g, want)
g2 = torch.ones(6) * (rank + 1)
h = ar.start(g2)            # asynchronous form used to overlap the all-reduce with the generator backward
ar.finish(h)
assert torch.allclose(g2, torch.ones(6) * 1.5), g2
bw = eg.dp.GradAllReduce(world, wire="bf16")      # optional bf16 wire: mean of the bf16-rounded values, widened back to fp32
g3 = torch.full((300,), 1.0 + 2.0 ** -10) * (rank + 1)
keep = g3.clone()
bw(g3)
want3 = sum(float((keep / (rank + 1) * (r + 1)).bfloat16().float()[0]) for r in range(world)) / world
assert g3.dtype == torch.float32 and torch.allclose(g3, torch.full((300,), want3), rtol=2 ** -8), (g3[:3], want3)
h = bw.start(g3)
bw.finish(h)
# bucketed form of the CelebA trainer: slices of one arena, each started as its layers complete, all finished at the end
arena = torch.arange(40, dtype=torch.float32) * (rank + 1)
hs = [ar.start(arena[lo:hi]) for lo, hi in ((24, 40), (8, 24), (0, 8))]
for h in hs:
    ar.finish(h)
assert torch.allclose(arena, torch.arange(40, dtype=torch.float32) * 1.5), arena
t = eg.dp.max_over_ranks(float(rank + 1), torch.device("cpu"))
assert t == float(world), t
eg.dp.barrier()
print("rank", rank, "ok")
"""


def test_data_parallel_gradient_average_two_gloo_ranks(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_bench_decides_graph_or_eager_from_the_probe_child(monkeypatch):
    """bench.decide_graph: the capture is probed in a child process before the measuring process touches the GPU; a probe that exits
    non-zero (engine.exit_after_capture_failure: code 3; a runtime crash: a negative code) selects eager launches and says so."""
    import argparse
    import bench
    calls = []

    class R:
        returncode = 0

    def fake_run(cmd, **kw):
        calls.append((cmd, kw.get("env", {}).get("MASTER_PORT")))
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3"])
    a = argparse.Namespace(no_graph=False, graph_dist=False, no_probe=False, probe_capture=False)
    assert bench.decide_graph(a, 1) == (True, "")
    assert calls and "--probe-capture" in calls[0][0] and "--steps" in calls[0][0] and calls[0][1] == "29501"    # the children rendezvous on their own port
    R.returncode = 3
    use, note = bench.decide_graph(a, 1)
    assert use is False and "rc 3" in note
    R.returncode = -11
    assert bench.decide_graph(a, 1)[0] is False
    n = len(calls)
    assert bench.decide_graph(a, 2) == (False, "")                                 # N > 1 defaults to eager: no probe
    assert bench.decide_graph(argparse.Namespace(no_graph=True, graph_dist=False, no_probe=False, probe_capture=False), 1) == (False, "")
    assert bench.decide_graph(argparse.Namespace(no_graph=False, graph_dist=False, no_probe=True, probe_capture=False), 1) == (True, "")
    assert bench.decide_graph(argparse.Namespace(no_graph=False, graph_dist=False, no_probe=False, probe_capture=True), 1) == (True, "")   # the child itself
    assert len(calls) == n
