"""Data-parallel CelebA train step on the real HIP path: two ranks share the one GPU of the test box and talk over gloo
(RCCL refuses two ranks on one device; the N-GPU RCCL path differs only in the backend of the same torch.distributed calls).
Checks, with all learning rates 0: the gradients every rank holds after the step are the mean of the per-shard gradients
(computed here without any collective), both ranks agree bit for bit, and the side-stream schedule (overlap=True, asynchronous
all-reduce of D's gradients beside the generator backward) gives the same numbers as the single-stream one."""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dp_worker  # noqa: E402

pytestmark = pytest.mark.gpu


def _spawn(world, outdir, B, overlap, port):
    mp.spawn(dp_worker.run_rank, args=(world, port, outdir, B, overlap), nprocs=world, join=True)


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_average_the_shard_gradients(overlap):
    B = 8
    with tempfile.TemporaryDirectory() as d:
        _spawn(2, d, B, overlap, 29533 + int(overlap))
        r0 = torch.load(os.path.join(d, "rank0_of2.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1_of2.pt"), weights_only=True)
        # the same two shards, each as a world-size-1 run (no collective anywhere)
        single = []
        for shard in (0, 1):
            mp.spawn(dp_worker.run_single_shard, args=(shard, 2, d, B), nprocs=1, join=True)
            single.append(torch.load(os.path.join(d, f"rank{shard}_of1.pt"), weights_only=True))
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["d"], r1["d"])          # replicas hold the same averaged gradients
    for key in ("g", "d"):
        want = (single[0][key] + single[1][key]) * 0.5
        err = float((r0[key] - want).norm() / want.norm())
        assert err < 1e-6, (key, err)
    # every rank reports the loss of ITS shard
    assert torch.allclose(r0["losses"], single[0]["losses"], atol=1e-6) and torch.allclose(r1["losses"], single[1]["losses"], atol=1e-6)


def test_sync_batchnorm_two_ranks_equal_one_rank_at_the_same_global_batch():
    """with synchronised BatchNorm (dp.SyncBN: 3*C floats per layer forward, 2*C backward) two ranks x 4 images
    reproduce one rank x 8 images: the mean of the rank losses is the global loss, the all-reduced gradients are the global-batch
    gradients, and the BatchNorm running statistics are those of the global batch.  (Gradient bound 2e-2 like the single-rank
    full-step test: a LeakyReLU unit within rounding of 0 may flip between the two summation orders.)"""
    B = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dp_worker.run_rank, args=(2, 29541, d, B, True, None, True), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0_of2.pt"), weights_only=True)
        r1 = torch.load(os.path.join(d, "rank1_of2.pt"), weights_only=True)
        one = os.path.join(d, "one")
        os.makedirs(one)
        mp.spawn(dp_worker.run_whole_batch, args=(one, B), nprocs=1, join=True)
        w = torch.load(os.path.join(one, "rank0_of1.pt"), weights_only=True)
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["d"], r1["d"])
    assert torch.allclose((r0["losses"] + r1["losses"]) * 0.5, w["losses"], atol=2e-5), (r0["losses"], r1["losses"], w["losses"])
    for key in ("g", "d"):
        err = float((r0[key] - w[key]).norm() / w[key].norm())
        assert err < 2e-2, (key, err)
    for key in ("rm", "rv"):
        assert torch.allclose(r0[key], w[key], rtol=1e-4, atol=1e-6) and torch.equal(r0[key], r1[key]), key


@pytest.mark.parametrize("family,port", [("mnist", 29551), ("dsprites", 29552)])
def test_small_network_generators_sync_batchnorm_two_ranks(family, port):
    """MNIST (MNIST/EAD-GAN_rpqmnxy.py:71-98) and dSprites (dSprites/rp.py:123-157) generators, three BatchNorm layers each: two ranks x 4
    images with dp.SyncBN give the images of one rank x 8 images; the rank gradient shares sum to the whole-batch gradient; the running
    statistics are those of the global batch on both ranks."""
    B = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dp_worker.run_generator_sync, args=(2, port, d, family, B), nprocs=2, join=True)
        mp.spawn(dp_worker.run_generator_sync, args=(1, port, d, family, B), nprocs=1, join=True)
        r0, r1, w = (torch.load(os.path.join(d, f"{family}_rank{r}_of{n}.pt"), weights_only=True) for r, n in ((0, 2), (1, 2), (0, 1)))
    img = torch.cat((r0["img"], r1["img"]))
    assert float((img - w["img"]).norm() / w["img"].norm()) < 1e-5
    gsum = r0["grad"] + r1["grad"]
    assert float((gsum - w["grad"]).norm() / w["grad"].norm()) < 2e-3           # LeakyReLU / ReLU units within rounding of 0 may flip
    assert torch.equal(r0["running"], r1["running"])
    assert torch.allclose(r0["running"], w["running"], rtol=1e-4, atol=1e-6)


def test_rccl_allreduce_inside_the_captured_step_child_process():
    """The data-parallel step with a real RCCL process group (one rank) captured into the hipGraph.  Run in a child process: the failure this
    guards against is a segfault inside hipStreamEndCapture (a side lane that RCCL's stream waited for and that later waits for RCCL's
    stream -- profiles/r01_timeline_notes.md), which no in-process assertion can catch."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 200))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--no-probe", "--no-cpu-baseline", "--no-roofline", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "hipGraph replay" in line["config"]["workload"], (line["config"], r.stderr[-2000:])     # captured, not the eager fallback
