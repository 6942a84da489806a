"""Sampling tools on the MI355X: the grid / normalise / quantise kernels against the CPU restatement of the torchvision
writers (oracle/sampling_oracle.py -- bit-exact bytes), and sample_image of the six scripts end to end with the eval- and train-mode
generators (inputs pinned to the reference by tests/golden/sample_plans.npz in the CPU suite)."""
import importlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import sampling_oracle as so

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")
    torch.set_num_threads(16)


def _batch(shape, seed, lo=-1.0, hi=1.0):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * (hi - lo) + lo


@pytest.mark.parametrize("shape,nrow", [((100, 1, 32, 32), 10), ((100, 3, 64, 64), 10), ((70, 1, 64, 64), 10), ((70, 3, 64, 64), 10),
                                        ((5, 1, 2, 3), 3), ((1, 3, 9, 7), 8), ((1, 1, 9, 7), 8), ((3, 2, 5, 5), 8), ((17, 3, 8, 8), 4)])
def test_make_grid_matches_oracle(shape, nrow):
    t = _batch(shape, 1, -3, 2)
    for padding, pad_value in ((2, 0.0), (0, 0.0), (3, -0.25)):
        want = so.make_grid(t, nrow, padding, pad_value)
        got = eg.sampling.make_grid(t.to(DEV), nrow, padding, pad_value=pad_value)
        assert got.shape == want.shape and torch.equal(got.cpu(), want)


def test_minmax():
    for n in (1, 63, 257, 100 * 3 * 64 * 64, 1 << 22):
        t = _batch((n,), n, -7, 5)
        out = eg.sampling._minmax(t.to(DEV)).cpu()
        assert float(out[0]) == float(t.min()) and float(out[1]) == float(t.max())


@pytest.mark.parametrize("shape", [(100, 1, 32, 32), (100, 3, 64, 64), (70, 1, 64, 64), (5, 1, 2, 3), (1, 3, 9, 7), (1, 1, 9, 7)])
@pytest.mark.parametrize("normalize", [False, True])
def test_save_image_bytes_match_oracle(shape, normalize):
    """both writer paths of the reference: save_image(batch, normalize) and make_grid(batch) -> save_image(grid, normalize=True)"""
    t = _batch(shape, 2, -1.0, 1.0) if normalize else _batch(shape, 2, -0.2, 1.2)
    want = so.to_uint8_hwc(t, 10, 2, normalize)
    got = eg.sampling.to_uint8_hwc(t.to(DEV), 10, 2, normalize)
    assert got.shape == want.shape and np.array_equal(got, want), int((got != want).sum())
    want = so.to_uint8_hwc(so.make_grid(t, 10), 10, 2, normalize)
    got = eg.sampling.to_uint8_hwc(eg.sampling.make_grid(t.to(DEV), 10), 10, 2, normalize)
    assert got.shape == want.shape and np.array_equal(got, want), int((got != want).sum())


def test_constant_image_and_cpu_tensor():
    t = torch.full((4, 1, 3, 3), 0.25)
    assert np.array_equal(eg.sampling.to_uint8_hwc(t.to(DEV), 2, 2, True), so.to_uint8_hwc(t, 2, 2, True))       # hi == lo: the 1e-5 guard
    with pytest.raises(RuntimeError):
        eg.sampling.make_grid(t)


def _generator(kind, dtype="f32"):
    torch.manual_seed(7)
    if kind.startswith("mnist"):
        return eg.mnist.Generator(dtype=dtype).to(DEV)
    if kind.startswith("celeba"):
        return eg.celeba.Generator(dtype=dtype).to(DEV)
    if kind.startswith("dsprites"):
        return eg.dsprites.Generator(code_dim=4, n_classes=3, channels=1, dtype=dtype).to(DEV)
    return eg.colored.Generator(dtype=dtype).to(DEV)


@pytest.mark.parametrize("kind", so.KINDS)
def test_sample_image_end_to_end(kind, tmp_path):
    """every PNG the script's sample_image writes == the oracle's bytes for the same generator outputs (tools: eval mode; training
    scripts: train mode, as the reference samples without .eval())"""
    PIL = pytest.importorskip("PIL.Image")
    n = 10
    G = _generator(kind)
    if kind.endswith("tool"):
        G.eval()
    shape = {"mnist": (1, 32, 32), "celeba": (3, 64, 64), "dsprites": (1, 64, 64), "colored": (3, 64, 64)}[kind.split("_")[0]]
    sprite = kind.startswith(("dsprites", "colored"))
    real = _batch((100,) + shape, 3, 0, 1) if sprite else _batch((100,) + shape, 3)
    trans = _batch((100,) + shape, 4, 0, 1) if sprite else _batch((100,) + shape, 4)
    paths = eg.sampling.sample_image(kind, G, n, 40, real.to(DEV), trans.to(DEV), out_dir=str(tmp_path), rng=np.random.RandomState(0))
    plan = so.sample_plan(kind, n, np.random.RandomState(0))
    assert [os.path.relpath(p, tmp_path) for p in paths] == [os.path.join(p["path"], "40.png") for p in plan]
    given = [real, trans]
    for fp, p in zip(paths, plan):
        with torch.no_grad():
            img = given.pop(0) if p["inputs"] is None else G(*[i.to(DEV) for i in p["inputs"]]).float().cpu()
        if p["sprite"]:
            img = (img - 0.5) * 2
        want = so.to_uint8_hwc(so.make_grid(img, n) if p["gridded"] else img, n, 2, True)
        got = np.asarray(PIL.open(fp))
        assert got.shape == want.shape and np.array_equal(got, want), (fp, int((got != want).sum()))
    assert len(given) == (0 if not kind.endswith("tool") else 2)


def test_traversal_moves_the_image():
    """an eval-mode CelebA generator: rows of a varying_c grid differ from each other (the code reaches the output), static inputs repeat"""
    G = _generator("celeba_tool").eval()
    plan = so.sample_plan("celeba_tool", 10)
    with torch.no_grad():
        a = G(*[i.to(DEV) for i in plan[0]["inputs"]]).float().cpu()
        b = G(*[i.to(DEV) for i in plan[0]["inputs"]]).float().cpu()
    assert torch.equal(a, b)
    assert float((a[0] - a[99]).abs().max()) > 1e-4


@pytest.mark.parametrize("kind,fmt", [("mnist_tool", "pt"), ("celeba_tool", "tar")])
def test_run_tool_from_checkpoint(kind, fmt, tmp_path):
    """generate_image.py / gen_imgs.py: train a little so BN running statistics are non-trivial, write the checkpoint in the script's
    format, load it into a fresh eval-mode generator and sample -- same PNG bytes as sampling the source generator in eval mode"""
    PIL = pytest.importorskip("PIL.Image")
    G = _generator(kind)
    with torch.no_grad():
        for s in range(3):                                   # train-mode forwards move the running statistics
            plan = so.sample_plan(kind, 10)
            G(*[(i + 0.1 * _batch(i.shape, s)).to(DEV) for i in plan[0]["inputs"]])
    if fmt == "pt":
        path = str(tmp_path / "generator_40000.pt")
        torch.save(G.state_dict(), path)                     # MNIST/EAD-GAN_rpqmnxy.py:465
    else:
        path = str(tmp_path / "checkpoint_600000.tar")
        D = eg.celeba.Discriminator().to(DEV)
        eg.sampling.save_checkpoint(path, G, D, epoch=3, batches_done=600000)
        ck = torch.load(path, map_location="cpu", weights_only=True)
        assert set(ck) == {"discriminator_state_dict", "generator_state_dict", "epoch", "batches_done"} and ck["batches_done"] == 600000
    a = eg.sampling.run_tool(kind, path, out_dir=str(tmp_path / "a"))
    b = eg.sampling.sample_image(kind, G.eval(), 10, 0, out_dir=str(tmp_path / "b"))
    assert len(a) == len(b) == (7 if kind == "mnist_tool" else 8)
    for fa, fb in zip(a, b):
        assert np.array_equal(np.asarray(PIL.open(fa)), np.asarray(PIL.open(fb)))
    assert len({open(f, "rb").read() for f in a}) > 1        # the grids differ from each other
