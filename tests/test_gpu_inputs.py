"""Device-side input pipeline: counter-based draws with the distributions of the reference's numpy calls
(celebA/EAD-GAN_celebA.py:308-317) and the DataLoader transforms (:194-206) as one gather kernel over a uint8 dataset in HBM."""
import importlib

import numpy as np
import pytest
import torch

from oracle import celeba_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")


def _fill(kind, n, a, b, seed, step, stream, dtype=torch.float32):
    ops = eg.ops
    out = torch.empty(n, device=DEV, dtype=dtype)
    st = torch.tensor([step], device=DEV, dtype=torch.int32)
    ops.rng_fill(kind, out, a, b, seed, st, stream)
    torch.cuda.synchronize()
    return out.cpu()


def test_rng_distributions_and_reproducibility():
    ops = eg.ops
    n = 1 << 20
    u = _fill(ops.RNG_UNIFORM, n, -1.0, 1.0, 7, 0, 4).double()
    assert u.min() >= -1.0 and u.max() < 1.0
    assert abs(u.mean()) < 3e-3 and abs(u.var() - 1.0 / 3.0) < 3e-3
    g = _fill(ops.RNG_NORMAL, n, 0.0, 1.0, 7, 0, 3).double()
    assert abs(g.mean()) < 4e-3 and abs(g.var() - 1.0) < 6e-3
    assert abs(((g - g.mean()) ** 4).mean() / g.var() ** 2 - 3.0) < 0.05          # kurtosis of a normal
    assert abs((g.abs() > 1.96).double().mean() - 0.05) < 2e-3                     # tails
    k = _fill(ops.RNG_RANDINT, n, 0, 10, 7, 0, 5, torch.int64)
    assert k.min() == 0 and k.max() == 9
    assert (torch.bincount(k, minlength=10).double() / n - 0.1).abs().max() < 2e-3
    f = _fill(ops.RNG_BERNOULLI, n, 0.5, 0.0, 7, 0, 2, torch.uint8)
    assert abs(f.double().mean() - 0.5) < 2e-3
    # same (seed, step, stream) -> same draws; any of the three changed -> fresh, uncorrelated draws
    assert torch.equal(_fill(ops.RNG_NORMAL, 4099, 0.0, 1.0, 7, 5, 3), _fill(ops.RNG_NORMAL, 4099, 0.0, 1.0, 7, 5, 3))
    base = _fill(ops.RNG_NORMAL, n, 0.0, 1.0, 7, 5, 3).double()
    for other in (_fill(ops.RNG_NORMAL, n, 0.0, 1.0, 8, 5, 3), _fill(ops.RNG_NORMAL, n, 0.0, 1.0, 7, 6, 3), _fill(ops.RNG_NORMAL, n, 0.0, 1.0, 7, 5, 4)):
        assert abs(float((base * other.double()).mean())) < 5e-3
    # sequential correlation inside one stream
    assert abs(float((base[:-1] * base[1:]).mean())) < 5e-3 and abs(float((base[:-4] * base[4:]).mean())) < 5e-3


@pytest.mark.parametrize("N,B", [(1000, 8), (1003, 8), (37, 128), (202599, 128)])
def test_epoch_permutation_visits_every_image_once_per_epoch(N, B):
    """DataLoader(shuffle=True) (celebA/EAD-GAN_celebA.py:204-206): RNG_EPOCH_PERM draws the dataset indices of stream positions
    step * B + i; every epoch (N consecutive positions) is a permutation of [0, N), epochs differ, batches straddle epoch ends."""
    ops = eg.ops
    steps = (2 * N + B - 1) // B + 1
    step = torch.zeros(1, device=DEV, dtype=torch.int32)
    out = torch.empty(B, device=DEV, dtype=torch.int64)
    seen = []
    for s in range(steps):
        ops.rng_fill(ops.RNG_EPOCH_PERM, out, N, 0, 123, step, 1)
        seen.append(out.clone())
        ops.counter_add(step, 1)
    idx = torch.cat(seen).cpu()
    e0, e1 = idx[:N], idx[N:2 * N]
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    assert torch.equal(torch.sort(e0).values, torch.arange(N)) and torch.equal(torch.sort(e1).values, torch.arange(N))
    if N > 100:
        assert not torch.equal(e0, e1) and float((e0 == torch.arange(N)).float().mean()) < 0.05     # shuffled, and differently per epoch
    # another seed / stream: another order
    ops.rng_fill(ops.RNG_EPOCH_PERM, out, N, 0, 124, torch.zeros(1, device=DEV, dtype=torch.int32), 1)
    if N > 1000:
        assert not torch.equal(out.cpu(), seen[0].cpu())


def test_gather_flip_normalize_and_onehot():
    ops = eg.ops
    g = torch.Generator().manual_seed(3)
    data = torch.randint(0, 256, (37, 3, 64, 64), generator=g, dtype=torch.uint8)
    idx = torch.randint(0, 37, (16,), generator=g)
    flip = torch.randint(0, 2, (16,), generator=g, dtype=torch.uint8)
    out = torch.empty(16, 3, 64, 64, device=DEV)
    ops.gather_u8_images(data.to(DEV), idx.to(DEV), flip.to(DEV), out, 16, 3, 64, 64, 2.0 / 255.0, -1.0)
    want = data[idx].float()
    want = torch.where(flip.view(-1, 1, 1, 1).bool(), want.flip(-1), want) / 255.0       # ToTensor
    want = (want - 0.5) / 0.5                                                             # Normalize((.5,.5,.5), (.5,.5,.5))
    torch.testing.assert_close(out.cpu(), want, rtol=0, atol=2e-7)
    oh = torch.empty(16, 10, device=DEV)
    lab = torch.randint(0, 10, (16,), generator=g)
    ops.onehot(lab.to(DEV), oh, 16, 10)
    assert torch.equal(oh.cpu(), torch.nn.functional.one_hot(lab, 10).float())


def test_trainer_feeds_itself_from_the_captured_graph():
    """capture(inputs=DeviceInputs(...)): every replay draws a fresh batch (dataset sampling + flip + normalise, z, code, labels) on the
    device and trains on it; eager and captured runs with the same seed see the same sequence of batches (identical losses)."""
    B = 8
    g = torch.Generator().manual_seed(11)
    data = (co.synthetic_real(64, seed=9) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).to(DEV)
    runs = []
    for capture in (False, True):
        orc = co.CelebAOracle(seed=2)
        G = eg.celeba.Generator(dtype="bf16").to(DEV)
        D = eg.celeba.Discriminator(dtype="bf16").to(DEV)
        G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
        D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
        inp = eg.celeba.DeviceInputs(data, seed=5)
        tr.inputs = inp
        out, batches = [], []
        for i in range(4):
            if capture and i == 1:
                tr.capture(inputs=inp)
            out.append(tr.step_resident().clone())
            batches.append((tr.real.clone(), tr.z.clone(), tr.labels.clone()))
        torch.cuda.synchronize()
        assert int(inp.step.item()) == 4
        assert not torch.equal(batches[0][1], batches[1][1]) and not torch.equal(batches[1][0], batches[2][0])
        assert float(tr.real.min()) >= -1.0 and float(tr.real.max()) <= 1.0
        runs.append((torch.stack(out).cpu(), [b[2].cpu() for b in batches]))
    assert torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))


def test_fused_iteration_head_draws_the_same_batches(monkeypatch):
    """The head of the CelebA iteration as 3 launches (all draws + one-hot in one, counter tick inside the gather, affine matrix + warp + loss
    clearing in one) against the 11 separate launches: identical real / z / code / labels / one-hot / theta / warped batch and losses."""
    B = 8
    data = (co.synthetic_real(64, seed=9) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).to(DEV)
    runs = []
    for fuse in (False, True):
        monkeypatch.setattr(eg.celeba, "FUSE_INPUTS", fuse)
        orc = co.CelebAOracle(seed=2)
        G = eg.celeba.Generator(dtype="bf16").to(DEV)
        D = eg.celeba.Discriminator(dtype="bf16").to(DEV)
        G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
        D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
        inp = eg.celeba.DeviceInputs(data, seed=5)
        tr.inputs = inp
        seen = []
        for i in range(3):
            loss = tr.step_resident().clone()
            seen.append([t.clone() for t in (tr.real, tr.z, tr.code, tr.labels, tr.onehot, tr.theta, tr.scaled, loss)])
        torch.cuda.synchronize()
        assert int(inp.step.item()) == 3
        runs.append(seen)
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def _run_small(family, capture, steps=4, B=8):
    """one of the small-network trainers feeding itself from a DeviceInputs; returns (losses per step, label draws, sampler, trainer)"""
    torch.manual_seed(0)
    if family == "mnist":
        from oracle import mnist_oracle as mo
        orc = mo.MnistOracle(seed=2, mlp=mo.make_approximator(123))
        eg.mnist.load_approximator(mo.make_approximator(123))
        mods = [eg.mnist.Generator(dtype="f32").to(DEV), eg.mnist.Discriminator(dtype="f32").to(DEV), eg.mnist.Encoder(dtype="f32").to(DEV)]
        for m, ref in zip(mods, (orc.G, orc.D, orc.E)):
            m.load_state_dict({k: v.detach() for k, v in ref.items()})
        tr = eg.mnist.MnistTrainer(*mods, B, dtype="f32")
        data = torch.randint(0, 256, (97, 1, 32, 32), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).to(DEV)
        inp = eg.mnist.DeviceInputs(data, seed=5)
    else:
        from oracle import dsprites_oracle as do
        color = family == "colored"
        mod = eg.colored if color else eg.dsprites
        orc = (do.ColoredOracle if color else do.DspritesOracle)(seed=2)
        mods = [mod.Encoder_pxy(dtype="f32").to(DEV), mod.Generator(dtype="f32").to(DEV), mod.Discriminator(dtype="f32").to(DEV), mod.Encoder(dtype="f32").to(DEV)]
        for m, ref in zip(mods, (orc.P, orc.G, orc.D, orc.E)):
            m.load_state_dict({k: v.detach() for k, v in ref.items()})
        tr = (mod.ColoredTrainer if color else mod.DspritesTrainer)(*mods, B, dtype="f32")
        inp = mod.DeviceInputs(do.synthetic_sprites(61, seed=4).to(DEV), seed=5)
    tr.inputs = inp
    out, draws = [], []
    for i in range(steps):
        if capture and i == 1:
            tr.capture(inputs=inp)
        out.append(tr.step_resident().clone())
        draws.append(((tr.code if family == "mnist" else tr.code1).clone(), (tr.real if family == "mnist" else tr.img).clone()))
    torch.cuda.synchronize()
    return torch.stack(out).cpu(), draws, inp, tr


@pytest.mark.parametrize("family", ["mnist", "dsprites", "colored"])
def test_small_network_trainers_feed_themselves(family):
    """SURVEY 8 row f1 for MNIST (rpqmnxy.py:233-246,351-357), dSprites (rp.py:236-262,389-430) and colored dSprites (rp_color.py:363-381):
    dataset sampling, transforms and the per-step draws run on the device inside the (captured) iteration; eager and captured runs of one
    seed see the same batches; the draws have the reference's ranges and change every step."""
    eager, d0, inp0, tr0 = _run_small(family, False)
    graph, d1, inp1, tr1 = _run_small(family, True)
    assert int(inp0.step.item()) == 4 and int(inp1.step.item()) == 4
    assert torch.isfinite(eager).all() and torch.equal(eager, graph)
    for (c0, x0), (c1, x1) in zip(d0, d1):
        assert torch.equal(c0, c1) and torch.equal(x0, x1)
    codes = torch.stack([c for c, _ in d0])
    assert float(codes.min()) >= -1.0 and float(codes.max()) < 1.0 and not torch.equal(codes[0], codes[1])
    img = d0[-1][1]
    if family == "mnist":
        assert img.shape == (8, 1, 32, 32) and float(img.min()) >= -1.0 and float(img.max()) <= 1.0 + 2e-7           # ToTensor + Normalize(.5,.5)
    elif family == "dsprites":
        assert img.shape == (8, 1, 64, 64) and set(img.unique().tolist()) <= {0.0, 1.0}
    else:
        gains = tr0.gains
        assert img.shape == (8, 3, 64, 64) and float(gains.min()) >= 0.5 and float(gains.max()) < 1.0
        on = img.amax(dim=(2, 3))                                                                              # sprite pixels carry the gain
        assert torch.allclose(on, gains, atol=1e-6)


@pytest.mark.parametrize("family", ["mnist", "dsprites", "colored"])
def test_small_network_draws_in_one_launch_are_the_same_draws(family, monkeypatch):
    """the iteration's draws as ONE launch (DeviceSampler.begin_draws / end_draws -> eg_rng_fill_multi, one-hot rows fused, MNIST's counter
    tick inside the gather) against one launch per draw: the same losses, codes and images, step for step"""
    runs = []
    for fuse in (False, True):
        monkeypatch.setattr(eg.engine, "FUSE_DRAWS", fuse)
        monkeypatch.setattr(eg.mnist, "FUSE_DRAWS", fuse)
        monkeypatch.setattr(eg.dsprites, "FUSE_DRAWS", fuse)
        runs.append(_run_small(family, False))
    (l0, d0, i0, _), (l1, d1, i1, _) = runs
    assert int(i0.step.item()) == 4 and int(i1.step.item()) == 4
    assert torch.equal(l0, l1)
    for (c0, x0), (c1, x1) in zip(d0, d1):
        assert torch.equal(c0, c1) and torch.equal(x0, x1)


@pytest.mark.parametrize("hw", [(218, 178), (178, 218), (100, 64), (70, 200), (64, 64)])
def test_resize_center_crop_matches_pil(hw):
    """f1 remainder: transforms.Resize(64) + CenterCrop(64) (celebA/EAD-GAN_celebA.py:194-196) on the device, bit for bit against PIL
    (CelebA's 218 x 178 portraits, a landscape, one edge already 64, no resize at all)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_dropin_host import _pil_resize_center_crop
    H, W = hw
    rng = np.random.RandomState(7)
    imgs = rng.randint(0, 256, (5, H, W, 3), dtype=np.uint8)
    got = eg.celeba.resize_center_crop_u8(torch.from_numpy(imgs).permute(0, 3, 1, 2).contiguous().to(DEV), 64)
    assert got.shape == (5, 3, 64, 64) and got.dtype == torch.uint8
    want = np.stack([_pil_resize_center_crop(im, 64) for im in imgs]).transpose(0, 3, 1, 2)
    assert np.array_equal(got.cpu().numpy(), want)
