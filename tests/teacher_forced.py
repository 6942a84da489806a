"""Teacher-forced loss curves (BASELINE.json north star: `loss curves within 1e-3 of reference for 1k steps') for the
MNIST, dSprites and colored-dSprites loops -- the CelebA one lives in tests/test_gpu_celeba.py.  The oracle free-runs the reference loop
(MNIST/EAD-GAN_rpqmnxy.py:338-446, dSprites/rp.py:363-482, colored_dSprites/rp_color.py:363-516); before EVERY iteration the HIP side
is reset to the oracle's state -- parameters, BatchNorm running statistics, spectral-norm u / v, moments and step counts of the
Adams -- then both run that iteration on the same inputs.  Returns the per-iteration |loss difference|, [steps, number of losses]."""
import importlib

import numpy as np
import torch

DEV = "cuda"


def curve(family, steps, B=4, seed=11, progress=None):
    eg = importlib.import_module("ead-gan_amd")
    rng = np.random.RandomState(seed)
    if family == "mnist":
        from oracle import mnist_oracle as mo
        mlp = mo.make_approximator(123)
        orc = mo.MnistOracle(seed=seed, mlp=mlp)
        eg.mnist.load_approximator(mlp)
        mods = [eg.mnist.Generator(dtype="f32").to(DEV), eg.mnist.Discriminator(dtype="f32").to(DEV), eg.mnist.Encoder(dtype="f32").to(DEV)]
        refs = lambda: (orc.G, orc.D, orc.E)
        tr = eg.mnist.MnistTrainer(*mods, B, dtype="f32")
        names = ("g_loss", "d_loss", "info_loss")
        data = mo.synthetic_real(B * steps, seed=4321).view(steps, B, 1, 32, 32)
        draw = lambda: mo.draw_step_inputs(rng, B)
        opts = lambda: (orc.opt_G, orc.opt_D, orc.opt_info)
    else:
        from oracle import dsprites_oracle as do
        color = family == "colored"
        mod = eg.colored if color else eg.dsprites
        pxy = do.make_encoder_pxy(654, ch=3, pxy_out=6) if color else do.make_encoder_pxy(321)
        orc = (do.ColoredOracle if color else do.DspritesOracle)(seed=seed, pxy=pxy)
        mods = [mod.Encoder_pxy(dtype="f32").to(DEV), mod.Generator(dtype="f32").to(DEV), mod.Discriminator(dtype="f32").to(DEV), mod.Encoder(dtype="f32").to(DEV)]
        mods[0].load_state_dict({k: v.detach() for k, v in pxy.items()})
        refs = lambda: (None, orc.G, orc.D, orc.E)
        tr = (mod.ColoredTrainer if color else mod.DspritesTrainer)(*mods, B, dtype="f32")
        names = ("d_loss", "g_loss", "info_loss", "affine_loss", "relative_cat_loss")
        data = do.synthetic_sprites(B * steps, seed=99).view(steps, B, 64, 64)
        draw = (lambda: do.draw_colored_inputs(rng, B)) if color else (lambda: do.draw_step_inputs(rng, B))
        opts = lambda: (orc.opt_D, orc.opt_info)
    dev = np.zeros((steps, len(names)))
    for i in range(steps):
        for m, ref in zip(mods, refs()):
            if ref is not None:
                m.load_state_dict({k: v.detach() for k, v in ref.items()})
        if i:
            tr.import_adam_state(*opts())
        inp = draw()
        got = tr.train_step(data[i].to(DEV), *[t.to(DEV) for t in inp])
        want = orc.train_step(data[i], *inp)
        dev[i] = [abs(got[k] - want[k]) for k in names]
        if progress is not None:
            progress(i, dev)
    return dev, names
