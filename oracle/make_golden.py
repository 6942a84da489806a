"""Generate tests/golden/*.npz from the reference itself (container only; needs /root/reference).

    python oracle/make_golden.py [celeba] [affine] ...

Each fixture holds inputs (or the seeds that regenerate them) and the reference's outputs; no reference
source travels.  Fixtures are committed; this script is committed so they can be regenerated.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh            # noqa: E402
from oracle import celeba_oracle as co          # noqa: E402
from oracle import mnist_oracle as mo           # noqa: E402
from oracle import dsprites_oracle as do        # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def probe_state(prefix, sd, out):
    """Small fingerprints of every tensor: first 8 values, sum, abs-sum (float64)."""
    for k, v in sd.items():
        t = v.detach().double().flatten()
        out[f"{prefix}/{k}/head"] = t[:8].numpy()
        out[f"{prefix}/{k}/sum"] = np.array(t.sum().item())
        out[f"{prefix}/{k}/abs"] = np.array(t.abs().sum().item())


def probe_grads(prefix, module, out):
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        t = p.grad.detach().double().flatten()
        out[f"{prefix}/{k}/head"] = t[:8].numpy()
        out[f"{prefix}/{k}/sum"] = np.array(t.sum().item())
        out[f"{prefix}/{k}/abs"] = np.array(t.abs().sum().item())


def make_celeba(B=4, steps=3, seed=0):
    """Losses of `steps` iterations + state/gradient fingerprints after the FIRST iteration.

    Post-Adam parameters amplify rounding noise wherever a gradient is ~0 (e.g. conv biases in front of
    a BatchNorm), so multi-step states are only comparable loosely; the 1-step gradients left in
    ``.grad`` by the info step are smooth and pin the backward pass tightly."""
    torch.set_num_threads(8)
    real = co.synthetic_real(B * steps, seed=1234).view(steps, B, 3, 64, 64)
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "real_seed": np.array(1234)}
    names = ("d_loss", "g_loss", "info_loss")
    for n in (1, steps):
        batches = [(real[i].clone(), torch.zeros(B, dtype=torch.int64)) for i in range(n)]
        g, recs = rh.run_script_loop("celebA/EAD-GAN_celebA.py", rh.celeba_opt(B), batches, names, seed)
        if n == 1:
            probe_state("G1", g["generator"].state_dict(), out)
            probe_state("D1", g["discriminator"].state_dict(), out)
            probe_grads("gG1", g["generator"], out)
            probe_grads("gD1", g["discriminator"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"celeba_b{B}_s{steps}.npz"), **out)
    print("celeba golden:", {k: out[k] for k in names})


def make_celeba_affine(B=16, seed=3):
    """Function-level vectors for utils_rpqxy.get_matrix / affine_regularzier and transformation_2D."""
    names = ("from_latent_vector_2_affine_para", "from_affine_para_2_latent_vector", "get_matrix", "affine_regularzier")
    g = rh.load_defs("celebA/utils_rpqxy.py", names)
    gm = rh.load_defs("celebA/EAD-GAN_celebA.py", ("transformation_2D",), opt=rh.celeba_opt(B))
    rng = np.random.RandomState(seed)
    code = torch.tensor(rng.uniform(-1, 1, (B, 8)), dtype=torch.float32)
    real_code = torch.tensor(rng.uniform(-1, 1, (B, 8)), dtype=torch.float32, requires_grad=True)
    trans_code = torch.tensor(rng.uniform(-1, 1, (B, 8)), dtype=torch.float32, requires_grad=True)
    img = co.synthetic_real(4, seed=77)
    with rh._cpu_only_patches():
        A = g["get_matrix"](code[:, :5])
        warped = gm["transformation_2D"]()(img, A[:4, 0:2])
        pred = g["affine_regularzier"](real_code, trans_code)
        w = torch.tensor(rng.normal(0, 1, (B, 5)), dtype=torch.float32)
        (pred * w).sum().backward()
    np.savez_compressed(os.path.join(GOLD, "celeba_affine.npz"), code=code.numpy(), A=A.detach().numpy(),
                        img_seed=np.array(77), warped=warped.detach().numpy(),
                        real_code=real_code.detach().numpy(), trans_code=trans_code.detach().numpy(),
                        pred=pred.detach().numpy(), w=w.numpy(), d_real=real_code.grad.numpy(),
                        d_trans=trans_code.grad.numpy())
    print("celeba affine golden written")


def make_mnist(B=8, steps=3, seed=0, mlp_seed=123):
    """MNIST loop (MNIST/EAD-GAN_rpqmnxy.py:338-446) on synthetic 32x32 batches; the frozen approximator the script loads at
    import (utils_rpqmnxy.py:36-43) is a seeded stand-in written to the temp cwd."""
    torch.set_num_threads(8)
    real = mo.synthetic_real(B * steps, seed=4321).view(steps, B, 1, 32, 32)
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "real_seed": np.array(4321), "mlp_seed": np.array(mlp_seed)}
    names = ("d_loss", "g_loss", "info_loss")
    mlp = mo.make_approximator(mlp_seed)

    real_save = torch.save          # torch.save is patched to a no-op inside the harness: keep the real one for the prerequisite

    def prereq2(tmp):
        real_save(mlp, os.path.join(tmp, "rpqmnxy_approximator.pt"))

    for n in (1, steps):
        batches = [(real[i].clone(), torch.zeros(B, dtype=torch.int64)) for i in range(n)]
        g, recs = rh.run_script_loop("MNIST/EAD-GAN_rpqmnxy.py", rh.mnist_opt(B), batches, names, seed, prereq=prereq2)
        if n == 1:
            probe_state("G1", g["generator"].state_dict(), out)
            probe_state("D1", g["discriminator"].state_dict(), out)
            probe_state("E1", g["encoder"].state_dict(), out)
            probe_grads("gG1", g["generator"], out)
            probe_grads("gE1", g["encoder"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"mnist_b{B}_s{steps}.npz"), **out)
    print("mnist golden:", {k: out[k] for k in names})


def make_mnist_affine(B=16, seed=5, mlp_seed=123):
    names = ("Affine_classifier", "from_latent_vector_2_affine_para", "from_affine_para_2_latent_vector", "get_matrix", "affine_regularizer")
    mlp = mo.make_approximator(mlp_seed)
    g = rh.load_defs("MNIST/utils_rpqmnxy.py", names)
    with rh._cpu_only_patches():
        net = g["Affine_classifier"]()
        net.load_state_dict(mlp)
        net.eval()
        g["BFGS_approximator"] = net
        rng = np.random.RandomState(seed)
        code = torch.tensor(rng.uniform(-1, 1, (B, 7)), dtype=torch.float32)
        real_code = torch.tensor(rng.uniform(-1, 1, (B, 7)), dtype=torch.float32, requires_grad=True)
        trans_code = torch.tensor(rng.uniform(-1, 1, (B, 7)), dtype=torch.float32, requires_grad=True)
        A = g["get_matrix"](code)
        pred = g["affine_regularizer"](real_code, trans_code)
        w = torch.tensor(rng.normal(0, 1, (B, 7)), dtype=torch.float32)
        (pred * w).sum().backward()
    np.savez_compressed(os.path.join(GOLD, "mnist_affine.npz"), code=code.numpy(), A=A.detach().numpy(), real_code=real_code.detach().numpy(),
                        trans_code=trans_code.detach().numpy(), pred=pred.detach().numpy(), w=w.numpy(), d_real=real_code.grad.numpy(),
                        d_trans=trans_code.grad.numpy(), mlp_seed=np.array(mlp_seed))
    print("mnist affine golden written")


def make_dsprites(B=8, steps=3, seed=0, pxy_seed=321):
    """dSprites/rp.py loop (:365-482) on synthetic uint8 sprites; encoder_pxy_50000.pt is a seeded stand-in in the temp cwd."""
    torch.set_num_threads(8)
    sprites = do.synthetic_sprites(B * steps, seed=99).view(steps, B, 64, 64)
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "sprite_seed": np.array(99), "pxy_seed": np.array(pxy_seed)}
    names = ("d_loss", "g_loss", "info_loss", "affine_loss", "relative_cat_loss")
    pxy = do.make_encoder_pxy(pxy_seed)
    real_save = torch.save

    def prereq(tmp):
        real_save(pxy, os.path.join(tmp, "encoder_pxy_50000.pt"))

    for n in (1, steps):
        batches = [sprites[i].clone() for i in range(n)]
        g, recs = rh.run_script_loop("dSprites/rp.py", rh.dsprites_opt(B), batches, names, seed, prereq=prereq)
        if n == 1:
            probe_state("G1", g["generator"].state_dict(), out)
            probe_state("D1", g["discriminator"].state_dict(), out)
            probe_state("E1", g["encoder"].state_dict(), out)
            probe_grads("gG1", g["generator"], out)
            probe_grads("gE1", g["encoder"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"dsprites_b{B}_s{steps}.npz"), **out)
    print("dsprites golden:", {k: out[k] for k in names})


def make_colored(B=8, steps=3, seed=0, pxy_seed=654):
    """colored_dSprites/rp_color.py loop (:365-516); encoder_pxy_color_50000.pt is a seeded stand-in."""
    torch.set_num_threads(8)
    sprites = do.synthetic_sprites(B * steps, seed=99).view(steps, B, 64, 64)
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "sprite_seed": np.array(99), "pxy_seed": np.array(pxy_seed)}
    names = ("d_loss", "g_loss", "cat_loss", "cont_loss", "affine_color_loss", "relative_cat_loss")
    pxy = do.make_encoder_pxy(pxy_seed, ch=3, pxy_out=6)
    real_save = torch.save

    def prereq(tmp):
        real_save(pxy, os.path.join(tmp, "encoder_pxy_color_50000.pt"))

    for n in (1, steps):
        batches = [sprites[i].clone() for i in range(n)]
        g, recs = rh.run_script_loop("colored_dSprites/rp_color.py", rh.colored_opt(B), batches, names, seed, prereq=prereq)
        if n == 1:
            probe_state("G1", g["generator"].state_dict(), out)
            probe_state("E1", g["encoder"].state_dict(), out)
            probe_grads("gG1", g["generator"], out)
            probe_grads("gE1", g["encoder"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"colored_b{B}_s{steps}.npz"), **out)
    print("colored golden:", {k: out[k] for k in names})


def make_celeba_curve(B=4, steps=120, seed=0):
    """Loss curves of a longer reference run (celebA/EAD-GAN_celebA.py:297-401, B=4, 120 iterations on seeded synthetic batches).
    Free-running trajectories of two fp32 implementations separate step by step (Adam's +-lr first updates amplify rounding
    noise), so these are compared as windowed means, not per step: the fixture pins the training DYNAMICS (optimizer
    schedules, BatchNorm / spectral-norm state evolution) that single-step vectors cannot."""
    torch.set_num_threads(8)
    real = co.synthetic_real(B * steps, seed=4242).view(steps, B, 3, 64, 64)
    names = ("d_loss", "g_loss", "info_loss")
    batches = [(real[i].clone(), torch.zeros(B, dtype=torch.int64)) for i in range(steps)]
    g, recs = rh.run_script_loop("celebA/EAD-GAN_celebA.py", rh.celeba_opt(B), batches, names, seed)
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "real_seed": np.array(4242)}
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, f"celeba_curve_b{B}_s{steps}.npz"), **out)
    print("celeba curve golden: first/last", {k: (float(out[k][0]), float(out[k][-1])) for k in names})


def make_pxy(B=8, steps=3, seed=0):
    """dSprites/pxy.py loop (:156-191, stage-1 trainer of Encoder_pxy) on synthetic uint8 sprites."""
    torch.set_num_threads(8)
    sprites = do.synthetic_sprites(B * steps, seed=98).view(steps, B, 64, 64)
    opt = rh.dsprites_opt(B)
    opt.lr, opt.code_dim = 0.0002, 3                 # argparse defaults pxy.py:37,42
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "sprite_seed": np.array(98)}
    names = ("affine_loss",)
    for n in (1, steps):
        batches = [sprites[i].clone() for i in range(n)]
        g, recs = rh.run_script_loop("dSprites/pxy.py", opt, batches, names, seed)
        if n == 1:
            probe_state("P1", g["encoder_pxy"].state_dict(), out)
            probe_grads("gP1", g["encoder_pxy"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"pxy_b{B}_s{steps}.npz"), **out)
    print("pxy golden:", {k: out[k] for k in names})


def make_pxy_color(B=8, steps=3, seed=0):
    """colored_dSprites/pxy_color.py loop (:160-216, stage-1 trainer of the colored Encoder_pxy) on synthetic uint8 sprites."""
    torch.set_num_threads(8)
    sprites = do.synthetic_sprites(B * steps, seed=97).view(steps, B, 64, 64)
    opt = rh.colored_opt(B)
    opt.lr, opt.code_dim = 0.0002, 6                 # argparse defaults pxy_color.py:34,39
    out = {"B": np.array(B), "steps": np.array(steps), "seed": np.array(seed), "sprite_seed": np.array(97)}
    names = ("affine_loss",)
    for n in (1, steps):
        batches = [sprites[i].clone() for i in range(n)]
        g, recs = rh.run_script_loop("colored_dSprites/pxy_color.py", opt, batches, names, seed)
        if n == 1:
            probe_state("P1", g["encoder_pxy"].state_dict(), out)
            probe_grads("gP1", g["encoder_pxy"], out)
    for k in names:
        out[k] = np.array([r[k] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"pxy_color_b{B}_s{steps}.npz"), **out)
    print("pxy_color golden:", {k: out[k] for k in names})


def make_approximator_fit(steps=5, seed=0):
    """MNIST/approximate_rpqmnxy.py __main__ (:109-153): losses of the first iterations + 1-step gradient/state fingerprints."""
    torch.set_num_threads(8)
    out = {"steps": np.array(steps), "seed": np.array(seed), "B": np.array(128)}
    g1, _ = rh.run_approximator_main(1, seed)
    probe_state("M1", g1["affine_classifier"].state_dict(), out)
    probe_grads("gM1", g1["affine_classifier"], out)
    _, losses = rh.run_approximator_main(steps, seed)
    out["affine_loss"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, f"approximator_fit_s{steps}.npz"), **out)
    print("approximator fit golden:", out["affine_loss"])


SAMPLE_SCRIPTS = {"mnist_train": ("MNIST/EAD-GAN_rpqmnxy.py", "mnist", (1, 32, 32), "train"),
                  "mnist_tool": ("MNIST/generate_image.py", "mnist", (1, 32, 32), "tool2"),
                  "celeba_train": ("celebA/EAD-GAN_celebA.py", "celeba", (3, 64, 64), "train"),
                  "celeba_tool": ("celebA/gen_imgs.py", "celeba", (3, 64, 64), "tool0"),
                  "dsprites_train": ("dSprites/rp.py", "dsprites", (1, 64, 64), "train"),
                  "colored_train": ("colored_dSprites/rp_color.py", "colored", (3, 64, 64), "train")}


def make_sample_plans(n=10):
    """What each script's own sample_image feeds its generator and writer: the functions are run with recording stand-ins
    (np.random seeded 0 for the static-sample z draw); stored per kind: every generator input and the (path, nrow, normalize, gridded) list."""
    import json
    out = {"n": np.array(n)}
    for kind, (path, optname, shape, call) in SAMPLE_SCRIPTS.items():
        opt = getattr(rh, optname + "_opt")(16)
        real = torch.zeros(n * n, *shape)
        args = {"train": (real, real, n, 0), "tool2": (n, 0), "tool0": ()}[call]
        calls, saves = rh.record_sample_image(path, opt, args, shape)
        for i, c in enumerate(calls):
            for j, t in enumerate(c):
                out[f"{kind}/call{i}/in{j}"] = t.numpy()
        out[f"{kind}/saves"] = np.array(json.dumps([[s[0], s[1], s[2], s[3].startswith("grid")] for s in saves]))
        print("sample plan", kind, len(calls), "generator calls,", len(saves), "files")
    np.savez_compressed(os.path.join(GOLD, "sample_plans.npz"), **out)


MAKERS = {"colored": make_colored, "dsprites": make_dsprites, "celeba": make_celeba, "celeba_affine": make_celeba_affine, "mnist": make_mnist, "mnist_affine": make_mnist_affine,
          "celeba_curve": make_celeba_curve, "pxy": make_pxy, "pxy_color": make_pxy_color, "approximator_fit": make_approximator_fit, "sample_plans": make_sample_plans}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or list(MAKERS)
    for w in which:
        MAKERS[w]()
