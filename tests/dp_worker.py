"""Worker of tests/test_gpu_dp.py: one data-parallel rank of the CelebA train step on the (shared) GPU, gloo backend.
Kept free of pytest imports so that torch.multiprocessing can spawn it."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def shard_inputs(co, B_global, rank, world, seed=5):
    rng = np.random.RandomState(seed)
    z, code, labels = co.draw_step_inputs(rng, B_global)
    real = co.synthetic_real(B_global, seed=77)
    b = B_global // world
    sl = slice(rank * b, (rank + 1) * b)
    return real[sl], z[sl], code[sl], labels[sl]


def run_single_shard(_, shard, nshards, outdir, B_global):
    """one shard of the global batch as a world-size-1 run: no process group, no collective, single stream"""
    run_rank(shard, 1, 0, outdir, B_global, False, nshards=nshards)


def run_whole_batch(_, outdir, B_global):
    """the global batch on one rank (reference for synchronised BatchNorm)"""
    run_rank(0, 1, 0, outdir, B_global, False, nshards=1)


def run_rank(rank, world, port, outdir, B_global, overlap, nshards=None, sync_bn=False):
    from oracle import celeba_oracle as co          # seeded initial weights + synthetic batch (test infrastructure)
    eg = importlib.import_module("ead-gan_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if world > 1:
        eg.dp.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    orc = co.CelebAOracle(seed=3, lrs=(0.0, 0.0, 0.0))
    G = eg.celeba.Generator(dtype="f32").to("cuda")
    D = eg.celeba.Discriminator(dtype="f32").to("cuda")
    G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
    D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
    ar = eg.dp.GradAllReduce(world) if world > 1 else None
    nshards = nshards or world
    b = B_global // nshards
    sync = eg.dp.SyncBN(world, rank) if (sync_bn and world > 1) else None
    tr = eg.celeba.CelebATrainer(G, D, b, dtype="f32", allreduce=ar, lr_g=0.0, lr_d=0.0, lr_info=0.0, overlap=overlap, sync_bn=sync)
    real, z, code, labels = shard_inputs(co, B_global, rank, nshards)
    losses = tr.train_step(real.cuda(), z.cuda(), code.cuda(), labels.cuda())
    torch.cuda.synchronize()
    out = {"losses": torch.tensor([losses["g_loss"], losses["d_loss"], losses["info_loss"]]),
           "g": G.arena.grad.detach().cpu().clone(), "d": D.arena.grad.detach().cpu().clone(),
           "rm": G.state_dict()["conv_blocks.2.running_mean"].detach().cpu().clone(), "rv": G.state_dict()["conv_blocks.8.running_var"].detach().cpu().clone()}
    torch.save(out, os.path.join(outdir, f"rank{rank}_of{world}.pt"))        # plain tensors: loaded back with weights_only=True
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_generator_sync(rank, world, port, outdir, family, B_global):
    """MNIST / dSprites generator (three BatchNorm layers) forward + backward on this rank's shard of a seeded global batch, with synchronised
    BatchNorm when world > 1; saves the images, this rank's gradient share and the running statistics."""
    import torch.nn.functional as F
    eg = importlib.import_module("ead-gan_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if world > 1:
        eg.dp.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    sync = eg.dp.SyncBN(world, rank) if world > 1 else None
    b = B_global // world
    sl = slice(rank * b, (rank + 1) * b)
    g = torch.Generator().manual_seed(5)
    if family == "mnist":
        from oracle import mnist_oracle as mo
        orc = mo.MnistOracle(seed=1, mlp=mo.make_approximator(123))
        G = eg.mnist.Generator(dtype="f32").to("cuda")
        z, code, labels = mo.draw_step_inputs(np.random.RandomState(5), B_global)
        args = (z[sl].cuda(), F.one_hot(labels, 10).float()[sl].cuda(), code[sl].cuda())
        shape = (B_global, 1, 32, 32)
    else:
        from oracle import dsprites_oracle as do
        orc = do.DspritesOracle(seed=1)
        G = eg.dsprites.Generator(dtype="f32").to("cuda")
        code = torch.rand(B_global, 4, generator=g) * 2 - 1
        onehot = F.one_hot(torch.randint(0, 3, (B_global,), generator=g), 3).float()
        args = (onehot[sl].cuda(), code[sl].cuda())
        shape = (B_global, 1, 64, 64)
    G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
    dimg = torch.randn(shape, generator=torch.Generator().manual_seed(3)) * 1e-2
    ge = G.engine(b)
    img = ge.forward(*args, sync=sync).float().cpu().clone()
    grad = torch.zeros_like(G.arena.grad)
    if family == "mnist":
        ge.backward(dimg[sl].cuda(), grad, None, sync=sync)
    else:
        ge.backward(dimg[sl].cuda(), grad, sync=sync)
    torch.cuda.synchronize()
    sd = G.state_dict()
    out = {"img": img, "grad": grad.cpu().clone(),
           "running": torch.cat([v.detach().float().cpu().flatten() for k, v in sd.items() if "running_" in k])}
    torch.save(out, os.path.join(outdir, f"{family}_rank{rank}_of{world}.pt"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
