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
