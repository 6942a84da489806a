"""Container-only loader for the read-only reference at /root/reference  --  TEST INFRASTRUCTURE.

The reference's training files are scripts (argparse + dataset + epoch loop run at import time,
unconditional ``.cuda()``, torchvision imports).  This module parses them with ``ast`` and executes
(a) only the class/function definitions, or (b) the whole script with its dataset / argparse /
makedirs statements removed and a synthetic ``dataloader`` injected, so that the reference's own loop
body runs unmodified on torch-CPU.  Nothing from the reference is copied into this repository: the
source is read from /root/reference at run time and only numbers are written out
(``oracle/make_golden.py``).  This module is never imported on the GPU box.
"""
from __future__ import annotations

import argparse
import ast
import contextlib
import itertools
import math
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Variable
from torch.nn.utils import spectral_norm

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(REF_ROOT)


@contextlib.contextmanager
def _cpu_only_patches():
    """identity .cuda(), stub torchvision, no-op torch.save  (oracle process only)."""
    saved = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.save)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.save = lambda *a, **k: None
    stubs = {}
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets"):
        stubs[name] = sys.modules.get(name)
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision.utils"].make_grid = lambda t, *a, **k: t
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.save = saved
        for name, mod in stubs.items():
            if mod is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = mod


def _base_globals(opt):
    return dict(torch=torch, nn=nn, F=F, np=np, spectral_norm=spectral_norm, Variable=Variable,
                itertools=itertools, math=math, os=os, argparse=argparse, opt=opt,
                FloatTensor=torch.FloatTensor, LongTensor=torch.LongTensor, __name__="ref_exec")


def load_defs(rel_path: str, names, opt=None, extra=None):
    """Exec only the named ClassDef/FunctionDef nodes of a reference file; returns the globals."""
    path = os.path.join(REF_ROOT, rel_path)
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    g = _base_globals(opt)
    if extra:
        g.update(extra)
    with _cpu_only_patches():
        exec(compile(ast.Module(keep, []), path, "exec"), g)
    return g


_DROP_ASSIGN = {"parser", "opt", "dataloader", "dataset", "transform", "dataset_zip", "x_train", "x_train_tensor"}


def _filter_script(tree, drop_funcs):
    body = []
    for n in tree.body:
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            mod = getattr(n, "module", None) or ""
            names = [a.name for a in n.names]
            if mod.startswith("torchvision") or any(x.startswith("torchvision") for x in names):
                continue
        if isinstance(n, ast.Assign):
            tg = {t.id for t in n.targets if isinstance(t, ast.Name)}
            if tg & _DROP_ASSIGN:
                continue
        if isinstance(n, ast.Expr):
            s = ast.unparse(n)
            if "parser.add_argument" in s or "os.makedirs" in s or s.startswith("print(opt"):
                continue
        if isinstance(n, ast.FunctionDef) and n.name in drop_funcs:
            continue
        body.append(n)
    return ast.Module(body, [])


class RecordingLoader:
    """Synthetic ``dataloader``: yields the given batches; before handing out batch i+1 (and at the
    end) it snapshots the loss globals that iteration i left behind."""

    def __init__(self, batches, loss_names):
        self.batches, self.loss_names = batches, loss_names
        self.g = None
        self.records = []

    def __len__(self):
        return len(self.batches)

    def _snap(self):
        rec = {}
        for k in self.loss_names:
            v = self.g.get(k)
            if v is not None:
                rec[k] = float(v)
        self.records.append(rec)

    def __iter__(self):
        for i, b in enumerate(self.batches):
            if i > 0:
                self._snap()
            yield b
        self._snap()


def run_script_loop(rel_path: str, opt, batches, loss_names, seed: int, prereq=None,
                    drop_funcs=("sample_image",)):
    """Run the reference script's training loop on synthetic ``batches``.

    ``sample_image`` (PNG visualisation, out of scope) is replaced by a no-op, so its np.random draw at
    batches_done==0 is absent from the RNG stream; ``torch.save`` is a no-op.  Returns
    (globals, per-step loss records)."""
    path = os.path.join(REF_ROOT, rel_path)
    tree = _filter_script(ast.parse(open(path).read()), set(drop_funcs))
    loader = RecordingLoader(batches, loss_names)
    g = _base_globals(opt)
    g.update(dataloader=loader, sample_image=lambda *a, **k: None,
             save_image=lambda *a, **k: None, make_grid=lambda t, *a, **k: t)
    loader.g = g
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="eadgan_ref_")
    sys.path.insert(0, os.path.dirname(path))
    try:
        os.chdir(tmp)
        with _cpu_only_patches():
            if prereq:
                prereq(tmp)
            torch.manual_seed(seed)
            np.random.seed(seed)
            exec(compile(tree, path, "exec"), g)
    finally:
        os.chdir(cwd)
        sys.path.pop(0)
        for m in [m for m in sys.modules if m.startswith("utils_")]:
            sys.modules.pop(m)
    return g, loader.records


def celeba_opt(batch_size):
    """argparse defaults of celebA/EAD-GAN_celebA.py:39-51 with n_epochs=1."""
    return argparse.Namespace(n_epochs=1, batch_size=batch_size, lr=0.0002, b1=0.5, b2=0.999, n_cpu=8,
                              latent_dim=200, code_dim=8, n_classes=10, img_size=64, channels=3,
                              sample_interval=4000)


def mnist_opt(batch_size):
    """argparse defaults of MNIST/EAD-GAN_rpqmnxy.py:35-48 with n_epochs=1."""
    return argparse.Namespace(n_epochs=1, batch_size=batch_size, lr=0.0001, b1=0.5, b2=0.999, n_cpu=8, latent_dim=62, code_dim=7,
                              n_classes=10, img_size=32, channels=1, sample_interval=4000)


def dsprites_opt(batch_size):
    """argparse defaults of dSprites/rp.py:40-51 with n_epochs=1."""
    return argparse.Namespace(n_epochs=1, batch_size=batch_size, lr=0.0001, b1=0.5, b2=0.999, n_cpu=8, latent_dim=200, code_dim=4, n_classes=3,
                              img_size=64, channels=1, sample_interval=1000)


def colored_opt(batch_size):
    """argparse defaults of colored_dSprites/rp_color.py:40-51 with n_epochs=1."""
    return argparse.Namespace(n_epochs=1, batch_size=batch_size, lr=0.0002, b1=0.5, b2=0.999, n_cpu=8, latent_dim=200, code_dim=7, n_classes=3,
                              img_size=64, channels=3, sample_interval=1000)


def run_approximator_main(steps: int, seed: int):
    """MNIST/approximate_rpqmnxy.py: its ``__main__`` block (:109-153) fits the affine-inverse MLP.  The module is executed with the loop
    bound 20001 replaced by ``steps`` and a recorder call appended to the loop body; returns (globals, [affine_loss per iteration])."""
    path = os.path.join(REF_ROOT, "MNIST/approximate_rpqmnxy.py")
    tree = ast.parse(open(path).read())
    losses = []
    hits = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Call) and getattr(node.iter.func, "id", "") == "range" \
                and node.iter.args and isinstance(node.iter.args[0], ast.Constant) and node.iter.args[0].value == 20001:
            node.iter.args[0] = ast.Constant(steps)
            node.body.append(ast.parse("__record__(affine_loss)").body[0])
            hits += 1
    assert hits == 1, "training loop of approximate_rpqmnxy.py not found"
    ast.fix_missing_locations(tree)
    g = _base_globals(None)
    g.update(__name__="__main__", __record__=lambda v: losses.append(float(v)))
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="eadgan_ref_")
    try:
        os.chdir(tmp)
        with _cpu_only_patches():
            torch.manual_seed(seed)
            np.random.seed(seed)
            exec(compile(tree, path, "exec"), g)
    finally:
        os.chdir(cwd)
    return g, losses


def record_sample_image(rel_path: str, opt, call_args, img_shape):
    """Run the reference's ``sample_image`` (the PNG visualisation of the training scripts / generate_image.py / gen_imgs.py) with
    recording stand-ins for the generator and the torchvision writers.  Only ``sample_image`` / ``to_categorical`` and the module-level
    ``static_*`` definitions of the file are executed.  Returns (generator inputs per call, save_image calls as (path, nrow,
    normalize, source) where source names which tensor was written)."""
    path = os.path.join(REF_ROOT, rel_path)
    keep = []
    for n in ast.parse(open(path).read()).body:
        if isinstance(n, ast.FunctionDef) and n.name in ("sample_image", "to_categorical"):
            keep.append(n)
        elif isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id.startswith("static_") for t in n.targets):
            keep.append(n)
        elif isinstance(n, ast.For) and "static_label.append" in ast.unparse(n):
            keep.append(n)
    gen_calls, saves = [], []

    class _Tagged(torch.Tensor):
        pass

    def tag(t, name):
        t = t.as_subclass(_Tagged)
        t.src = name
        return t

    def generator(*inputs):
        gen_calls.append([i.detach().clone().float() for i in inputs])
        return tag(torch.zeros(inputs[0].shape[0], *img_shape), f"gen{len(gen_calls) - 1}")

    def src_of(t):
        return getattr(t, "src", None) or getattr(getattr(t, "data", None), "src", "arg")

    def make_grid(t, nrow=8, **k):
        return tag(torch.zeros(1), f"grid({src_of(t)},nrow={nrow})")

    def save_image(t, fp, nrow=8, normalize=False, **k):
        saves.append((str(fp), int(nrow), bool(normalize), src_of(t)))

    g = _base_globals(opt)
    g.update(generator=generator, make_grid=make_grid, save_image=save_image)
    state = np.random.get_state()
    with _cpu_only_patches():
        exec(compile(ast.Module(keep, []), path, "exec"), g)
        np.random.seed(0)
        g["sample_image"](*call_args)
    np.random.set_state(state)
    return gen_calls, saves
