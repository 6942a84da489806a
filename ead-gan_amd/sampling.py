"""Sampling tools around the eval-mode generators: the reference's ``sample_image`` functions and the torchvision writers
they call, on the device.

* ``make_grid`` / ``save_image``: torchvision.utils semantics at the reference's call sites (MNIST/EAD-GAN_rpqmnxy.py:281-330,
  MNIST/generate_image.py:122-138, celebA/EAD-GAN_celebA.py:238-287, celebA/gen_imgs.py:183-199, dSprites/rp.py:299-353,
  colored_dSprites/rp_color.py:297-353): tiling, the normalize=True range and the uint8 quantisation run as HIP kernels
  (``eg_make_grid``, ``eg_minmax_f32``, ``eg_quantize_u8``); only the finished HWC bytes cross PCIe, and the PNG container is written on
  the host with zlib (torchvision hands the same bytes to PIL).
* ``sample_image(kind, generator, ...)``: the latent-traversal plans of the six scripts, one table row per script.

There is no CPU path: tensors must live on the GPU and the calls go through the C ABI.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np
import torch

from . import ops


def _dev(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("sampling: tensors must be on the GPU (there is no CPU fallback)")
    return t.detach().float().contiguous()


def grid_shape(B, C, H, W, nrow=8, padding=2):
    xmaps = min(nrow, B)
    ymaps = -(-B // xmaps)
    return (3 if C == 1 else C, ymaps * (H + padding) + padding, xmaps * (W + padding) + padding)


def _minmax(t):
    out = torch.empty(2, device=t.device, dtype=torch.float32)
    ws = torch.empty(ops.minmax_ws_floats(), device=t.device, dtype=torch.float32)
    ops.minmax_f32(t, t.numel(), ws, out)
    return out


def _tile(t, nrow, padding, pad_value, rng2):
    B, C, H, W = t.shape
    if B == 1:                                        # a single image is returned as it is (3 channels, no border)
        nrow, padding = 1, 0
    grid = torch.empty(grid_shape(B, C, H, W, nrow, padding), device=t.device, dtype=torch.float32)
    ops.make_grid(t, B, C, H, W, nrow, padding, pad_value, rng2, grid)
    return grid


def _as4d(tensor):
    t = _dev(tensor)
    while t.dim() < 4:
        t = t.unsqueeze(0)
    return t


def make_grid(tensor, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """[B,C,H,W] (or one [C,H,W] image) -> [3 or C, Hg, Wg] fp32 on the device; normalize=True maps the images (not the gaps) to [0,1]."""
    t = _as4d(tensor)
    return _tile(t, nrow, padding, pad_value, _minmax(t) if normalize else None)


def to_uint8_hwc(tensor, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """The bytes save_image writes, as a host [H,W,C] uint8 array.  A 4-D batch is normalised BEFORE tiling (gaps stay pad_value); a 3-D
    tensor -- e.g. a grid made earlier -- is normalised as it is, gaps included (the reference's make_grid -> save_image(normalize=True))."""
    g = make_grid(tensor, nrow, padding, normalize, pad_value)
    C, H, W = g.shape
    out = torch.empty(H, W, C, device=g.device, dtype=torch.uint8)
    ops.quantize_u8(g, C, H, W, None, out)
    return out.cpu().numpy()


def encode_png(hwc: np.ndarray, level=6) -> bytes:
    """8-bit RGB / greyscale PNG (filter 0 on every row): the container PIL writes for torchvision."""
    assert hwc.dtype == np.uint8 and hwc.ndim == 3 and hwc.shape[2] in (1, 3)
    H, W, C = hwc.shape
    raw = np.concatenate((np.zeros((H, 1), np.uint8), hwc.reshape(H, W * C)), 1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2 if C == 3 else 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b"")


def save_image(tensor, fp, nrow=8, padding=2, normalize=False, pad_value=0.0):
    data = encode_png(to_uint8_hwc(tensor, nrow, padding, normalize, pad_value))
    d = os.path.dirname(fp)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(fp, "wb") as f:
        f.write(data)


# ------------------------------------------------------------------------------------------------
# sample_image of the six scripts.  Per kind: generator input widths, the class pattern of the rows, the traversal values and which
# code columns each varying_c<i> grid moves (the tools move two codes together in places -- kept as the scripts have it).
# ------------------------------------------------------------------------------------------------
def _lin(lo, hi, n):
    return np.linspace(lo, hi, n)


_PLANS = {
    # kind: (root, nz, ncls, ncode, label pattern, traversal values, moved columns, static sample?, extra real/transformed grids, sprite range)
    "mnist_train": ("images", 62, 10, 7, "class_major", lambda n: np.tile(_lin(-2, 2, n), n), [[0], [1], [2], [3], [4], [5], [6]],
                    True, ("original", "scaled"), False),                                       # MNIST/EAD-GAN_rpqmnxy.py:276-330
    "mnist_tool": ("test", 62, 10, 7, "class_major", lambda n: -np.tile(_lin(-1, 1, n), n), [[0], [1, 2], [2], [3], [4], [5], [6]],
                   False, (), False),                                                           # MNIST/generate_image.py:97-138
    "celeba_train": ("images", 200, 10, 8, "class_minor", lambda n: np.repeat(_lin(-1, 1, n), n), [[i] for i in range(8)],
                     True, ("original", "scaled"), False),                                      # celebA/EAD-GAN_celebA.py:233-287
    "celeba_tool": ("images", 200, 10, 8, "class_minor", lambda n: np.repeat(_lin(-1, 1, n), n), [[0], [1, 2], [2], [3, 4], [4], [5], [6], [7]],
                    False, (), False),                                                          # celebA/gen_imgs.py:158-199
    "dsprites_train": ("images", 0, 3, 4, "sprite", lambda n: np.tile(_lin(-1, 1, n), 7), [[0], [1], [2], [3], [0], [0], [0]],
                       False, ("original", "trans"), True),                                     # dSprites/rp.py:293-353
    "colored_train": ("images", 0, 3, 7, "sprite", lambda n: np.tile(_lin(-1, 1, n), 7), [[i] for i in range(7)],
                      False, ("original", "trans"), True),                                      # colored_dSprites/rp_color.py:291-353
}
KINDS = tuple(_PLANS)


def _labels(pattern, ncls, n):
    if pattern == "class_major":
        y = np.repeat(np.arange(10), 10)
    elif pattern == "class_minor":
        y = np.tile(np.arange(ncls), ncls)
    else:
        y = np.repeat(np.array([0, 1, 2, 0, 1, 2, 0]), n)
    out = np.zeros((len(y), ncls), np.float32)
    out[np.arange(len(y)), y] = 1.0
    return out


def sample_inputs(kind, n=10, rng=None):
    """-> [(directory, generator inputs as host float32 arrays or None for the real / transformed batch, gridded first?)] in file order.
    ``rng`` (np.random.RandomState-like) supplies the z ~ N(0,1) draw of the training scripts' static sample."""
    root, nz, ncls, ncode, pattern, values, moved, static, extra, sprite = _PLANS[kind]
    lab = _labels(pattern, ncls, n)
    rows = lab.shape[0]
    out = []
    if static:
        z = (rng if rng is not None else np.random).normal(0, 1, (n ** 2, nz)).astype(np.float32)
        out.append((f"{root}/static", (z, lab, np.zeros((ncls ** 2, ncode), np.float32)), False))
    out += [(f"{root}/{e}", None, True) for e in extra]
    v = values(n).astype(np.float64)
    for i, cols in enumerate(moved):
        code = np.zeros((rows, ncode), np.float32)
        for c in cols:
            code[:, c] = v
        inputs = (np.concatenate((lab, code), 1),) if sprite else (np.zeros((rows, nz), np.float32), lab, code)
        out.append((f"{root}/varying_c{i + 1}", inputs, True))
    return out


def sample_image(kind, generator, n=10, batches_done=0, real=None, trans=None, out_dir=".", rng=None, device="cuda"):
    """One call of the script's ``sample_image``: runs the generator (in whatever mode the caller left it -- the training scripts sample in
    train mode, the tools after ``.eval()``) on every traversal input and writes ``<out_dir>/<dir>/<batches_done>.png``.
    ``real`` / ``trans``: the two image batches the training scripts also dump.  Returns the written paths."""
    sprite = _PLANS[kind][9]
    given = [real, trans]
    paths = []
    with torch.no_grad():
        for d, inputs, gridded in sample_inputs(kind, n, rng):
            if inputs is None:
                img = given.pop(0)
                if img is None:
                    raise ValueError(f"sample_image({kind!r}) writes {d}: pass the real and transformed batches")
                img = _dev(img)
            else:
                img = _dev(generator(*[torch.from_numpy(a).to(device) for a in inputs]))
            if sprite:
                img = (img - 0.5) * 2
            fp = os.path.join(out_dir, d, f"{batches_done}.png")
            save_image(make_grid(img, nrow=n) if gridded else img, fp, nrow=n, normalize=True)
            paths.append(fp)
    return paths


# ------------------------------------------------------------------------------------------------
# checkpoints the tools read and the training scripts write
# ------------------------------------------------------------------------------------------------
def _new_generator(kind, dtype):
    from . import celeba, colored, dsprites, mnist
    family = kind.split("_")[0]
    if family == "mnist":
        return mnist.Generator(dtype=dtype)
    if family == "celeba":
        return celeba.Generator(dtype=dtype)
    if family == "dsprites":
        return dsprites.Generator(code_dim=4, n_classes=3, channels=1, dtype=dtype)
    if family == "colored":
        return colored.Generator(dtype=dtype)
    raise ValueError(kind)


def load_generator(kind, path, dtype="f32", device="cuda"):
    """What the tools do before sampling: MNIST/generate_image.py:143-152 (``generator_40000.pt`` is a bare state_dict) and
    celebA/gen_imgs.py:106-114 (``checkpoint_600000.tar`` holds it under 'generator_state_dict'); the generator comes back in eval
    mode.  The file is read with ``weights_only=True`` (tensors only, nothing in it is executed)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "generator_state_dict" in sd:
        sd = sd["generator_state_dict"]
    G = _new_generator(kind, dtype).to(device)
    G.load_state_dict(sd)
    return G.eval()


def save_checkpoint(path, generator, discriminator, epoch, batches_done):
    """celebA/EAD-GAN_celebA.py:414-423: the ``checkpoint_<n>.tar`` dictionary gen_imgs.py reads back"""
    torch.save({"discriminator_state_dict": discriminator.state_dict(), "generator_state_dict": generator.state_dict(),
                "epoch": epoch, "batches_done": batches_done}, path)


def run_tool(kind, checkpoint, out_dir=".", n=10, batches_done=0, dtype="f32", device="cuda"):
    """generate_image.py / gen_imgs.py as one call: load the checkpoint, eval mode, write the varying_c<i> grids."""
    if kind not in ("mnist_tool", "celeba_tool"):
        raise ValueError("run_tool: kind is 'mnist_tool' (MNIST/generate_image.py) or 'celeba_tool' (celebA/gen_imgs.py)")
    return sample_image(kind, load_generator(kind, checkpoint, dtype, device), n, batches_done, out_dir=out_dir, device=device)
