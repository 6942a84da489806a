"""TEST INFRASTRUCTURE -- CPU restatement of the reference's sampling tools.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path never does.

Two parts with different pinning:

* ``sample_plan(kind, n)``: the generator inputs and file plan of each script's ``sample_image`` -- MNIST/EAD-GAN_rpqmnxy.py:276-330,
  MNIST/generate_image.py:97-138, celebA/EAD-GAN_celebA.py:233-287, celebA/gen_imgs.py:158-199, dSprites/rp.py:293-353,
  colored_dSprites/rp_color.py:291-353.  PINNED: tests/golden/sample_plans.npz holds what the reference functions themselves fed their
  generator and writer (recorded by oracle/ref_harness.record_sample_image, made by oracle/make_golden.py sample_plans).
* ``make_grid`` / ``to_uint8_hwc``: torchvision.utils.make_grid / save_image.  torchvision is a third-party dependency that is neither
  vendored in the reference nor importable here (release pairing the reference's torch 1.7.1: 0.8.2); this restates its published
  algorithm.  PARITY UNPINNED for this part: no reference fixture holds a written PNG; the tests anchor it on hand-checked small cases.
"""
import numpy as np
import torch

KINDS = ("mnist_train", "mnist_tool", "celeba_train", "celeba_tool", "dsprites_train", "colored_train")


def _onehot(y, n):
    out = np.zeros((len(y), n))
    out[range(len(y)), y] = 1.0
    return out


def _cols(ncode, n_rows, varied, which):
    """[n_rows, ncode] with column(s) ``which`` = varied, the rest 0 (the reference's np.concatenate((zeros, c_varied, ...), -1))"""
    c = np.zeros((n_rows, ncode))
    for w in which:
        c[:, w] = varied
    return c


def sample_plan(kind, n=10, rng=None):
    """-> list of dict(path=<dir under the script's image root>, inputs=tuple of float tensors or None for the real / transformed
    batches, gridded=<make_grid first, then save_image(normalize=True) of the grid>, sprite=<(x - 0.5) * 2 applied before the grid>)."""
    f = lambda a: torch.from_numpy(np.asarray(a)).float()
    plan = []
    if kind in ("mnist_train", "mnist_tool"):
        nz, ncls, ncode = 62, 10, 7
        label = _onehot(np.repeat(np.arange(10), 10), ncls)              # i-major: rows of one class (rpqmnxy.py:265-270, generate_image.py:88-93)
        z0 = np.zeros((ncls * 10, nz))
        if kind == "mnist_train":
            z = rng.normal(0, 1, (n ** 2, nz))                           # :279
            plan.append(dict(path="images/static", inputs=(f(z), f(label), f(np.zeros((ncls ** 2, ncode)))), gridded=False, sprite=False))
            plan.append(dict(path="images/original", inputs=None, gridded=True, sprite=False))
            plan.append(dict(path="images/scaled", inputs=None, gridded=True, sprite=False))
            varied = np.tile(np.linspace(-2, 2, n), n)                   # :297
            sets = [[i] for i in range(7)]
            root = "images"
        else:
            varied = -np.tile(np.linspace(-1, 1, n), n)                  # generate_image.py:103
            sets = [[0], [1, 2], [2], [3], [4], [5], [6]]                # c2 moves p and q together (:105)
            root = "test"
        for i, w in enumerate(sets):
            plan.append(dict(path=f"{root}/varying_c{i + 1}", inputs=(f(z0), f(label), f(_cols(ncode, n * n, varied, w))), gridded=True, sprite=False))
    elif kind in ("celeba_train", "celeba_tool"):
        nz, ncls, ncode = 200, 10, 8
        label = _onehot(np.array([num for _ in range(ncls) for num in range(ncls)]), ncls)     # class-minor (celebA.py:224-226)
        z0 = np.zeros((ncls ** 2, nz))
        if kind == "celeba_train":
            z = rng.normal(0, 1, (n ** 2, nz))                           # :236
            plan.append(dict(path="images/static", inputs=(f(z), f(label), f(np.zeros((ncls ** 2, ncode)))), gridded=False, sprite=False))
            plan.append(dict(path="images/original", inputs=None, gridded=True, sprite=False))
            plan.append(dict(path="images/scaled", inputs=None, gridded=True, sprite=False))
            sets = [[i] for i in range(8)]
        else:
            sets = [[0], [1, 2], [2], [3, 4], [4], [5], [6], [7]]        # gen_imgs.py:166,168: c2 and c4 move two codes together
        varied = np.repeat(np.linspace(-1, 1, n), n)                     # celebA.py:250, gen_imgs.py:164
        for i, w in enumerate(sets):
            plan.append(dict(path=f"images/varying_c{i + 1}", inputs=(f(z0), f(label), f(_cols(ncode, n * n, varied, w))), gridded=True, sprite=False))
    elif kind in ("dsprites_train", "colored_train"):
        ncls = 3
        ncode = 4 if kind == "dsprites_train" else 7
        plan.append(dict(path="images/original", inputs=None, gridded=True, sprite=True))
        plan.append(dict(path="images/trans", inputs=None, gridded=True, sprite=True))
        label = _onehot(np.repeat([0, 1, 2, 0, 1, 2, 0], n), ncls)       # rp.py:307-309
        varied = np.tile(np.linspace(-1, 1, n), 7)                       # :312
        sets = [[0], [1], [2], [3], [0], [0], [0]] if kind == "dsprites_train" else [[i] for i in range(7)]   # rp.py:319-321: c5..c7 repeat c1
        for i, w in enumerate(sets):
            cz = np.concatenate((label, _cols(ncode, n * 7, varied, w)), 1)
            plan.append(dict(path=f"images/varying_c{i + 1}", inputs=(f(cz),), gridded=True, sprite=True))
    else:
        raise ValueError(kind)
    return plan


def make_grid(t, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid(normalize=False) for a [B,C,H,W] tensor: single-channel images are replicated to 3 channels; a batch
    of one is returned as the image itself; cell k sits at row k // xmaps, column k % xmaps, each cell preceded by ``padding`` pixels."""
    t = t.detach().float().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.shape[1] == 1:
        t = torch.cat((t, t, t), 1)
    if t.shape[0] == 1:
        return t[0]
    B, C, H, W = t.shape
    xmaps = min(nrow, B)
    ymaps = -(-B // xmaps)
    ch, cw = H + padding, W + padding
    grid = torch.full((C, ch * ymaps + padding, cw * xmaps + padding), float(pad_value))
    for k in range(B):
        y, x = divmod(k, xmaps)
        grid[:, y * ch + padding:y * ch + padding + H, x * cw + padding:x * cw + padding + W] = t[k]
    return grid


def normalize_range(t):
    """norm_ip(t, float(t.min()), float(t.max())): clamp, subtract min, divide by (max - min + 1e-5) formed in double precision"""
    lo, hi = float(t.min()), float(t.max())
    return t.clamp(lo, hi).add(-lo).div(hi - lo + 1e-5)


def to_uint8_hwc(t, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """save_image's bytes: grid = make_grid(tensor, nrow, padding, normalize) (normalisation BEFORE tiling, so gaps stay pad_value; a
    3-D input -- an already tiled grid -- is normalised as it is, gaps included), then mul(255).add(0.5).clamp(0,255).to(uint8), HWC."""
    t = t.detach().float().cpu()
    if t.dim() == 3 and t.shape[0] == 1:
        t = torch.cat((t, t, t), 0)
    if normalize:
        t = normalize_range(t)
    g = make_grid(t, nrow, padding, pad_value) if t.dim() == 4 else t
    return g.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
