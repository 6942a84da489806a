"""ead-gan_amd: MI355X-native (gfx950) implementation of EAD-GAN's adversarial-training hot path.

Host layer in Python (this package) over a C-ABI shared library of hand-written HIP kernels
(``csrc/`` -> ``libeadgan_hip.so``, declared in ``include/eadgan_hip.h``).  The directory name contains a
hyphen, so import it with ``importlib.import_module("ead-gan_amd")`` or through the root alias module
``eadgan`` (``import eadgan``).
"""
from . import _lib, engine, ops            # noqa: F401
from . import celeba, colored, dp, dsprites, mnist, sampling, trunk     # noqa: F401

__all__ = ["ops", "engine", "celeba", "mnist", "dsprites", "colored", "trunk", "dp"]
