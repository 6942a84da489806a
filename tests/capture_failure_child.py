"""Child process of tests/test_gpu_capture.py: a hipGraph capture that is known to fail (a device synchronisation inside the captured body
invalidates the capture) must end the process the documented way -- reason on stderr, exit code 3, no further GPU work."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

eg = importlib.import_module("ead-gan_amd")


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = 4
    G, D = eg.celeba.Generator().to(dev), eg.celeba.Discriminator().to(dev)
    tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
    g = torch.Generator(device=dev).manual_seed(1)
    tr.load_inputs(torch.rand((B, 3, 64, 64), device=dev, generator=g) * 2 - 1, torch.randn((B, 200), device=dev, generator=g),
                   torch.rand((B, 8), device=dev, generator=g) * 2 - 1, torch.randint(0, 10, (B,), device=dev, generator=g))
    tr.step_resident()
    torch.cuda.synchronize()
    body = tr._step_body

    def bad_body():
        body()
        torch.cuda.synchronize()          # illegal under capture: hipErrorStreamCaptureUnsupported, the capture is invalidated

    tr._step_body = bad_body
    try:
        tr.capture()
    except eg.engine.CaptureFailed as exc:
        try:
            tr.step_resident()             # the trainer refuses to launch after a failed capture
        except eg.engine.CaptureFailed:
            print("[child] trainer refuses further launches", file=sys.stderr, flush=True)
        else:
            os._exit(7)
        eg.engine.exit_after_capture_failure(exc)
    os._exit(0)                            # the capture unexpectedly succeeded


if __name__ == "__main__":
    main()
