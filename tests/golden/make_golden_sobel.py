#!/usr/bin/env python3
"""The fixture that would PIN the Sobel filter of `paste_front`:

    kornia.filters.sobel(x)     (kornia 0.6.5, _env/Dockerfile:55; training/triplane.py:632,652)

kornia is not installed in the build container and cannot be (no package index); `panic3d_amd.paste.sobel_magnitude` restates its
published definition (3x3 kernels / 8, replicate padding, sqrt(gx^2 + gy^2 + 1e-6)) and is checked against an independent dense
convolution only (tests/test_host_cpu.py).  Run this where kornia exists:

    python tests/golden/make_golden_sobel.py        -> tests/golden/sobel_kornia.npz

It FAILS (exit code 2, instructions on stderr) where kornia is missing.  tests/test_host_cpu.py::test_sobel_against_kornia compares
the restatement (and, on the GPU box, tests/test_hip_synthesis.py the paste kernel's edge mask) with the fixture when it exists; where
kornia IS importable and the fixture is not there, the test FAILS with the command above instead of skipping; with neither it skips
and the row stays "parity unpinned" (paste.sobel_magnitude and the bench line's config.workload say so)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs():
    import torch
    g = torch.Generator().manual_seed(0)
    return torch.rand(2, 3, 64, 64, generator=g)


def main():
    try:
        import torch  # noqa: F401
        import kornia
    except ImportError:
        sys.stderr.write("make_golden_sobel.py: kornia is not importable here — nothing written, the paste's Sobel filter stays UNPINNED.\n"
                         "Run this script where `pip install kornia==0.6.5` is possible and commit tests/golden/sobel_kornia.npz.\n")
        return 2
    x = inputs()
    np.savez_compressed(os.path.join(HERE, "sobel_kornia.npz"), x=x.numpy(), sobel=kornia.filters.sobel(x).numpy(),
                        kornia_version=str(kornia.__version__))
    print("sobel_kornia.npz written with kornia", kornia.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())
