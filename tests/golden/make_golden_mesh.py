#!/usr/bin/env python3
"""Fixtures that would PIN the two third-party semantics this repo restates without a reference run:

  * skimage.measure.marching_cubes(vol, level, method='lewiner', gradient_direction='descent')   (_util/eg3d_metrics3d.py:186-210)
  * kornia.filters.sobel (kornia 0.6.5)                                                           (training/triplane.py:632,652)

Neither package is installed in the build container and `pip install scikit-image kornia==0.6.5` has no index to talk to
(profiles/history/r02_notes.txt records the attempt), so this script is the generator to run in an environment that has them:

    python tests/golden/make_golden_mesh.py        -> tests/golden/mesh_lewiner_{64,128}.npz, tests/golden/sobel_kornia.npz

tests/test_mcubes_cpu.py / tests/test_hip_synthesis.py pick the files up when they exist (vertex-set equality up to order,
surface area and Hausdorff distance of the meshes; Sobel magnitude on random images to 1e-6).  Until then both rows stay
"parity unpinned" (oracle/p3d_oracle_mc.c and paste.py say so)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def volumes():
    out = {}
    for n in (64, 128):
        g = np.linspace(-1, 1, n, dtype=np.float32)
        z, y, x = np.meshgrid(g, g, g, indexing="ij")
        rng = np.random.default_rng(n)
        vol = (0.6 - np.sqrt(x * x + 0.8 * y * y + 1.3 * z * z)) + 0.05 * rng.standard_normal((n, n, n)).astype(np.float32)
        out[n] = (1.0 / (1.0 + np.exp(-8.0 * vol))).astype(np.float32)  # densities in (0,1), surface at 0.5
    return out


def main():
    missing = []
    try:
        from skimage import measure
    except ImportError:
        measure = None
        missing.append("scikit-image")
    try:
        import torch
        import kornia
    except ImportError:
        kornia = None
        missing.append("kornia==0.6.5")
    if measure is not None:
        for n, vol in volumes().items():
            v, f, nrm, val = measure.marching_cubes(vol, 0.5, spacing=(1, 1, 1), gradient_direction="descent", method="lewiner")
            np.savez_compressed(os.path.join(HERE, f"mesh_lewiner_{n}.npz"), vol_seed=n, verts=v, faces=f, normals=nrm, values=val)
            print(f"mesh_lewiner_{n}.npz: {len(v)} vertices, {len(f)} faces")
    if kornia is not None:
        g = torch.Generator().manual_seed(0)
        x = torch.rand(2, 3, 64, 64, generator=g)
        np.savez_compressed(os.path.join(HERE, "sobel_kornia.npz"), x=x.numpy(), sobel=kornia.filters.sobel(x).numpy(),
                            kornia_version=str(kornia.__version__))
        print("sobel_kornia.npz written with kornia", kornia.__version__)
    if missing:
        print("not importable here:", ", ".join(missing), "-> those fixtures were NOT generated (parity stays unpinned)")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
