#!/usr/bin/env python3
"""The fixture that would PIN the iso-surface triangulation this repo restates without a reference run:

    skimage.measure.marching_cubes(vol, level, method='lewiner', gradient_direction='descent')   (_util/eg3d_metrics3d.py:186-210)

scikit-image is not installed in the build container and cannot be (no package index), so this is the generator to run in an
environment that has it (the reference pins scikit-image 0.19: _env/Dockerfile):

    python tests/golden/make_golden_mesh.py        -> tests/golden/mesh_lewiner_{64,128}.npz

It FAILS (exit code 2, instructions on stderr) where scikit-image is missing — it never writes a fixture from anything else.
tests/test_mcubes_cpu.py::test_triangulation_against_skimage_lewiner compares oracle/p3d_oracle_mc.c (which the HIP kernel equals bit
for bit: tests/test_hip_mcubes.py) with the fixture when it exists; where skimage IS importable and the fixture is not there, that
test FAILS with the command above instead of skipping; with neither, it skips and the row stays "parity unpinned"
(oracle/p3d_oracle_mc.c, volume.marching_cubes and the bench line's config.workload say so).  The Sobel filter of the paste has its own
generator: tests/golden/make_golden_sobel.py."""
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
    try:
        import skimage
        from skimage import measure
    except ImportError:
        sys.stderr.write("make_golden_mesh.py: scikit-image is not importable here — nothing written, the triangulation stays UNPINNED.\n"
                         "Run this script where `pip install scikit-image==0.19.*` is possible and commit tests/golden/mesh_lewiner_*.npz.\n")
        return 2
    for n, vol in volumes().items():
        v, f, nrm, val = measure.marching_cubes(vol, 0.5, spacing=(1, 1, 1), gradient_direction="descent", method="lewiner")
        np.savez_compressed(os.path.join(HERE, f"mesh_lewiner_{n}.npz"), vol_seed=n, verts=v, faces=f, normals=nrm, values=val,
                            skimage_version=str(skimage.__version__))
        print(f"mesh_lewiner_{n}.npz: {len(v)} vertices, {len(f)} faces (scikit-image {skimage.__version__})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
