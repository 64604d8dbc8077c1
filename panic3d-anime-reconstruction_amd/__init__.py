"""panic3d-anime-reconstruction_amd — MI355X-native (gfx950) triplane volumetric-rendering hot path of PAniC-3D.

Import name: `panic3d_amd` (see the shim panic3d_amd.py at the repository root; the directory name carries a hyphen).
Layout: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding), ops.py (tensor-level operators),
renderer.py (mirror of the reference's ImportanceRenderer call surface).
"""
from . import _build, _lib, memo, ops, cameras, sharding, stylegan2, generator, volume, outputs  # noqa: F401
from .renderer import ImportanceRenderer, decoder_params  # noqa: F401

build = _build.build
