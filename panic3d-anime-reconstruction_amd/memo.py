"""One switch for the host-side memo layers in front of a `G.f` call.

Five results are reused between calls when nothing they were computed from has changed (object identity + `_version` of the
tensors involved): the latents of `TriPlaneGenerator.f` (`_ws_memo`), the styles / demodulation coefficients of a `StylePlan`,
the conditioning terms prepared per level (`SynthesisNetwork._cond_prepared`), the rays and label of a view
(`cameras.cached_view`) and the channels-last copy of the planes (`ImportanceRenderer._nhwc`).  They exist for `generate.py`'s
loop — 16 views of one subject with the same seeds and conditioning tensors — and save ~0.7 ms of a 2.8 ms call.

What they cannot see: a write that bypasses the version counter — `t.data.copy_(...)`, `t.data.mul_(...)`, a DLPack / raw-pointer
alias, a C extension writing into the storage.  PyTorch bumps `_version` for in-place operations on the tensor itself (and on its
views and `detach()`), not for those.  Such writes are UNSUPPORTED while memoisation is on: either write through the tensor
(`t.copy_`, `load_state_dict`, `misc.copy_params_and_buffers` all do), or call `TriPlaneGenerator.clear_memo()` after the write,
or turn the layers off: `P3D_NO_MEMO=1` in the environment, or `panic3d_amd.memo.set_enabled(False)`.
The switch also turns off the parameter-derived caches, which are keyed the same way — pre-scaled weights, f16 operand copies,
transposed ToRGB weights, `noise_const * strength`, the per-element strengths of a NoisePool, the StylePlan's concatenated affine
weights, flipped FIR filters — so with it
off EVERY call works from the tensors as they are in memory (and pays for it: a pass re-derives 30 M parameters); `clear_memo()`
drops them once.
"""
import os

_enabled = os.environ.get("P3D_NO_MEMO", "0") in ("", "0")


def enabled():
    return _enabled


def set_enabled(state):
    """Turn all five memo layers on / off together (process-wide); returns the previous state."""
    global _enabled
    prev, _enabled = _enabled, bool(state)
    if not _enabled:
        from . import cameras
        cameras.cached_view_clear()
    return prev
