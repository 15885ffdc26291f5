"""Parameter containers of the hot path with the reference's checkpoint schema.

These modules own the tensors (so optimisers, ``state_dict``/``load_state_dict`` and
``nn.DataParallel`` wrappers of the calling scripts keep working) but the per-point network is never
evaluated by PyTorch: ``NeRF`` parameters are consumed by the HIP kernels through
:class:`mofanerf_amd.hipnet.HipNet`.  Only the two per-call encoders (StyleModule: one 50-d row per
call; texture encoder: one 512² image per call) run as PyTorch-ROCm modules, as SURVEY.md §2 scopes.

State-dict keys equal the reference's (models/model.py:97-110,177-187,206-223;
models/tex_encoder_mod.py:39-73) — see :mod:`mofanerf_amd.schema`.
"""
from __future__ import annotations

import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import schema


_EMBEDDED_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()     # NeRF module -> HipNet (NeRF.forward on embedded inputs)


def _xavier_relu_(m: nn.Module) -> None:
    for mod in m.modules():
        if isinstance(mod, nn.Linear):
            nn.init.xavier_uniform_(mod.weight.data, gain=nn.init.calculate_gain("relu"))


class _Stack(nn.Module):
    """Children named ``Linear0..LinearN-1`` (the reference interleaves parameter-free ReLU modules,
    which contribute no keys)."""

    def __init__(self, dims):
        super().__init__()
        for i, (cin, cout) in enumerate(dims):
            self.add_module(f"Linear{i}", nn.Linear(cin, cout))

    def layers(self):
        return [getattr(self, f"Linear{i}") for i in range(len(self._modules))]


class SkipMLP(nn.Module):
    def __init__(self, D: int, W: int, input_ch: int, skip):
        super().__init__()
        if skip is None:                     # xyzEncode: D+1 layers
            self.linears1 = _Stack([(input_ch, W)] + [(W, W)] * D)
            self.linears2 = _Stack([])
        else:
            self.linears1 = _Stack([(input_ch, W)] + [(W, W)] * skip)
            self.linears2 = _Stack([(W + input_ch, W)] + [(W, W)] * (D - skip - 2))


class NeRF(nn.Module):
    """Same constructor signature and parameter names as the reference ``NeRF`` (models/model.py:80-114)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, input_ch_textureCodes=10, input_ch_shapeCodes=128,
                 output_ch=4, skips=(4,), use_viewdirs=False):
        super().__init__()
        if not use_viewdirs:
            raise NotImplementedError("use_viewdirs=False: the reference's own NeRF.forward cannot run without view directions "
                                      "(models/model.py:121-137 uses alpha_linear / rgb_linear, created only for use_viewdirs=True)")
        if list(skips) != [schema.SKIP]:
            raise NotImplementedError("skips must be [4] (tools/create_model_condition.py:23)")
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.input_ch_shapeCodes, self.input_ch_textureCodes = input_ch_shapeCodes, input_ch_textureCodes
        self.skips, self.use_viewdirs = list(skips), use_viewdirs
        self.xyzEncode = SkipMLP(3, W, input_ch, None)
        self.linear_BiM_xyz = SkipMLP(D, W, input_ch_shapeCodes + W, skips[0])
        self.linear_uv_xyzBiM = SkipMLP(D, W, input_ch_textureCodes + W, skips[0])
        self.linear_view_xyBMuv = nn.Sequential(nn.Linear(input_ch_views + W, W // 2))
        self.alpha_linear = nn.Sequential(nn.Linear(W, 1))
        self.rgb_linear = nn.Linear(W // 2, 3)
        _xavier_relu_(self)

    def ordered_linears(self):
        """The 2D+7 Linear modules in state-dict (= C-ABI) order."""
        out = []
        for blk in (self.xyzEncode, self.linear_BiM_xyz, self.linear_uv_xyzBiM):
            out += blk.linears1.layers() + blk.linears2.layers()
        return out + [self.linear_view_xyBMuv[0], self.alpha_linear[0], self.rgb_linear]

    def forward(self, input_pts, input_bmCodes, input_views, input_uvCodes):
        """The reference module's own call form (models/model.py:121-137): per-point ALREADY-EMBEDDED inputs
        ``[n,93] [n,50] [n,27] [n,256] -> [n,4]`` (rgb pre-sigmoid, sigma pre-ReLU) — what the reference's eager ``batchify``
        hands to the network.  Runs on the same MFMA layer kernel without the constant folding (HipNet.forward_embedded).
        Inference only: the renderer's own path (``run_network`` / ``render`` / ``render_fitting``) is the one with a HIP backward,
        so a call that would need gradients here fails loudly instead of silently detaching."""
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(torch.is_tensor(t) and t.requires_grad for t in (input_pts, input_bmCodes, input_views, input_uvCodes))):
            raise RuntimeError("NeRF.forward on embedded inputs is inference-only on the HIP path (call it under torch.no_grad()); "
                               "gradients flow through Renderer.run_network / render / render_fitting")
        from . import lib
        from .hipnet import HipNet
        if not (torch.is_tensor(input_pts) and input_pts.is_cuda and next(self.parameters()).is_cuda):
            raise lib.MofaError("NeRF.forward needs its parameters and inputs on the GPU (net.cuda()); there is no CPU path")
        h = _EMBEDDED_CACHE.get(self)        # device-side cache kept OUTSIDE the module: deepcopy / pickling of the module stay plain
        if h is None:
            # embedded=True: on embedded inputs EVERY column of input_ch / input_ch_views is a per-point column — any widths, like the
            # reference's module; only the Linear shapes are checked
            h = _EMBEDDED_CACHE[self] = HipNet(self, weak=True, embedded=True)      # weak back-reference: the entry dies with the module
        return h.forward_embedded(input_pts, input_bmCodes, input_views, input_uvCodes)


class StyleModule(nn.Module):
    """Identity-specific expression modulation (models/model.py:174-199); 1×50 input, runs once per call."""

    def __init__(self, D=4, W=256, input_ch_bm=50, out_ch=30):
        super().__init__()
        self.linears1 = _Stack([(input_ch_bm, W)] + [(W, W)] * (D - 1))
        self.linears_scale = nn.Linear(W, out_ch)
        self.linears_bias = nn.Linear(W, out_ch)
        _xavier_relu_(self)

    def forward(self, bmcodes):
        h = bmcodes
        for lin in self.linears1.layers():
            h = F.relu(lin(h))
        return self.linears_scale(h), self.linears_bias(h)


class _TexEncoderCore(nn.Module):
    def __init__(self, code_len: int):
        super().__init__()
        chans = [3, 32, 32, 32, 32, 64, 128, 256]
        seq = []
        for i in range(7):
            seq += [nn.Conv2d(chans[i], chans[i + 1], 4, 2, 1), nn.LeakyReLU(0.2)]
        self.down1 = nn.ModuleList([nn.Sequential(*seq)])
        self.down2 = nn.Sequential(nn.Linear(256 * 4 * 4, 512), nn.LeakyReLU(0.2))
        self.mu = nn.Linear(512, code_len)
        self.logstd = nn.Linear(512, code_len)          # in the checkpoint, never used (tex_encoder_mod.py:56,94)
        self.decoding = nn.Sequential(nn.Linear(code_len, code_len), nn.LeakyReLU(0.1),
                                      nn.Linear(code_len, code_len), nn.LeakyReLU(0.1),
                                      nn.Linear(code_len, code_len), nn.LeakyReLU(0.1))
        _xavier_relu_(self)

    def forward(self, x):
        x = self.down1[0](x).reshape(-1, 256 * 4 * 4)
        return self.decoding(self.mu(self.down2(x)))


class EnDeUVmap(nn.Module):
    """Texture map -> 256-d texture code (models/tex_encoder_mod.py:7-100).  Returns ``(code, {})``."""

    def __init__(self, uvCodesLen=256):
        super().__init__()
        self.encoder = _TexEncoderCore(uvCodesLen)

    def forward(self, uvMap, lossList=None):
        return self.encoder(uvMap), {}
