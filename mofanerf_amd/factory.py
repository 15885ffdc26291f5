"""``create_nerf(args)`` with the reference's contract (tools/create_model_condition.py:10-117).

Returns ``(render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, logger, render)`` with the
same dict keys, loads the reference's ``*.tar`` checkpoint schema unchanged, and places everything on
``args.device`` (the reference relies on a global ``set_default_tensor_type('torch.cuda.FloatTensor')``).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .embedder import get_embedder
from .model import NeRF
from .renderer import Renderer


def default_args(**over) -> SimpleNamespace:
    """Hot-path flags with the defaults of tools/config_parser.py overridden by configs/exp_mofanerf.txt."""
    a = dict(expname="mofanerf", basedir="./logs", netdepth=8, netwidth=256, netdepth_fine=10, netwidth_fine=1024,
             N_rand=1024, lrate=5e-5, lrate_decay=500, chunk=196608, netchunk=196608, no_reload=False, ft_path=None,
             N_samples=64, N_importance=64, perturb=1., use_viewdirs=True, i_embed=0, multires=10, multires_views=4,
             raw_noise_std=0., white_bkgd=False, dataset_type="blender", no_ndc=False, lindisp=False,
             input_ch_shapeCodes=50, input_ch_textureCodes=256, input_ch_expCodes=30,
             device="cuda" if torch.cuda.is_available() else "cpu")
    a.update(over)
    return SimpleNamespace(**a)


class _NullLogger:
    def write(self, *a, **k):
        pass


def _unwrap(m):
    return m.module if isinstance(m, torch.nn.DataParallel) else m


def save_checkpoint(path: str, global_step: int, render_kwargs, render, optimizer) -> str:
    """Write the reference's ``{:06d}.tar`` dictionary (run_train.py:369-379) so that the reference and this package
    resume from each other's files: same top-level keys, same state-dict key names and shapes (tests/golden/schema.json).
    ``DataParallel`` wrappers, which the reference unwraps with ``.module``, are accepted either way."""
    fine = render_kwargs.get("network_fine")
    blob = {
        "global_step": global_step,
        "network_fn_state_dict": _unwrap(render_kwargs["network_fn"]).state_dict(),
        "network_fine_state_dict": _unwrap(fine).state_dict() if fine is not None else None,
        "network_render_textureEncoder": _unwrap(render.texEncoder).state_dict(),
        "network_render_idSpecific": _unwrap(render.idSpecificMod).state_dict(),
        "optimizer_state_dict": optimizer.state_dict(),
        "expression_latent_codes_sigma": render.expCodes_Sigma,
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(blob, path)
    return path


def create_nerf(args):
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    if not args.use_viewdirs:
        raise NotImplementedError("use_viewdirs=False: the reference's own path for it cannot run (models/model.py:121-137 uses "
                                  "alpha_linear / rgb_linear, which exist only for use_viewdirs=True)")
    embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    dev = torch.device(getattr(args, "device", "cuda"))

    def mk(D, W):
        return NeRF(D=D, W=W, input_ch_shapeCodes=args.input_ch_shapeCodes, input_ch_textureCodes=args.input_ch_textureCodes,
                    input_ch=input_ch + args.input_ch_expCodes, output_ch=output_ch, skips=[4],
                    input_ch_views=input_ch_views, use_viewdirs=True).to(dev)

    model = mk(args.netdepth, args.netwidth)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = mk(args.netdepth_fine, args.netwidth_fine)
        grad_vars += list(model_fine.parameters())
    render = Renderer(embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=args.netchunk,
                      uvCodesLen=args.input_ch_textureCodes, expCodesLen=args.input_ch_expCodes).to(dev)
    grad_vars += list(render.grad_parameter())
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))

    start = 0
    ckpts = []
    if args.ft_path is not None and args.ft_path != "None":
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        if os.path.isdir(d):
            ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f]
    if len(ckpts) > 0 and not args.no_reload:
        ckpt_path = ckpts[-1]
        ckpt = torch.load(ckpt_path, map_location=dev)
        start = ckpt["global_step"]
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        model.load_state_dict(ckpt["network_fn_state_dict"])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt["network_fine_state_dict"])
        render.texEncoder.load_state_dict(ckpt["network_render_textureEncoder"])
        render.idSpecificMod.load_state_dict(ckpt["network_render_idSpecific"])
        for latent, saved in zip(render.expCodes_Sigma, ckpt["expression_latent_codes_sigma"]):
            latent.data[:] = saved[:].detach().clone()
        name = os.path.basename(ckpt_path)[:-4]
        start = int(name) if name.isdigit() else start

    if dev.type == "cuda":      # bind the networks to the device NOW (plan check, mofa_device_init's census): no first-call work inside render()
        render.bind(model, model_fine)

    render_kwargs_train = {
        "network_query_fn": render.run_network, "perturb": args.perturb, "N_importance": args.N_importance,
        "network_fine": model_fine, "N_samples": args.N_samples, "network_fn": model,
        "use_viewdirs": args.use_viewdirs, "white_bkgd": args.white_bkgd, "raw_noise_std": args.raw_noise_std,
    }
    if args.dataset_type != "llff" or args.no_ndc:
        render_kwargs_train["ndc"] = False
        render_kwargs_train["lindisp"] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test["perturb"] = False
    render_kwargs_test["raw_noise_std"] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, _NullLogger(), render
