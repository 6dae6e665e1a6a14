"""HRNet on MI355X: same import path, constructor, call signature and state_dict as the reference's
`src/DeepNetworks/HRNet.py` (HRNet :172-211; Encoder :36-74; RecuversiveNet :77-134; Decoder :138-169;
ResidualBlock :7-33), computed by the hand-written gfx950 kernels of libhrnet_hip.so.

The nn.Module tree below only *holds parameters* under the reference's names (so checkpoints load and
`torch.manual_seed` reproduces the reference's default initialisation: the same layer types are created in
the same order).  None of the sub-modules is ever called: `HRNet.forward` hands the tensors to the C ABI
(`hrn_hrnet_forward`), which runs median -> stem -> encoder -> recursive pairwise fusion -> decoder.

Precision (`HRNet.precision`, or key "precision" in the config dict, or env HRNET_HIP_PRECISION):
    "fp32" (default)  exact-fp32 MFMA; matches the reference forward to ~1e-6 relative
    "bf16"            bf16 activations/weights, fp32 accumulation; the throughput path (BASELINE config 3)
    "bf16x3"          split-bf16: every fp32 value as a (hi, lo) pair of bf16, three bf16 MFMAs per product, fp32 accumulation;
                      matches the reference forward to ~1e-5 relative at ~3x the fp32 path's speed
"""
import os

import torch
import torch.nn as nn

from hrnet_hip import binding

_PRECISIONS = {"fp32": binding.F32, "f32": binding.F32, "float32": binding.F32, "bf16": binding.BF16, "bfloat16": binding.BF16,
               "bf16x3": binding.BF16X3}


class _Holder(nn.Module):
    """Parameter container: exists for state_dict()/parameters()/to(); never executed."""

    def forward(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__} holds parameters only; call HRNet(lrs, alphas)")


class ResidualBlock(_Holder):
    def __init__(self, channel_size=64, kernel_size=3):
        super().__init__()
        pad = kernel_size // 2
        self.block = nn.Sequential(nn.Conv2d(channel_size, channel_size, kernel_size, padding=pad), nn.PReLU(),
                                   nn.Conv2d(channel_size, channel_size, kernel_size, padding=pad), nn.PReLU())


class Encoder(_Holder):
    def __init__(self, config):
        super().__init__()
        cin, layers, k, c = config["in_channels"], config["num_layers"], config["kernel_size"], config["channel_size"]
        self.init_layer = nn.Sequential(nn.Conv2d(cin, c, k, padding=k // 2), nn.PReLU())
        self.res_layers = nn.Sequential(*[ResidualBlock(c, k) for _ in range(layers)])
        self.final = nn.Sequential(nn.Conv2d(c, c, k, padding=k // 2))


class RecuversiveNet(_Holder):
    def __init__(self, config):
        super().__init__()
        self.input_channels = config["in_channels"]
        self.num_layers = config["num_layers"]
        self.alpha_residual = config["alpha_residual"]
        k = config["kernel_size"]
        c = self.input_channels
        self.fuse = nn.Sequential(ResidualBlock(2 * c, k), nn.Conv2d(2 * c, c, k, padding=k // 2), nn.PReLU())


class Decoder(_Holder):
    def __init__(self, config):
        super().__init__()
        d, f = config["deconv"], config["final"]
        self.deconv = nn.Sequential(nn.ConvTranspose2d(d["in_channels"], d["out_channels"], d["kernel_size"], stride=d["stride"]),
                                    nn.PReLU())
        self.final = nn.Conv2d(f["in_channels"], f["out_channels"], f["kernel_size"], padding=f["kernel_size"] // 2)


def _check_config(config):
    e, r, d = config["encoder"], config["recursive"], config["decoder"]
    ok = (e["in_channels"] == 2 and e["kernel_size"] == 3 and e["channel_size"] == 64 and 0 <= e["num_layers"] <= binding.MAX_RES_LAYERS
          and r["in_channels"] == 64 and r["kernel_size"] == 3
          and d["deconv"]["in_channels"] == 64 and d["deconv"]["out_channels"] == 64 and d["deconv"]["kernel_size"] == 3
          and d["deconv"]["stride"] == 3 and d["final"]["in_channels"] == 64 and d["final"]["out_channels"] == 1
          and d["final"]["kernel_size"] == 1)
    if not ok:
        raise NotImplementedError(
            "the gfx950 kernels are specialised for the reference's shipped network (config/config.json:8-34): "
            "2->64 stem, 64-channel 3x3 convs, stride-3 k3 deconv, 1x1 final; got " + repr(config))


class _HRNetLazyTrainFunction(torch.autograd.Function):
    """`.train()` mode with `precision="bf16"`: the forward runs the bf16 INFERENCE kernels (what `precision` asks for; this is
    the path `src/predict.py` takes, whose `load_model` never calls `.eval()` and whose `get_sr_and_score` does not use
    `no_grad`: predict.py:86-100, :17-49) and keeps nothing; IF a backward pass follows, it re-runs the forward on the fp32
    training kernels first (hrn_hrnet_forward_train) and then hrn_hrnet_backward.  Gradients are those of the fp32 model at the
    same parameters; the one returned tensor carries bf16 rounding."""

    @staticmethod
    def forward(ctx, module, names, lrs, alphas, *params):
        packed, dt = module.packed_parameters()
        ctx.module, ctx.names = module, names
        ctx.versions = tuple(p._version for p in params)
        ctx.save_for_backward(lrs, alphas, *params)
        return binding.hrnet_forward(packed, dt, module._num_layers, module.fuse.alpha_residual, lrs, alphas)

    _warned = False

    @staticmethod
    def backward(ctx, d_sr):
        lrs, alphas, *params = ctx.saved_tensors
        m = ctx.module
        if not _HRNetLazyTrainFunction._warned:
            _HRNetLazyTrainFunction._warned = True
            import warnings
            warnings.warn("HRNet(precision='bf16').train(): the forward ran the bf16 inference kernels, this backward recomputes "
                          "the forward on the fp32 training kernels and returns the fp32 model's gradients (two forwards per step, "
                          "loss and gradient ~1e-2 apart). Train with precision='fp32' or 'bf16x3'.", RuntimeWarning, stacklevel=2)
        if tuple(p._version for p in params) != ctx.versions:
            raise RuntimeError("HRNet bf16 train-mode backward: a parameter was modified between forward and backward")
        packed = m._packed_f32()
        _, tws = binding.hrnet_forward_train(packed, lrs, alphas, m._num_layers, m.fuse.alpha_residual)
        named = dict(zip(ctx.names, params))
        grads = {k: torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for k, p in named.items()}
        binding.hrnet_backward(packed, named, grads, m._num_layers, m.fuse.alpha_residual, lrs, alphas, d_sr.contiguous(), tws)
        return (None, None, None, None) + tuple(grads[k].to(named[k].dtype) for k in ctx.names)


class HRNet(nn.Module):
    """HRNet(config["network"]); forward(lrs (B,L,H,W), alphas (B,L)) -> (B,1,3H,3W)."""

    def __init__(self, config):
        super().__init__()
        _check_config(config)
        self.encode = Encoder(config["encoder"])
        self.fuse = RecuversiveNet(config["recursive"])
        self.decode = Decoder(config["decoder"])
        self._num_layers = config["encoder"]["num_layers"]
        self.precision = config.get("precision", os.environ.get("HRNET_HIP_PRECISION", "fp32"))
        self._packed = None
        self._packed_key = None

    # -- packed-parameter cache: re-packed whenever a parameter was modified (optimizer step, load_state_dict, .to())
    def _dtype(self):
        try:
            return _PRECISIONS[str(self.precision).lower()]
        except KeyError:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}; got {self.precision!r}")

    def packed_parameters(self):
        dt = self._dtype()
        named = dict(self.named_parameters())
        key = (dt, binding.param_epoch) + tuple((p.data_ptr(), p._version) for p in named.values())
        if self._packed is None or self._packed_key != key:
            self._packed = binding.hrnet_pack(named, self._num_layers, dt)
            self._packed_key = key
        return self._packed, dt

    def forward(self, lrs, alphas):
        if lrs.dim() != 4:
            raise ValueError(f"lrs must be (B, L, H, W); got {tuple(lrs.shape)}")
        if lrs.shape[2] != lrs.shape[3]:
            raise ValueError("square low-res images only: the reference reinterprets (H,W) as (W,H) in its .view() "
                             "(HRNet.py:204), which is the identity only for H == W")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # .train() mode with grad enabled - the training loop (train.py:160-190), but also src/predict.py, which never calls
            # .eval() and uses no no_grad (predict.py:86-100, :17-49).  Autograd cannot tell us whether a backward pass will follow:
            #   precision "fp32" (default) / "bf16x3": the training forward of that precision (torch.ops.hrnet_hip.hrnet_forward_train), which
            #                               keeps its intermediates for the HIP backward (same numbers as the inference kernels of that mode);
            #   precision "bf16":           the bf16 inference kernels, as asked for; a backward pass, if one comes, first
            #                               recomputes the forward on the fp32 training kernels (_HRNetLazyTrainFunction).
            # In .eval() mode (validation, train.py:196-215) the inference kernels run and the result carries no autograd graph.
            names = [k for k, _ in self.named_parameters()]
            params = [p for _, p in self.named_parameters()]
            if self._dtype() == binding.BF16:
                return _HRNetLazyTrainFunction.apply(self, names, lrs.detach(), alphas.detach(), *params)
            # the dispatcher-registered training op (binding.py): hrn_hrnet_forward_train with hrn_hrnet_backward as its autograd formula
            if names != binding.hrnet_param_names(self._num_layers):
                raise RuntimeError("HRNet parameters are not in the reference's registration order")
            # precision "bf16x3" trains in split-bf16 too (conv forward, data and weight gradients on the bf16 matrix cores, ~2^-16 per product)
            dt = self._dtype()
            packed = self.packed_parameters()[0] if dt == binding.BF16X3 else self._packed_f32()
            sr, _tws = torch.ops.hrnet_hip.hrnet_forward_train(packed, lrs.detach().float().contiguous(), alphas.detach().float().contiguous(),
                                                               params, self._num_layers, bool(self.fuse.alpha_residual), dt)
            return sr
        packed, dt = self.packed_parameters()
        return torch.ops.hrnet_hip.hrnet_forward(packed, dt, self._num_layers, bool(self.fuse.alpha_residual), lrs.detach(), alphas.detach())

    def _packed_f32(self):
        named = dict(self.named_parameters())
        key = (binding.param_epoch,) + tuple((p.data_ptr(), p._version) for p in named.values())
        if getattr(self, "_packed32", None) is None or self._packed32_key != key:
            self._packed32 = binding.hrnet_pack(named, self._num_layers, binding.F32)
            self._packed32_key = key
        return self._packed32

    # -- staged access for parity tests / profiling (channels-last tensors in the storage dtype)
    def encode_views(self, lrs):
        packed, dt = self.packed_parameters()
        return binding.hrnet_encoder(packed, dt, self._num_layers, lrs)

    def fuse_views(self, emb, alphas):
        packed, dt = self.packed_parameters()
        return binding.hrnet_fuse(packed, dt, self._num_layers, self.fuse.alpha_residual, emb, alphas)

    def decode_state(self, fused):
        packed, dt = self.packed_parameters()
        return binding.hrnet_decoder(packed, dt, self._num_layers, fused)
