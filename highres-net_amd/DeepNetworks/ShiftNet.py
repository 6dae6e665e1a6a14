"""ShiftNet on MI355X: same import path, constructor, forward/transform signatures and state_dict as the
reference's `src/DeepNetworks/ShiftNet.py` (:9-47 layers, :49-75 forward, :77-90 transform).

The layer objects only hold parameters / BatchNorm buffers under the reference's names; `forward` runs
`hrn_shiftnet_forward` (mean subtraction, 8 x conv+BN+ReLU(+pool) on the fp32 MFMA path, fc1+ReLU, fc2) and
`transform` runs the fused Lanczos kernel (`hrn_lanczos_shift`).
"""
import torch
import torch.nn as nn

import lanczos
from hrnet_hip import binding


class ShiftNet(nn.Module):
    def __init__(self, in_channel=1):
        super().__init__()
        if in_channel != 1:
            raise NotImplementedError("the gfx950 ShiftNet is specialised for in_channel=1 (the only value train.py uses)")
        chans = [(2 * in_channel, 64), (64, 64), (64, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128)]
        for i, (ci, co) in enumerate(chans, start=1):
            mods = [nn.Conv2d(ci, co, 3, padding=1), nn.BatchNorm2d(co), nn.ReLU()]
            if i in (2, 4, 6):
                mods.append(nn.MaxPool2d(2))
            setattr(self, f"layer{i}", nn.Sequential(*mods))
        self.drop1 = nn.Dropout(p=0.5)
        self.fc1 = nn.Linear(128 * 16 * 16, 1024)
        self.activ1 = nn.ReLU()
        self.fc2 = nn.Linear(1024, 2, bias=False)
        self.fc2.weight.data.zero_()        # identity transformation at start (reference ShiftNet.py:47)
        self._packed = None
        self._packed_key = None

    def _named(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def packed_parameters(self):
        named = dict(self.named_parameters())
        key = (binding.param_epoch,) + tuple((k, p.data_ptr(), p._version) for k, p in named.items() if ".1." not in k)   # BN tensors are read live
        if self._packed is None or self._packed_key != key:
            self._packed = binding.shiftnet_pack(self._named())
            self._packed_key = key
        return self._packed

    def forward(self, x):
        """x (B, 2, 128, 128) pairs (reference, image) -> (B, 2) translations (dx, dy)."""
        mask = None
        if self.training and self.drop1.p > 0:
            if self.drop1.p != 0.5:
                raise NotImplementedError("dropout p must be 0.5 (reference ShiftNet.py:43)")
            mask = (torch.rand((x.shape[0], 32768), device=x.device) >= 0.5).to(torch.uint8)
        named = self._named()
        if self.training and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training path: train-mode forward that keeps its intermediates + the HIP backward (parameters and input pairs)
            # the dispatcher-registered training op (binding.py): hrn_shiftnet_forward_train with hrn_shiftnet_backward as its autograd
            # formula (parameters AND the input pairs: the SR crops come from HRNet, so the registration loss trains it too)
            if [k for k, _ in self.named_parameters()] != binding.SHIFTNET_PARAM_NAMES:
                raise RuntimeError("ShiftNet parameters are not in the reference's registration order")
            buffers = [named[k] for k in binding.SHIFTNET_BUFFER_NAMES]
            theta, _tws, new_running = torch.ops.hrnet_hip.shiftnet_forward_train(
                self.packed_parameters(), x if x.dtype == torch.float32 else x.float(), [p for _, p in self.named_parameters()], buffers,
                float(self.layer1[1].momentum), mask)
            with torch.no_grad():
                torch._foreach_copy_(buffers, new_running)
                for i in range(1, 9):
                    getattr(self, f"layer{i}")[1].num_batches_tracked += 1
            return theta
        theta = binding.shiftnet_forward(self.packed_parameters(), named, x.detach(), train_bn=self.training,
                                         momentum=self.layer1[1].momentum, dropout_mask=mask)
        if self.training:
            for i in range(1, 9):
                getattr(self, f"layer{i}")[1].num_batches_tracked += 1
        return theta

    def transform(self, theta, I, device="cpu"):
        """Shift images I (B, 1, H, W) by theta (B, 2) = (dx, dy) with Lanczos interpolation -> (1, 1, B, H, W)."""
        self.theta = theta
        return lanczos.lanczos_shift(img=I.transpose(0, 1), shift=self.theta.flip(-1), a=3, p=5)[:, None]
