"""CPU: host-side logic of the package: module surface / state_dict contract, error behaviour without a GPU,
the sorting network used by the median kernel, the XCD tile remap and the weight-pack index math."""
import itertools
import os
import re

import numpy as np
import pytest
import torch

from oracle import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "highres-net_amd", "hrnet_hip", "csrc")


def test_state_dict_contract():
    from DeepNetworks.HRNet import HRNet
    from DeepNetworks.ShiftNet import ShiftNet
    m = HRNet(weights.HRNET_CONFIG)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in weights.HRNET_SHAPES]
    assert sum(p.numel() for p in m.parameters()) == 591818               # paper.txt:824
    s = ShiftNet()
    assert [(k, tuple(v.shape)) for k, v in s.state_dict().items()] == [(k, tuple(sh)) for k, sh in weights.shiftnet_shapes()]
    assert sum(p.numel() for p in s.parameters()) == 34187648             # paper.txt:730
    assert torch.count_nonzero(s.fc2.weight) == 0
    m.load_state_dict(weights.to_torch_state(weights.hrnet_state()))
    s.load_state_dict(weights.to_torch_state(weights.shiftnet_state()))


def test_no_cpu_fallback_and_config_checks():
    from DeepNetworks.HRNet import HRNet
    import lanczos
    m = HRNet(weights.HRNET_CONFIG).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm"):
        m(torch.zeros(1, 2, 8, 8), torch.ones(1, 2))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2, 8, 4), torch.ones(1, 2))
    bad = {k: dict(v) for k, v in weights.HRNET_CONFIG.items()}
    bad["encoder"]["channel_size"] = 32
    with pytest.raises(NotImplementedError):
        HRNet(bad)
    with pytest.raises(NotImplementedError):
        lanczos.lanczos_kernel(torch.zeros(1, 1), a=2)
    with pytest.raises(RuntimeError):
        lanczos.lanczos_shift(torch.zeros(1, 1, 8, 8), torch.zeros(1, 2))
    m.train()                                            # the training path has no CPU fallback either
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm"):
        m(torch.zeros(1, 2, 8, 8), torch.ones(1, 2))


def test_median_sorting_network_sorts():
    """The compare-exchange list in stem.hip must be a sorting network for 9 keys (0-1 principle, all 512 inputs)."""
    src = open(os.path.join(CSRC, "stem.hip")).read()
    body = src[src.index("sorting network for 9 keys"):src.index("const int k = (n - 1) >> 1")]
    net = [(int(a), int(b)) for a, b in re.findall(r"cswap\(v\[(\d)\], v\[(\d)\]\)", body)]
    assert len(net) == 25
    for bits in itertools.product((0, 1), repeat=9):
        v = list(bits)
        for a, b in net:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
        assert v == sorted(v)


@pytest.mark.parametrize("nwg", [1, 7, 8, 9, 63, 64, 1000, 65536])
def test_xcd_remap_is_a_bijection(nwg):
    """conv3x3.hip: logical = f(blockIdx) must hit every tile exactly once for any grid size."""
    bid = np.arange(nwg)
    xcd, q8, r8 = bid & 7, nwg >> 3, nwg & 7
    logical = np.where(xcd < r8, xcd * (q8 + 1), r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3)
    assert sorted(logical.tolist()) == list(range(nwg))


@pytest.mark.parametrize("cin,cout,es", [(64, 64, 2), (128, 128, 2), (128, 64, 4), (64, 128, 4)])
def test_conv_pack_index_math(cin, cout, es):
    """Restates conv_pack_kernel's index decode: every (co, ci, tap) is written exactly once, and the K order inside a
    128-byte row is the channel order the MFMA fragments assume."""
    kb = 128 // es
    nhalf = cout // 64
    total = cin * cout * 9
    idx = np.arange(total)
    kk, col, step = idx % kb, (idx // kb) % 64, idx // (kb * 64)
    half, ct = step % nhalf, step // nhalf
    tap, chunk = ct % 9, ct // 9
    co, ci = half * 64 + col, chunk * kb + kk
    flat = (co * cin + ci) * 9 + tap
    assert np.array_equal(np.sort(flat), np.arange(total))
    assert step.max() + 1 == (cin // kb) * 9 * nhalf


def test_stamp_instrumenters_still_apply(tmp_path):
    """tools/stamps/instr_*.py patch s_memtime stamps into copies of the conv kernels by text anchors (profiles/
    r01_final_inkernel_stamps.txt was made with them; conv3x3_v6 carries its stamps itself, -DV6_STAMP): every anchor must still
    exist in the current sources."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "highres-net_amd", "hrnet_hip", "csrc")
    for which, fn, marker in (("r64", "conv3x3_r64.hip", "hrn_dbg_read_stamps"),):
        d = tmp_path / which
        d.mkdir()
        shutil.copy(os.path.join(csrc, fn), d / fn)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "stamps", f"instr_{which}.py"), str(d)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        text = (d / fn).read_text()
        assert marker in text and text.count("STAMP(") >= 7
