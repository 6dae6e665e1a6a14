"""Time ShiftNet's fc1 (the library named by HRNET_HIP_LIB) at B = 32 and check it against torch: prints us per launch and the max error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "highres-net_amd"), ROOT]
import torch
from DeepNetworks.ShiftNet import ShiftNet
from hrnet_hip import binding
sn = ShiftNet().cuda().eval()
x = torch.rand(32, 2, 128, 128, device="cuda")
with torch.no_grad():
    for _ in range(3):
        sn(x)
    binding.profile_enable(True)
    for _ in range(20):
        sn(x)
    torch.cuda.synchronize()
    binding.profile_enable(False)
    v = binding.profile_read()["fc1"]
print(f"{v['ms'] / v['launches'] * 1e3:.1f} us  {v['bytes'] / v['ms'] / 1e6:.0f} GB/s")
