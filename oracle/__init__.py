"""CPU oracle for the HighRes-net hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy, plus an own torch-CPU port used only
for the `cpu_baseline` timing) of the reference algorithm on the hot path:
  * HRNet.forward            /root/reference/src/DeepNetworks/HRNet.py:186-211
  * ShiftNet.forward/.transform  /root/reference/src/DeepNetworks/ShiftNet.py:49-90
  * lanczos_kernel / lanczos_shift   /root/reference/src/lanczos.py:5-107
  * get_loss / cPSNR / shift_cPSNR (callers; used for the parity metric)
        /root/reference/src/train.py:66-87, /root/reference/src/Evaluator.py:11-73

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product path (`highres-net_amd/`) never does and raises loudly
when the HIP library is missing.

Parity pin: the reference ships no tests, golden vectors or fixtures for this
path, so the oracle is pinned by outputs of the reference
itself, generated in the build container by `oracle/make_goldens.py` (which
imports /root/reference/src/{DeepNetworks,lanczos}) and committed as small
`.npz` fixtures under `tests/golden/`.
"""
