"""No kernel reads past the end of an operand: an operand placed so that its last byte is the last byte of its own hipMalloc segment
(tools/guard_probe.py) turns an over-read into a GPU memory fault, which kills the process -- so every probe is a subprocess.
Round 5: the ragged right-to-left scans read [0, K) of a front-padded B / C row from lanes beyond the row -- K - seqlen elements past
the end of a row shorter than K, i.e. past the tensor behind its last row (found by tools/fuzz_modules.py after 63 cases had lined the
allocator up; selective_scan_fwd_pair.hip RawP::load_s, scan_bwd_helpers.h RawB::load_s)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "guard_probe.py")


def _probe(case, k):
    r = subprocess.run([sys.executable, PROBE, case, str(k)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (case, k, (r.stdout + r.stderr)[-800:])


@pytest.mark.parametrize("arg", [3, 4])          # B, C: the caller-padded operands (vms_hip.h bc_pad)
@pytest.mark.parametrize("case", ["scan_fwd_padded b3 d128 L8 bfloat16 rev1", "scan_bwd_padded b3 d128 L8 bfloat16 rev1",
                                  "scan_bwd_padded b2 d64 L5 bfloat16 rev1", "scan_fwd_padded b1 d32 L13 float16 rev0"])
def test_padded_bc_rows_are_not_read_past_the_tensor(case, arg):
    _probe(case, arg)


@pytest.mark.parametrize("case,arg", [("scan_fwd b3 d128 L8 bfloat16 rev1", 0), ("scan_fwd b3 d128 L8 bfloat16 rev1", 1),
                                      ("scan_bwd b2 d96 L1040 bfloat16 rev0", 8), ("conv_fwd b1 d32 L13 float16", 0),
                                      ("conv_bwd b2 d64 L24 bfloat16", 3)])
def test_activations_are_not_read_past_the_tensor(case, arg):
    _probe(case, arg)
