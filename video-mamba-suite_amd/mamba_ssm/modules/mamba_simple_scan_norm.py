"""ViM block with an RMSNorm between the merged scans and out_proj -- import path of the
reference's mamba/mamba_ssm/modules/mamba_simple_scan_norm.py."""
from mamba_ssm.modules._core import Block, MambaCore  # noqa: F401


class Mamba(MambaCore):
    variant = "vim_norm"
