"""ViM-style bidirectional ("v2") Mamba block -- import path of the reference's
mamba/mamba_ssm/modules/mamba_simple.py (Mamba :34-155, 201-379; Block :381-437)."""
from mamba_ssm.modules._core import Block, MambaCore  # noqa: F401


class Mamba(MambaCore):
    variant = "vim"
