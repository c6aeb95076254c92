"""DBM (decomposed bidirectional Mamba) block -- import path of the reference's
mamba/mamba_ssm/modules/mamba_new.py (Mamba :34-122, 168-229)."""
from mamba_ssm.modules._core import Block, MambaCore  # noqa: F401


class Mamba(MambaCore):
    variant = "dbm"

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, init_layer_scale=None, scan_checkpoints=None):
        super().__init__(d_model, d_state=d_state, d_conv=d_conv, expand=expand, dt_rank=dt_rank, dt_min=dt_min,
                         dt_max=dt_max, dt_init=dt_init, dt_scale=dt_scale, dt_init_floor=dt_init_floor,
                         conv_bias=conv_bias, bias=bias, use_fast_path=use_fast_path, layer_idx=layer_idx,
                         device=device, dtype=dtype, init_layer_scale=init_layer_scale, scan_checkpoints=scan_checkpoints)
