"""Import-only stand-ins for mamba/mamba_ssm/utils/hf.py (vivim.py:22-23 imports the names; there
is no network in this build and HF loading is out of scope)."""


def load_config_hf(model_name):
    raise NotImplementedError("HuggingFace hub access is outside this build's scope")


def load_state_dict_hf(model_name, device=None, dtype=None):
    raise NotImplementedError("HuggingFace hub access is outside this build's scope")
