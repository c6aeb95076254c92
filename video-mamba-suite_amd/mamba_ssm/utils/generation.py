"""Minimal stand-in for mamba/mamba_ssm/utils/generation.py: only what the video task code touches
(SURVEY.md 8b): the InferenceParams duck type read by Mamba.forward (:17-36) and an importable
GenerationMixin name (vivim.py imports it; LM decoding is out of scope)."""
from dataclasses import dataclass, field
from typing import Optional

from torch import Tensor


@dataclass
class InferenceParams:
    """Decode-time bookkeeping a caller hands to `Mamba.forward(..., inference_params=)`.

    What this build reads: `seqlen_offset` (0 = the prompt pass, which fills the per-layer caches; > 0 = single-token steps
    through `Mamba.step`) and `key_value_memory_dict`, which maps a layer's `layer_idx` to its (conv_state, ssm_state)
    pair (created on first use by `Mamba._get_states_from_cache`).  The remaining fields exist because the reference's
    decoding loop sets them (field names and defaults follow mamba_ssm/utils/generation.py:17-36 so that its callers run
    unchanged); nothing in the video suite's code paths consumes them."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[Tensor] = None

    def reset(self, max_seqlen, max_batch_size):
        """Back to the start of a new sequence with the same (re-usable) per-layer caches."""
        self.max_seqlen, self.max_batch_size, self.seqlen_offset = max_seqlen, max_batch_size, 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()


class GenerationMixin:
    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError

    def generate(self, *args, **kwargs):
        raise NotImplementedError("autoregressive text generation is outside this build's scope")
