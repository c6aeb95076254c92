"""Minimal stand-in for mamba/mamba_ssm/utils/generation.py: only what the video task code touches
(SURVEY.md 8b): the InferenceParams duck type read by Mamba.forward (:17-36) and an importable
GenerationMixin name (vivim.py imports it; LM decoding is out of scope)."""
from dataclasses import dataclass, field
from typing import Optional

from torch import Tensor


@dataclass
class InferenceParams:
    """Inference parameters that are passed to the main model in order
    to efficiently calculate and store the context during inference."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[Tensor] = None

    def reset(self, max_seqlen, max_batch_size):
        self.max_seqlen = max_seqlen
        self.max_batch_size = max_batch_size
        self.seqlen_offset = 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()


class GenerationMixin:
    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError

    def generate(self, *args, **kwargs):
        raise NotImplementedError("autoregressive text generation is outside this build's scope")
