"""ac_tsr_amd -- MI355X-native calibrated self-attention path of AC-TSR (AC-SASRec).

Only what the hot path needs: the HIP kernels + C ABI under `csrc/`,
their ctypes binding, and host-side mirrors of the reference's operator surface
(`AttackRTransformerEncoder`, `ACSASRec`, the two-pass trainer step, batch data-parallelism).
"""
from . import _lib  # noqa: F401
from .layers import (AttackRMultiHeadAttention, AttackRTransformerEncoder, AttackRTransformerLayer,  # noqa: F401
                     FeedForward)
from .model import ACSASRec, AcBERT4Rec, DictConfig, ItemCount, SequentialRecommender  # noqa: F401
from .ops import (AttentionConfig, ExplicitRandomness, StructuredMask, calibrated_attention,  # noqa: F401
                  materialize_randomness)
from .trainer import AttackSASRecTrainer, ACSASRecTrainer, is_attack_param  # noqa: F401

__all__ = [
    "ACSASRec", "AcBERT4Rec", "ACSASRecTrainer", "AttackRMultiHeadAttention", "AttackRTransformerEncoder",
    "AttackRTransformerLayer", "AttackSASRecTrainer", "AttentionConfig", "DictConfig", "ExplicitRandomness",
    "FeedForward", "ItemCount", "SequentialRecommender", "StructuredMask", "calibrated_attention",
    "is_attack_param", "materialize_randomness",
]
