"""``from field import Field`` (train_finetune.py:16)."""
from quadraturefields_amd.field import BasicDecoder, Field  # noqa: F401
