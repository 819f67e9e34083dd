"""``from radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew`` (train_finetune.py:15)."""
from quadraturefields_amd.radiance_fields.ngp import *  # noqa: F401,F403
from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew  # noqa: F401
