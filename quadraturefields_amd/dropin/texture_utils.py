"""``from texture_utils import FeatureCompression`` (test_baking_texture_images.py:332)."""
from quadraturefields_amd.texture_utils import FeatureCompression  # noqa: F401
