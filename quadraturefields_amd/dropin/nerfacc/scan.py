"""``from nerfacc.scan import exclusive_prod, exclusive_sum`` (field_rendering.py:11)."""
from quadraturefields_amd.field_rendering import exclusive_prod, exclusive_sum  # noqa: F401
