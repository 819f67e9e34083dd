"""``from nerfacc.pack import pack_info`` (field_rendering.py:10)."""
from quadraturefields_amd.field_rendering import pack_info  # noqa: F401
