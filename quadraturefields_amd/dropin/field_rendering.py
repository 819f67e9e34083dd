"""``from field_rendering import rendering_field`` (utils.py:23)."""
from quadraturefields_amd.field_rendering import *  # noqa: F401,F403
