"""``import kaolin.render.spc as spc_render`` (utils.py:24, mesh_utils.py:13)."""
from quadraturefields_amd.spc_render import exponential_integration, mark_pack_boundaries, sum_reduce  # noqa: F401
