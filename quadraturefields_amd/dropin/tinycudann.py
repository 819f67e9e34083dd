"""``import tinycudann as tcnn`` (radiance_fields/ngp.py:26, field.py)."""
from quadraturefields_amd.tinycudann import Encoding, Network, NetworkWithInputEncoding  # noqa: F401
