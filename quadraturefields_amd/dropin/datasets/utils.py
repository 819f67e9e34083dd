"""``from datasets.utils import Rays, namedtuple_map`` (utils.py:16)."""
from quadraturefields_amd.datasets.utils import Rays, namedtuple_map  # noqa: F401
