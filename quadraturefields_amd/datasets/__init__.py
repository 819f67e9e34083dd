from .utils import Rays, namedtuple_map  # noqa: F401
