"""The ray container at the boundary.

Field names and tuple behaviour match ``Rays`` of the reference (``examples/datasets/utils.py:7``), so code written
against it -- ``rays.origins``, ``rays.viewdirs``, unpacking, ``namedtuple_map`` -- runs unchanged.
"""
from typing import Any, Callable, NamedTuple


class Rays(NamedTuple):
    origins: Any     # [..., 3] ray origins
    viewdirs: Any    # [..., 3] unit directions


def namedtuple_map(fn: Callable, tup):
    """Rebuild ``tup`` with ``fn`` applied to every field that is not None."""
    mapped = [fn(field) if field is not None else None for field in tup]
    return tup.__class__(*mapped)
