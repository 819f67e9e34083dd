"""Ray container at the boundary: same names as examples/datasets/utils.py:7-12 of the reference."""
import collections

Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))


def namedtuple_map(fn, tup):
    """Apply `fn` to every non-None field of a namedtuple and rebuild it."""
    return type(tup)(*(None if x is None else fn(x) for x in tup))
