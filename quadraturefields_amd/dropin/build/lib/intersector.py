"""``from build.lib import intersector`` (train_finetune.py:216, mesh_utils.py:77)."""
from quadraturefields_amd.intersector import Intersector  # noqa: F401
