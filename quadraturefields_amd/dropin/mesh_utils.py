"""``from mesh_utils import MeshIntersection, MeshFinetune`` (train_finetune.py:2)."""
from quadraturefields_amd.mesh_utils import *  # noqa: F401,F403
from quadraturefields_amd.mesh_utils import MeshFinetune, MeshIntersection, RayIntersector  # noqa: F401
