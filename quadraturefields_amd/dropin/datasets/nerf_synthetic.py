"""``from datasets.nerf_synthetic import SubjectLoader`` (train_finetune.py:272)."""
from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader, generate_rays  # noqa: F401
