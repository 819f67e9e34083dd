"""``from nerfacc.estimators.occ_grid import OccGridEstimator`` (train_finetune.py:25)."""
from quadraturefields_amd.estimators import OccGridEstimator  # noqa: F401
