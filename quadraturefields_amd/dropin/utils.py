"""``from utils import (...)`` (train_finetune.py:17-24, test_baking_texture_images.py:22-29)."""
from quadraturefields_amd.utils import *  # noqa: F401,F403
from quadraturefields_amd.utils import (MIPNERF360_UNBOUNDED_SCENES, NERF_SYNTHETIC_SCENES, compress_sigma,  # noqa: F401
                                        derive_properties, generate_splits, inverse_of_compressed_sigma,
                                        render_image_bake_texture_images_with_occgrid, render_image_field_with_occgrid,
                                        render_image_finetune_with_occgrid, render_image_fit_sg_with_occgrid,
                                        render_image_with_occgrid, set_random_seed)
