"""``from nerfacc.volrend import accumulate_along_rays_, render_weight_from_density, rendering`` (utils.py:22)."""
from quadraturefields_amd.field_rendering import (accumulate_along_rays, accumulate_along_rays_,  # noqa: F401
                                                  render_transmittance_from_alpha, render_transmittance_from_density,
                                                  render_visibility_from_alpha, render_visibility_from_density,
                                                  render_weight_from_alpha, render_weight_from_density, rendering)
