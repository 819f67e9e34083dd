"""nerfacc 0.5.3 names the reference imports, on the gfx950 kernels."""
