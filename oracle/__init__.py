"""CPU oracle for the quadrature-field render path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU eager / numpy / one C file) of the
reference algorithm for the hot path named in BASELINE.json.  It is the checker
that ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` compare the HIP path against.  Nothing under
``quadraturefields_amd/`` imports, links or executes anything from here: the
product path fails loudly when the HIP extension is missing and has no CPU
fallback.

Pinning status:

* compositing (``oracle.volrend``): pinned by the seven docstring known-answer
  vectors of ``examples/field_rendering.py`` and by fixtures produced by
  executing the reference's own function bodies (``tests/golden/gen_from_reference.py``).
* quantisers / SG head / texture decode / ray generation / sample ordering:
  pinned by fixtures produced from the reference's own function bodies.
* hash grid, fused MLP, spherical harmonics (tiny-cuda-nn, un-vendored, unpinned
  git HEAD), kaolin 0.14 ``render.spc``, Embree/trimesh 3.23.5 multi-hit:
  **parity unpinned** -- the third-party source is absent from the container, so
  these are restatements of the published algorithms and
  the reference holds no test vector for them.
"""
