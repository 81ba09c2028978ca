"""CPU oracle for the U2MKD training hot path.  TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``u2mkd_amd/`` imports it, and the product path raises if
the HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):

* ``ts_ref`` restates torchsparse **v1.4.0** (git+https://github.com/mit-han-lab/
  torchsparse.git@v1.4.0, pinned by reference README.md:44-48).  That library
  is NOT vendored under /root/reference and cannot be installed here, and the
  reference holds no test that pins its results: **parity unpinned** at that
  boundary.  The restatement is anchored instead on independent dense
  identities (``torch.nn.functional.conv3d`` / ``conv_transpose3d`` on dense
  grids, fp64 gradcheck, bincount means, the manual trilinear formula) in
  ``tests/test_oracle_torchsparse.py``.
* ``sptr_ref`` restates the in-tree CUDA sources
  ``third_party/SparseTransformer/src/sptr/**.cu`` and is pinned on the
  reference's own known-answer fixture ``test/test_precompute_all.py:9-19``
  plus a brute-force dense per-window attention.
* model wiring / losses are pinned on golden vectors generated in the build
  container by importing the reference's own Python modules over
  ``oracle.torchsparse_cpu`` (``tests/golden/make_golden.py``).
* ``f16x2_ref`` is not a restatement of reference code: a numpy model of the
  PRODUCT's own f16x2 multiply (two scaled fp16 planes per fp32 operand, three
  partial products), so that its accuracy bound against float64 can be checked
  on the CPU (``tests/test_f16x2_arithmetic.py``); the GPU kernels are held to
  the same bound in ``tests/test_gpu_conv_f16x2.py``.
"""
