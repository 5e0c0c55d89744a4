"""CPU oracle for the HuManiFlow inference hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / NumPy restatement of the reference's
arithmetic for the path  ResNet features -> ancestor-conditioned SO(3) flow
-> SMPL blend-shapes + LBS.  It exists to *check* the HIP product path in
``humaniflow_amd``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never
does (``tests/test_boundary.py::test_product_never_imports_oracle`` enforces it).

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.so3``, ``oracle.flow.radial_tanh_*``, ``oracle.flow.so3_pushforward_log_prob``
  and ``oracle.resnet`` are PINNED: checked against the reference's own Python
  (imported from /root/reference in the build container by
  ``tests/golden/make_golden.py``) through the committed ``tests/golden/*.npz``.
* ``oracle.spline`` / ``oracle.flow`` hyper-network + coupling (pyro-ppl==1.7.0)
  and ``oracle.smpl`` (smplx==0.1.26) restate third-party arithmetic that is NOT
  vendored under /root/reference and not installable here (no network):
  **parity unpinned** for those two pieces -- they are anchored on the
  reference's call sites plus property tests (invertibility, log-det vs
  autograd, LBS identities) instead of golden vectors.
"""
