"""TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.

``oracle/`` holds the CPU checker for the TargetDiff denoising hot path:

* ``shims.py``            pure-torch stand-ins for the third-party ops the reference calls
                          (torch_scatter 2.1.0, torch_cluster 1.6.0 via torch_geometric 2.2.0);
                          semantics follow SURVEY.md Appendix B.  The reference has no tests at
                          these boundaries, so their tie/rounding rules are *parity-unpinned*
                          upstream; we define them here and the HIP kernels follow the same rule.
* ``reference_loader.py`` imports the real reference model files from /root/reference (only in the
                          build container; never on the GPU box) to generate golden vectors.
* ``restatement.py``      our own CPU restatement of the hot path (each function cites the
                          reference file:line it follows).  This is what travels to the GPU box
                          and what the ``-m gpu`` parity tests compare the HIP path against.  It is
                          pinned by ``tests/test_oracle_golden.py`` against fixtures produced by
                          the real reference (``make_golden.py``).
* ``weights.py`` / ``inputs.py``  deterministic weights and inputs shared by tests and bench.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
from this package.  Nothing under ``targetdiff_amd/`` does.
"""
