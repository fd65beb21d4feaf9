"""TEST INFRASTRUCTURE — not product code.

CPU restatement (torch-CPU / numpy, op-for-op in the reference's order) of the GDR
inference hot path, used only as the checker:

  * tests/                      parity of the HIP path against it
  * __graft_entry__.smoke()     one small check on cuda:0
  * bench.py `cpu_baseline`     the "port" timed on the GPU box's host cores

Nothing under gdr_amd/ may import this package; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Parity pin: every function here is checked against outputs of the reference itself,
imported on CPU in the build container by tests/golden/make_golden.py (the recipe of
SURVEY.md Appendix D); the resulting arrays are committed under tests/golden/ and
re-checked by tests/test_oracle_golden.py.  The reference ships no golden vectors or
tests of its own for this path.
"""
