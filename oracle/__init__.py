"""CPU oracle for the EVI-RAG retriever hot path — TEST INFRASTRUCTURE ONLY.

A plain numpy / torch-CPU restatement of the reference's algorithms (each function cites the
reference file:line it follows).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this package, and only as the checker / the timed CPU baseline.
The product path (evi_rag_amd/) never imports it and has no CPU fallback.

Parity pins: tests/golden/*.npz hold inputs and outputs produced by running the reference's own
functions in the build container (tests/golden/make_golden.py); tests/test_oracle_golden.py checks
every function here against them.  Rows with no reference-side pin are marked "parity unpinned"
in their docstring and in DESIGN.md.
"""
