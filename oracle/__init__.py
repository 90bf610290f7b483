"""ORACLE -- test infrastructure, not product code.

A CPU restatement (numpy + plain C) of the reference's GP / mutual-information candidate-selection
path, used ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg.
Nothing under ital_amd/ imports this package; the product path fails loudly without its HIP library.

Pinned against the real reference: tests/golden/*.npz were produced by tests/golden/make_golden.py,
which imports /root/reference in the build container (serial mode, fresh process per fixture), and
tests/test_oracle_*.py check every function here against those vectors.  The MVNDST restatement
(oracle/mvndst_oracle.c) reproduces scipy.stats._mvn.mvndst bit for bit, random stream included.
"""
