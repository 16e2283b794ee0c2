"""CPU restatement of the reference's hot path — TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; the product path (`lmms_owc_amd/`) never does and fails loudly without its HIP extension.
"""
