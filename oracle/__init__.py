"""CPU oracle for the MERLOT pretraining hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``merlot_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / timed baseline.

PARITY STATUS: **parity unpinned** for the floating-point model path.  The
reference (rowanz/merlot) is TF-1.15 graph code, ships no tests or golden
vectors, and tensorflow is neither installed nor installable here, so the
restatement in ``merlot_oracle.py`` cannot be executed against the reference.
It is pinned only where the reference IS importable in the build container:
``downstream/sort_story/score_permutations.py`` (four pure-python functions,
AST-extracted) and the tokenizer constants of ``utils/encode/encoder.py``;
see ``tests/golden/make_golden.py``.
"""
