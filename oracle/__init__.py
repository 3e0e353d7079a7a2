"""CPU oracle for the matcher train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product package
(``glue_factory_amd``) never imports this package and has no CPU fallback.

Pinning status: the reference (cvg/glue-factory) ships no golden vectors or
known-answer tests for this path, so the oracle is pinned
against outputs of the reference itself: ``oracle/gen_golden.py`` imports the
unmodified reference modules from ``/root/reference`` (build container only),
runs them on seeded inputs/weights and commits the results under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every restatement in
this package against those fixtures.
"""
