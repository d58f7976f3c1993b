"""TEST INFRASTRUCTURE ONLY - CPU oracle for the PDP hot path.

Nothing under oracle/ is part of the shipped product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it, and only as the checker.  The product path
(pontryagin-differentiable-programming_amd/) never imports or links this package and fails loudly
when its HIP extension is missing.
"""
