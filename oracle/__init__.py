"""CPU oracle for the MERTools hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker.  The product (mertools_amd) never imports it and has no CPU fallback.
"""
