"""CPU restatement of the lyd405121/wcsph hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported baseline.  PARITY UNPINNED (see wcsph_ref.c).
"""
from .oracle import Oracle, build_oracle, set_num_threads, get_max_threads, poisson_sample, mc_value, mc_crossings, mc_triangles  # noqa: F401
