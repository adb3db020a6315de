"""CPU oracle of the rollout hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (see oracle/oracle.cpp header).  PARITY UNPINNED: the reference has
no golden vectors for this path.
"""
from .oracle_py import (OracleResult, load, tick, samples, generate, velocity_iterator, radius_count, feed, path_blocked,  # noqa: F401
                        MarkingOracle, in_lidar_observation)
