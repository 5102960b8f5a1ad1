"""Parity oracle — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot-path arithmetic (see the headers of ekf_ref.cpp,
lqr_ref.cpp, mpc_ref.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package; nothing under cpprobotics_amd/ does.
"""
from .oracle_lib import *  # noqa: F401,F403
