"""CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/pba_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under photobundle_amd/ imports it.
"""
