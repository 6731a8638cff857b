"""CPU oracle for the LSI/PIP hot path -- TEST INFRASTRUCTURE, never imported by rayjoin_amd.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
