"""CPU oracle for the afskmodem Receiver hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package ``afskmodem_amd`` never does.
"""
