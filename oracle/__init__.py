"""TEST INFRASTRUCTURE -- CPU restatements of the reference's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under dkt_stereo_amd/ imports it (tests/test_layout.py
enforces that).
"""
