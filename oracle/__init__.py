"""ORACLE package — CPU restatement of the reference algorithm. TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product package."""
