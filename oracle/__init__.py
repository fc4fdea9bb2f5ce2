"""TEST INFRASTRUCTURE ONLY.

CPU oracles for the hot path.  Importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from fembrain_amd/ (the product path).
"""
