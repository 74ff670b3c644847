"""Test infrastructure only (see oracle/turtle_oracle.h).  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
turtle_amd/."""
