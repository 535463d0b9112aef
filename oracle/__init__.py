"""CPU oracle -- test infrastructure only (see beam_oracle.py / beam_oracle.c headers).
Importable solely from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
