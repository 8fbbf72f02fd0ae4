"""CPU oracle for the hot path -- test infrastructure only (see m3_oracle.py)."""
