"""CPU oracle for the RSIS hot path -- test infrastructure only (see rsis_oracle.py header)."""
