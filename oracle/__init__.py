"""CPU oracle (test infrastructure only).  See oracle/m2f_oracle.py."""
