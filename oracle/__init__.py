"""CPU oracle package (test infrastructure only; see oracle/oracle.py)."""
