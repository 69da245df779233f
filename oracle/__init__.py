"""CPU oracle (test infrastructure only; see sdf_oracle.py header)."""
