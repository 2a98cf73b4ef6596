"""CPU oracle: test infrastructure only (see oracle/mlp_oracle.py header)."""
