"""CPU oracle: test infrastructure only (see ssl_oracle.py header)."""
