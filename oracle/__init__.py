"""CPU oracle for the sparse-GP VMP hot path -- test infrastructure only (see sgp_oracle.py header)."""
