"""CPU oracle for the ELIC_united hot path -- test infrastructure only (see oracle/rans_oracle.c header)."""
