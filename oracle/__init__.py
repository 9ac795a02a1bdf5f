"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see dfm_oracle.c header)."""
