"""CPU float64 oracle of the PMC hot path -- TEST INFRASTRUCTURE ONLY (see pmc_oracle.c)."""
