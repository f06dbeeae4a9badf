"""CPU oracle of the reference algorithms -- test infrastructure only (see oracle/sloika_oracle.c)."""
