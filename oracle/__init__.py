"""CPU oracle of the plane-sweep path (test infrastructure only; see cds_oracle.py)."""
