"""CPU oracle (test infrastructure only - see matcher_ref.py header)."""
