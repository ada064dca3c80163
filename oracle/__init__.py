"""CPU oracle (test infrastructure). See oracle/model.py for the import rules."""
