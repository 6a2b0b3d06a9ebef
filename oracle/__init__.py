"""Test-infrastructure oracle for the KGAT propagation path (see kgat_oracle.py header)."""
