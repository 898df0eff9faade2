"""ORACLE — test infrastructure only. See oracle/nvae_oracle.py."""
