"""Test-only CPU oracle (see cca_oracle.py).  Never imported by the product packages."""
