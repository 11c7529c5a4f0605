#!/usr/bin/env python
"""Randomised parity sweep of the HIP path against the oracle (GPU box): python tools/fuzz_parity.py [--cases N] [--seed S]
[--kind rays|frames|all] [--mode relu|softplus|abs] [--only i,j,...] [--verbose].  The case generators are tests/parity_fuzz.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.parity_fuzz import main  # noqa: E402

if __name__ == "__main__":
    main()
