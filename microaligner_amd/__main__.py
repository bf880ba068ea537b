"""`python -m microaligner_amd config.yaml` (counterpart of the `microaligner` console script, setup.py:70)."""
import sys

from .pipeline import main

if __name__ == "__main__":
    sys.exit(main())
