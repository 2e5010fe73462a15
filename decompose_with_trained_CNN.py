#!/usr/bin/env python
"""Drop-in command line: same flags as the reference's decompose_with_trained_CNN.py."""
import sys

from reflectance_filtering_amd.decompose_with_trained_CNN import *  # noqa: F401,F403
from reflectance_filtering_amd.decompose_with_trained_CNN import main

if __name__ == "__main__":
    sys.exit(main())
