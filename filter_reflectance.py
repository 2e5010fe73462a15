#!/usr/bin/env python
"""Drop-in command line: same flags as the reference's filter_reflectance.py."""
import sys

from reflectance_filtering_amd.filter_reflectance import *  # noqa: F401,F403
from reflectance_filtering_amd.filter_reflectance import main

if __name__ == "__main__":
    sys.exit(main())
