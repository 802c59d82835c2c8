#!/usr/bin/env python3
"""Drop-in for the reference's `ntlink_liftover_mappings.py` (see ntlink_amd/liftover.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
from ntlink_amd.liftover import main

sys.exit(main())
