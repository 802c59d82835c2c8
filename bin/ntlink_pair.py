#!/usr/bin/env python3
"""Drop-in for the reference's `ntlink_pair.py` on the pair stage (see ntlink_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
from ntlink_amd.cli import ntlink_pair_main

sys.exit(ntlink_pair_main())
