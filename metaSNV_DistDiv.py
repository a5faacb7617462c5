#!/usr/bin/env python
"""Launcher with the reference's name and argv (metaSNV_DistDiv.py); the work is metasnv_amd/distdiv.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from metasnv_amd.distdiv import main

if __name__ == "__main__":
    main()
