#!/usr/bin/env python
"""Drop-in launcher with the reference driver's name and argv (metaSNV.py DIR all_samples REF_DB ...):
the work is done by libmsnv.so on the GPU, see metasnv_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from metasnv_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
