#!/usr/bin/env python
"""Launcher with the reference's name and argv (src/subpopr/inst/convertSNVtoAlleleFreq.py); the work is metasnv_amd/subpopr.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from metasnv_amd.subpopr import convert_snv_to_allele_freq_main as main

if __name__ == "__main__":
    main()
