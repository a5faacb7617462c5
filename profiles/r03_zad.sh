#!/bin/bash
export TMPDIR=/tmp
timeout 200 python3 profiles/repro_wide.py calling_threshold=1 min_coverage=1 2>&1 | cut -c1-200 | head -30
