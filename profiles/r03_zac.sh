#!/bin/bash
export TMPDIR=/tmp
timeout 200 python3 profiles/repro_wide.py 2>&1 | cut -c1-220
echo "--- single-end"; timeout 200 python3 profiles/repro_wide.py frac_paired=0.0 2>&1 | cut -c1-220 | head -8
echo "--- no clips"; timeout 200 python3 profiles/repro_wide.py frac_clip_reads=0.0 2>&1 | cut -c1-220 | head -8
echo "--- overlaps counted"; timeout 200 python3 profiles/repro_wide.py ignore_overlaps=0 2>&1 | cut -c1-220 | head -8
