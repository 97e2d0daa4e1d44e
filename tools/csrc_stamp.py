#!/usr/bin/env python
"""sha256 of every kernel source (pydem_amd/csrc/*) as JSON: written next to a PMC profile when it is collected
(tools/collect_profiles.sh -> profiles/rNN_pmc_fetch_write_<size>.meta.json), so that bench.py only quotes `traffic`
from a profile whose kernels are the ones it is running (bench.csrc_hashes)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

print(json.dumps({'csrc_sha256': bench.csrc_hashes()}, indent=1, sort_keys=True))
