#!/usr/bin/env python3
"""Stamps a profile JSON written on the GPU box (no .git there) with the commit it was captured from:
    python tools/stamp_profile.py profiles/pmc_traffic.json [more.json ...]
Run in the build container right after copying the capture into profiles/ and BEFORE committing further source changes: the
recorded commit is HEAD, the commit whose tree was sent to the box."""
import json
import subprocess
import sys

head = subprocess.check_output(["git", "rev-parse", "HEAD"]).decode().strip()
count = int(subprocess.check_output(["git", "rev-list", "--count", "HEAD"]).decode())
for path in sys.argv[1:]:
    with open(path) as f:
        d = json.load(f)
    d["captured_commit"], d["captured_commit_count"] = head, count
    with open(path, "w") as f:
        json.dump(d, f, indent=1)
    print(path, head[:10], count)
