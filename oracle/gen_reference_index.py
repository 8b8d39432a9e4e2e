#!/usr/bin/env python3
"""Writes tests/golden/reference_index.txt: the relative path and the line count of every source file of the reference
(/root/reference, build container only).  Data, not source: tests/test_citations.py uses it to check that every
`file.rs:line` / `file.py:line` citation in this repository names a reference file that exists and has that many lines —
on any box, without the reference present."""
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_index.txt")
EXT = (".rs", ".py", ".toml", ".md", ".json", ".txt", ".lock")
rows = []
for root, dirs, files in os.walk(REF):
    dirs[:] = sorted(d for d in dirs if d not in (".git", "target", "__pycache__"))
    for f in sorted(files):
        if f.endswith(EXT):
            p = os.path.join(root, f)
            try:
                n = sum(1 for _ in open(p, "rb"))
            except OSError:
                continue
            rows.append(f"{os.path.relpath(p, REF)}\t{n}")
open(OUT, "w").write("\n".join(rows) + "\n")
print(f"{len(rows)} files -> {OUT}")
