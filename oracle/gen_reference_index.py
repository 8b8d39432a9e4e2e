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

# ---- tests/golden/reference_symbols.txt: where every named item of the reference's Rust and Python sources starts and ends
# (path, first line, last line, name) — an index, not source text.  tests/test_citations.py holds a citation that names an
# item ("generate_all_flat_moves_pov (chess.rs:439-481)") to that item's span, so that a citation whose lines have drifted
# fails instead of pointing a reader at the neighbouring function.
import re

SYM = os.path.join(os.path.dirname(OUT), "reference_symbols.txt")
RS = re.compile(r"^(\s*)(?:pub(?:\([^)]*\))?\s+)?(?:async\s+)?(?:unsafe\s+)?(?:const\s+)?(fn|struct|enum|trait|type|const|static|mod|macro_rules!)\s+([A-Za-z_][A-Za-z0-9_]*)")
IMPL = re.compile(r"^(\s*)(?:unsafe\s+)?impl\b(?:\s*<.*?>)?\s+(?:.*?\bfor\s+)?([A-Za-z_][A-Za-z0-9_]*)")
PY = re.compile(r"^(\s*)(?:async\s+)?(def|class)\s+([A-Za-z_][A-Za-z0-9_]*)")
sym_rows = []
for root, dirs, files in os.walk(REF):
    dirs[:] = sorted(d for d in dirs if d not in (".git", "target", "__pycache__"))
    for f in sorted(files):
        if not f.endswith((".rs", ".py")):
            continue
        p = os.path.join(root, f)
        try:
            lines = open(p, errors="ignore").read().split("\n")
        except OSError:
            continue
        pat = RS if f.endswith(".rs") else PY
        items = []  # (indent, start, name)
        for i, ln in enumerate(lines, 1):
            m = pat.match(ln)
            if m:
                items.append((len(m.group(1).expandtabs(4)), i, m.group(3)))
                continue
            mi = IMPL.match(ln) if f.endswith(".rs") else None
            if mi:  # an impl block ends the item in front of it; it is indexed under the type's name
                items.append((len(mi.group(1).expandtabs(4)), i, mi.group(2)))
        n = len(lines)
        for k, (ind, start, name) in enumerate(items):
            end = n
            for ind2, start2, _ in items[k + 1:]:
                if ind2 <= ind:
                    end = start2 - 1
                    break
            sym_rows.append(f"{os.path.relpath(p, REF)}\t{start}\t{end}\t{name}")
open(SYM, "w").write("\n".join(sym_rows) + "\n")
print(f"{len(sym_rows)} items -> {SYM}")
