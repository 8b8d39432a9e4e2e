"""Every `file.rs:line` / `file.py:line` citation of the reference in this repository's sources and documents must name
a reference file that exists and is at least that long (tests/golden/reference_index.txt: paths and line counts of the
reference, written by oracle/gen_reference_index.py in the build container — the reference itself is not needed here).
Round 2 shipped a citation of `cuda_network.rs`, a file the reference does not have."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INDEX = {}
for line in open(os.path.join(REPO, "tests", "golden", "reference_index.txt")):
    path, n = line.rstrip("\n").split("\t")
    INDEX[path] = int(n)

CITE = re.compile(r"(?<![\w/.-])((?:[\w.-]+/)*[\w-]+\.(?:rs|py|toml)):(\d+)(?:-(\d+))?")
OWN_DIRS = ("kzero_amd", "tests", "tools", "oracle", "examples")


def own_files():
    names = set()
    for root, dirs, files in os.walk(REPO):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__") and not d.startswith("build")]
        names.update(files)
    return names


def scanned_files():
    out = []
    for top in ("kzero_amd", "include", "oracle", "tests", "examples"):
        for root, dirs, files in os.walk(os.path.join(REPO, top)):
            dirs[:] = [d for d in dirs if d not in ("golden", "__pycache__") and not d.startswith("build")]
            out += [os.path.join(root, f) for f in files if f.endswith((".hip", ".hpp", ".cpp", ".h", ".rs", ".py", ".c", ".sh"))]
    out += [os.path.join(REPO, f) for f in ("INTEGRATION.md", "DESIGN.md", "README.md", "bench.py", "__graft_entry__.py")]
    return out


def resolve(cited):
    return [p for p in INDEX if p == cited or p.endswith("/" + cited)]


def test_every_reference_citation_resolves():
    mine = own_files()
    bad, checked = [], 0
    for path in scanned_files():
        text = open(path, errors="ignore").read()
        for m in CITE.finditer(text):
            cited, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            base = os.path.basename(cited)
            matches = resolve(cited)
            if not matches:
                if base in mine:  # a citation of one of this repository's own files (bench.py:427, ...)
                    continue
                bad.append(f"{os.path.relpath(path, REPO)}: {m.group(0)} — no such file in the reference")
                continue
            checked += 1
            if not any(lo <= hi <= INDEX[p] for p in matches):
                bad.append(f"{os.path.relpath(path, REPO)}: {m.group(0)} — {matches[0]} has {INDEX[matches[0]]} lines")
    assert checked > 150, checked
    assert not bad, "\n".join(bad)
