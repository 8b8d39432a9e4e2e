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


# ---- a citation that NAMES an item of the cited file must point at that item (round-4 review: `mapping.hpp` cited
# `generate_all_flat_moves_pov` with lines 459-507 of chess.rs while the function is at :439-481) ----
SYMBOLS = {}
for line in open(os.path.join(REPO, "tests", "golden", "reference_symbols.txt")):
    path, lo, hi, name = line.rstrip("\n").split("\t")
    SYMBOLS.setdefault(path, {}).setdefault(name, []).append((int(lo), int(hi)))
IDENT = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")
SLACK = 3  # lines a citation may stick out of the item it names (attributes and doc comments above, a closing brace below)
# names that are also ordinary words of the surrounding prose or generic ("new", "default" ...): not taken as naming an item
COMMON = {"new", "default", "from", "main", "test", "map", "get", "len", "index", "value", "policy", "values", "board", "moves",
          "run", "size", "clear", "push", "all", "evaluate", "forward", "Game", "build", "step", "flip", "name", "update", "load",
          "softmax", "executor", "storage", "graph", "ataxx", "chess", "tests"}


ADJACENT = re.compile(r"([A-Za-z_][A-Za-z0-9_]*)(?:\(\))?[`'\"*]*\s*[(\[]?\s*[`'\"]*$")


def named_item_spans(text, pos, cited_paths):
    """(name, hull of its spans) of the item named IMMEDIATELY in front of the citation at `pos` — `name (file:lo-hi)`,
    `name` file:lo-hi, Type::name (file:lo) — when that name is an item of the cited file; None otherwise.  A type with several
    impl blocks counts from its first line to the end of its last block."""
    line_start = text.rfind("\n", 0, pos) + 1
    m = ADJACENT.search(text[line_start:pos])
    if not m:
        return None
    ident = m.group(1)
    if ident in COMMON or len(ident) < 4:
        return None
    spans = [s for p in cited_paths for s in SYMBOLS.get(p, {}).get(ident, [])]
    if not spans:
        return None
    return ident, [(min(a for a, _ in spans), max(b for _, b in spans))]


def test_a_citation_that_names_an_item_points_at_that_item():
    bad, checked = [], 0
    for path in scanned_files():
        if path.endswith(("DESIGN.md", "HISTORY.md")):
            continue  # (prose with many citations per sentence: the sources, INTEGRATION.md and README.md are held to this)
        text = open(path, errors="ignore").read()
        for m in CITE.finditer(text):
            cited, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            matches = resolve(cited)
            if not matches:
                continue
            named = named_item_spans(text, m.start(), matches)
            if not named:
                continue
            ident, spans = named
            checked += 1
            if not any(s - SLACK <= lo and hi <= e + SLACK for s, e in spans):
                bad.append(f"{os.path.relpath(path, REPO)}: `{ident}` {m.group(0)} — the item spans {spans}")
    assert checked > 25, checked
    assert not bad, "\n".join(bad)
