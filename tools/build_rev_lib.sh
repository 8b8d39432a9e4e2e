#!/bin/bash
# Builds kzero_amd/libkzhip_<name>.so from the kernel sources of a git revision (default HEAD), for same-box A/Bs against
# the working tree's library (tools/go_ab.sh, tools/split_ab.sh).  Usage: tools/build_rev_lib.sh [rev] [name]
set -euo pipefail
REV=${1:-HEAD}; NAME=${2:-cur}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" kzero_amd/csrc include | tar -x -C "$TMP"
(cd "$TMP/kzero_amd/csrc" && KZ_OUT="$ROOT/kzero_amd/libkzhip_$NAME.so" bash build.sh)
