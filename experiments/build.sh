#!/bin/bash
# Builds experiments/libkzhip_exp.so: the product sources with -DKZ_EXPERIMENTS plus the rejected kernel organisations of
# experiments/csrc/ (see README.md here).  The product build (kzero_amd/csrc/build.sh, no flag) compiles none of this.
set -euo pipefail
KZ_EXPERIMENTS=1 exec bash "$(cd "$(dirname "$0")/.." && pwd)/kzero_amd/csrc/build.sh"
