#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python tools/dbg_fp8a8_slices.py 2>&1 < /dev/null | tail -16
