#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ab in 0 4 16; do
echo "ablate=$ab"; CS_CHAIN_ABLATE=$ab python tests/chain_stamps.py 8192
done
