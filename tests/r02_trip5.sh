#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for b in 1024 2048 4096; do
  for ab in 0 8; do echo "== B=$b CS_CHAIN_ABLATE=$ab"; CS_CHAIN_ABLATE=$ab timeout 300 python tests/chain_stamps.py $b 2>&1 | grep -E "^\{|fwd grid|bwd grid|step timeline"; done
done
