#!/usr/bin/env bash
# timings of the product library under environment knobs, ROUNDS rounds: tools/r05_env.sh <out-subdir> "VAR=val ..." ...
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05e}; shift
mkdir -p "$out"
for round in $(seq 1 ${ROUNDS:-2}); do
for v in "" "$@"; do
  echo "=== env '${v:--}'" >> "$out/env.log"
  env $v timeout 120 python tools/variant_bench.py 1 20 >> "$out/env.log" 2>&1
  env $v timeout 120 python tools/variant_bench.py 2 8 >> "$out/env.log" 2>&1
done
done
grep -v "^  " "$out/env.log" | sed 's/defines .*: carve/carve/; s/, groups.*//' | paste - - -
