#!/bin/bash
# Dev tool: build libisdf_hip.so of ANOTHER git revision into variants/lib_<name>.so (the A/B partner of the working tree:
# ISDF_HIP_LIB=$PWD/variants/lib_<name>.so python bench.py ...; tools/ab_bench.sh runs every variants/lib_*.so back to back).
#   tools/build_rev_lib.sh <rev> <name>
set -e
REV=${1:-HEAD}; NAME=${2:-prev}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/isdf_rev_XXXX)
git -C "$ROOT" archive "$REV" isdf_amd/csrc include isdf_amd/build.py | tar -x -C "$TMP"
mkdir -p "$ROOT/variants" "$TMP/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -fno-gpu-rdc"
pids=()
for f in "$TMP"/isdf_amd/csrc/*.hip; do
  b=$(basename "$f" .hip); extra=""
  extra=$(python3 - "$TMP/isdf_amd/build.py" "$b.hip" <<'PY'
import re, sys, ast
src = open(sys.argv[1]).read()
m = re.search(r"^PER_FILE = (\{.*?\})\s*$", src, re.S | re.M)
d = ast.literal_eval(m.group(1)) if m else {}
print(" ".join(d.get(sys.argv[2], [])))
PY
)
  /opt/rocm/bin/hipcc $FLAGS $extra -c "$f" -o "$TMP/obj/$b.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants/lib_$NAME.so" "$TMP"/obj/*.o
rm -rf "$TMP"
echo built "$ROOT/variants/lib_$NAME.so"
