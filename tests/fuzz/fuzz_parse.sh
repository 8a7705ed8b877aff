#!/bin/bash
# Builds tests/fuzz/fuzz_parse.cpp + the ABI with ASan / UBSan (host code) and fuzzes movi_index_parse on one small index per type.
# usage: tests/fuzz/fuzz_parse.sh [iterations per image = 20000] [work dir = /tmp/movi_fuzz]
set -e
IT=${1:-20000}; W=${2:-/tmp/movi_fuzz}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$W"
python3 - "$ROOT" "$W" <<'PY'
import sys
root, w = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from oracle import build_index as B
ref = B.read_fasta(root + "/tests/golden/ref.fasta")[0][1][:20000]
for mode in (2, 3, 5, 6, 7, 8):
    for sep in (False, True):
        open("%s/m%d%s.movi" % (w, mode, "s" if sep else ""), "wb").write(B.build_index_from_seqs([ref], mode, separators=sep))
PY
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer --offload-arch=gfx950 \
    -o "$W/fuzz_parse" "$ROOT/tests/fuzz/fuzz_parse.cpp" "$ROOT"/movi_amd/csrc/movi_abi.hip "$ROOT"/movi_amd/csrc/movi_kernels.hip "$ROOT"/movi_amd/csrc/movi_walk*.hip "$ROOT"/movi_amd/csrc/movi_expand_host.cpp
ASAN_OPTIONS=detect_leaks=0 "$W/fuzz_parse" "$IT" "$W"/m*.movi
