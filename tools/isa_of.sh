#!/bin/bash
# tools/isa_of.sh <object or .so> <demangled-name substring> [out.s]: disassembly of one gfx950 kernel of the library (design tooling)
# e.g. tools/isa_of.sh movi_amd/lib/obj/movi_walk_u32.o "flatp<6, unsigned int, 0, 0, 0, 1, 1, 0, 1>" /tmp/c3.s
set -e
OBJ=$1; PAT=$2; OUT=${3:-/dev/stdout}
W=$(mktemp -d)
python3 - "$OBJ" "$W" <<'PY'
import struct, sys
data = open(sys.argv[1], "rb").read()
pos, k = 0, 0
while True:
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
    if i < 0: break
    n = struct.unpack_from("<Q", data, i + 24)[0]
    p = i + 32
    for e in range(n):
        off, size, ts = struct.unpack_from("<QQQ", data, p); p += 24
        triple = data[p:p + ts].decode(); p += ts
        if "gfx950" in triple and size > 0:
            open("%s/co%d.o" % (sys.argv[2], k), "wb").write(data[i + off:i + off + size]); k += 1
    pos = i + 24
PY
B=/opt/rocm/lib/llvm/bin
for co in $W/co*.o; do
  sym=$($B/llvm-readelf -sW $co | awk '$4=="FUNC"{print $8}' | while read s; do d=$(echo $s | c++filt); case "$d" in *"$PAT"*) echo $s; break;; esac; done)
  if [ -n "$sym" ]; then $B/llvm-objdump -d --disassemble-symbols=$sym $co | sed -E 's/ *\/\/.*$//; s/<_Z[^>]*\+(0x[0-9a-f]+)>/<+\1>/' > $OUT; break; fi
done
rm -rf $W
