#!/bin/bash
# disassemble one kernel of a built libt41rx: tools/isa_of.sh LIB MANGLED_SYMBOL OUT.s   (+ OUT_annot.txt with branch targets as line numbers)
set -e
LIB=$1; SYM=$2; OUT=$3
T=$(mktemp -d)
cp "$LIB" $T/lib.so
( cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null )
CO=$(ls -S $T/lib.so.*hipv4* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$SYM $CO > $OUT
python3 - "$OUT" <<'PY'
import re, sys
path = sys.argv[1]
lines = open(path).read().split('\n')
addr2line, base = {}, None
for i, l in enumerate(lines):
    m = re.search(r'//\s*([0-9A-F]{12}):', l)
    if m:
        a = int(m.group(1), 16)
        base = a if base is None else base
        addr2line[a] = i + 1
out = []
for i, l in enumerate(lines):
    m = re.search(r'<_Z.*\+0x([0-9a-f]+)>', l)
    t = ' --> L%d' % addr2line.get(base + int(m.group(1), 16), -1) if m else ''
    mm = re.match(r'\s+(\S+)\s+(.*?)\s*//', l)
    if mm:
        out.append('%5d  %s %s%s' % (i + 1, mm.group(1), mm.group(2)[:80], t))
open(path.replace('.s', '_annot.txt'), 'w').write('\n'.join(out))
PY
rm -rf $T
