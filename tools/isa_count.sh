#!/bin/bash
# Static instruction mix of fx::k_search<2> in the built library.  Usage: tools/isa_count.sh [lib.so]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=${1:-$ROOT/fuxi-planner_amd/libfxjps.so}
W=${TMPDIR:-/tmp}/fxisa.$$; mkdir -p $W; cd $W
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --input=$LIB --output=$W/dev.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null || \
  python3 - "$LIB" "$W/dev.co" <<'PY'
import sys
b=open(sys.argv[1],'rb').read()
i=b.find(b'\x7fELF',1)
# find the embedded AMDGPU ELF (e_machine 224)
while i>=0:
    if b[i+18:i+20]==b'\xe0\x00': break
    i=b.find(b'\x7fELF',i+1)
open(sys.argv[2],'wb').write(b[i:])
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $W/dev.co > $W/dev.s 2>/dev/null
awk '/<_ZN2fx8k_searchILi2ELb0ELb1EEEvNS_10SearchArgsE>:/{f=1;next} /^[0-9a-f]+ <.*>:/{f=0} f' $W/dev.s > $W/k2.s
echo "k_search<2, false, true>: total $(grep -cE '^\s+[a-z]' $W/k2.s)  valu $(grep -cE '^\s+v_' $W/k2.s)  salu $(grep -cE '^\s+s_' $W/k2.s)  cndmask $(grep -c v_cndmask $W/k2.s)  v_mov $(grep -c 'v_mov_b' $W/k2.s)  readlane $(grep -c v_readlane $W/k2.s)  writelane $(grep -c v_writelane $W/k2.s)  rfl $(grep -c v_readfirstlane $W/k2.s)  saveexec $(grep -c saveexec $W/k2.s)"
echo "asm: $W/k2.s"
