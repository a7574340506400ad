#!/bin/bash
# instruction mix of conv64_ws_kernel<0>'s hot loop in an object file:  tools/ws_isa.sh <object>
D=$(mktemp -d /tmp/wsisa.XXXX)
cp "$1" $D/tu3.o
cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading tu3.o >/dev/null 2>&1
f=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $f > all.s
awk '/^[0-9a-f]+ <_Z16conv64_ws_kernelILi0ELi3EEv8ConvArgs>:/{p=1} /^[0-9a-f]+ <_Z16conv64_ws_kernelILi1ELi0/{p=0} p' all.s > ws0.s
wc -l ws0.s
a=$(grep -n v_mfma ws0.s | head -1 | cut -d: -f1); b=$(grep -n v_mfma ws0.s | tail -1 | cut -d: -f1); echo "hot loop lines $a $b"
sed -n "${a},${b}p" ws0.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-30}
echo "scratch ops: $(grep -c scratch_ ws0.s)   in hot loop: $(sed -n "${a},${b}p" ws0.s | grep -c scratch_)"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | grep -A40 "conv64_ws_kernelILi0" | grep -E "vgpr_count|agpr_count|spill|private_segment_fixed" | head
echo $D
