#!/bin/bash
# Instruction mix of the hot loop (first to last MFMA) of one conv64_ws_kernel instantiation in an object file:
#   tools/ws_isa.sh <object> [top N] [NRM] [EPI]      (default: 30, 0, 0)
D=$(mktemp -d /tmp/wsisa.XXXX)
cp "$1" $D/tu.o
cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading tu.o >/dev/null 2>&1
f=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $f > all.s
K="_Z16conv64_ws_kernelILi${3:-0}ELi${4:-0}EEv8ConvArgs"
awk -v k="$K" '/^[0-9a-f]+ <.*>:$/{p=0} $0 ~ "<" k ">:"{p=1} p' all.s > k.s
wc -l k.s
a=$(grep -n v_mfma k.s | head -1 | cut -d: -f1); b=$(grep -n v_mfma k.s | tail -1 | cut -d: -f1); echo "hot loop lines $a $b"
sed -n "${a},${b}p" k.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-30}
echo "scratch ops: $(grep -c scratch_ k.s)   in hot loop: $(sed -n "${a},${b}p" k.s | grep -c scratch_)"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | grep -A40 "$K" | grep -E "vgpr_count|agpr_count|spill|private_segment_fixed" | head -5
echo $D
