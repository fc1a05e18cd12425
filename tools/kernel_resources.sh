#!/bin/bash
# VGPRs / scratch / LDS of every kernel of the library: metadata notes of the gfx950 code object in each translation
# unit's object file (quantize_amd/_ext/obj/*.o).   usage: tools/kernel_resources.sh [obj dir]
DIR=${1:-quantize_amd/_ext/obj}
TMP=$(mktemp -d)
for o in $DIR/*.o; do
  objcopy -O binary --only-section=.hip_fatbin $o $TMP/fat.bin 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co
done | python3 -c '
import sys, re
rows=[]; cur={}
for l in sys.stdin:
    m=re.match(r"\s+\.name:\s+(_Z\S+)", l)
    if m: cur["name"]=m.group(1)
    for k in (".vgpr_count", ".agpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
        m2=re.match(r"\s+%s:\s+(\d+)" % re.escape(k), l)
        if m2: cur[k]=int(m2.group(1))
    if ".wavefront_size" in l and cur.get("name"):
        rows.append(cur); cur={}
bad=0
for r in sorted(rows, key=lambda r:r["name"]):
    s=r.get(".private_segment_fixed_size",0)
    bad+= s>0
    print("%4d vgpr %3d agpr %5d scratch %6d lds  %s" % (r.get(".vgpr_count",0), r.get(".agpr_count",0), s, r.get(".group_segment_fixed_size",0), r["name"]))
print("kernels: %d, with scratch: %d" % (len(rows), bad))
'
rm -rf $TMP
