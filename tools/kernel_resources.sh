#!/bin/bash
# Register / scratch / LDS use of the kernels in the built objects (code-object metadata of rails_amd/csrc/*.o).
#   tools/kernel_resources.sh [regex]      default regex: mol_score
RE=${1:-mol_score}
DIR=$(dirname "$0")/../rails_amd/csrc
T=$(mktemp -d)
for o in "$DIR"/*.o; do
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$T/fat" "$o" 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$T/fat" --output="$T/co" --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$T/co" | RE="$RE" python3 -c "
import sys,re,os,subprocess
txt=sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt):
    m=re.search(r'\.name:\s+(\S+)',blk)
    if not m or not re.search(os.environ['RE'], m.group(1)): continue
    blk='.agpr_count'+blk
    g=lambda k:(re.search(re.escape(k)+r':\s+(\d+)',blk) or [None,'?'])[1]
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    print(f\"vgpr {g('.vgpr_count'):>4} agpr {g('.agpr_count'):>4} sgpr {g('.sgpr_count'):>4} scratch {g('.private_segment_fixed_size'):>5} lds {g('.group_segment_fixed_size'):>6}  {name[:160]}\")
"
done
rm -rf "$T"
