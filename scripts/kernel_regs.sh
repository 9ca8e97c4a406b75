# kernel_regs.sh <object with a HIP fat binary> [grep pattern]: VGPR / SGPR / scratch / LDS of every gfx950 kernel in it (code-object metadata)
set -e
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb $1
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb --output=$T/co --unbundle
$L/llvm-readelf --notes $T/co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    print('%4s vgpr %4s sgpr %5s scratch  %s'%(m.group(4),m.group(3),m.group(2),m.group(1)))
" | c++filt | grep -E "${2:-.}" || true
rm -rf $T
