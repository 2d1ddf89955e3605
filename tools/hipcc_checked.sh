#!/bin/bash
# Compiles one HIP translation unit for gfx950 with the device code going through assembly text, so that the assembly can be
# checked -- and repaired -- for the compiler defect described in DESIGN.md ("Build-variant fragility: root cause") before it
# is assembled:  hipcc -S (device) -> tools/check_isa_exec_prologue.py --fix -> assembler -> code object -> bundle -> host
# compile with that bundle embedded.  The report stays next to the object as <obj>.isa.log; KEEP_ISA=1 keeps the assembly too.
# usage: hipcc_checked.sh <out.o> <src.hip> <hipcc flags...>
set -e
out=$1; src=$2; shift 2
here=$(cd "$(dirname "$0")" && pwd)
B=${ROCM_LLVM_BIN:-/opt/rocm/lib/llvm/bin}
HIPCC=${HIPCC:-hipcc}
t=${out%.o}
$HIPCC "$@" --cuda-device-only -S "$src" -o "$t.s"
python3 "$here/check_isa_exec_prologue.py" --fix "$t.s" > "$t.isa.log"
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$t.s" -o "$t.dev.o"
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$t.hsaco" "$t.dev.o"
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input="$t.hsaco" -output="$t.hipfb"
$HIPCC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$t.hipfb" -c "$src" -o "$out"
rm -f "$t.dev.o" "$t.hsaco" "$t.hipfb"
[ -n "$KEEP_ISA" ] || rm -f "$t.s"
head -1 "$t.isa.log"
