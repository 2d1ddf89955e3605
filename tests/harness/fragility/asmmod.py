#!/usr/bin/env python3
"""Investigation tool: rewrites the assembly of qp_solve_kernel in a device .s file.  usage: asmmod.py in.s out.s mode[,mode...]
modes: mfma_after, mfma_before, wait_mem (full s_waitcnt after every memory instruction), nop_all (s_nop 7 after every instruction),
lane (s_nop 7 after readlane/writelane/readfirstlane/dpp), spill (full waitcnt + nops around scratch_ and v_writelane/v_readlane)"""
import re, sys
src, dst, modes = sys.argv[1], sys.argv[2], set(sys.argv[3].split(","))
lines = open(src).read().split("\n")
out = []
inside = False
guard = 0
n = {}
def add(key, txt):
    out.append("\t" + txt); n[key] = n.get(key, 0) + 1
for ln in lines:
    if re.match(r"^_ZN.*qp_solve_kernel.*:", ln): inside = True
    if ln.startswith(".Lfunc_end") : inside = False
    st = ln.strip()
    is_inst = inside and ln.startswith("\t") and st and not st.startswith((".", ";", "//")) and not st.endswith(":")
    op = st.split()[0] if is_inst else ""
    if is_inst and guard == 0:
        if "mfma_before" in modes and op.startswith("v_mfma"):
            add("mb", "s_nop 15"); add("mb", "s_nop 15")
    out.append(ln)
    if not is_inst: continue
    if op == "s_getpc_b64": guard = 3
    if guard > 0:
        guard -= 1
        continue
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")): continue
    if "mfma_after" in modes and op.startswith("v_mfma"):
        add("ma", "s_nop 15"); add("ma", "s_nop 15")
    if "wait_mem" in modes and op.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_", "s_load", "s_buffer_load", "s_store")):
        add("wm", "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)")
    if "lane" in modes and (op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) or "dpp" in st or "row_" in st):
        add("ln", "s_nop 7")
    if "spill" in modes and (op.startswith(("scratch_", "v_writelane", "v_readlane"))):
        add("sp", "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"); add("sp", "s_nop 7")
    if "nop_all" in modes:
        add("na", "s_nop 7")
open(dst, "w").write("\n".join(out))
print(dst, n)
