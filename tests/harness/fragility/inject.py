#!/usr/bin/env python3
"""Investigation tool: makes the solve kernel of a .loc-annotated device .s stop at a chosen basic block and dump its LDS to P.dump.
usage: inject.py in.s out.s <src_lo> <src_hi> [nth=1] [lds_bytes=40960]
Injection point = the nth basic-block label of qp_solve_kernel whose first .loc (file 0) lies in [src_lo, src_hi]."""
import os, re, sys
src, dst, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
nth = int(sys.argv[5]) if len(sys.argv) > 5 else 1
nbytes = int(sys.argv[6]) if len(sys.argv) > 6 else 40960
L = open(src).read().split("\n")
start = next(i for i, l in enumerate(L) if l.startswith("_ZN") and "qp_solve_kernel" in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
target = None; cnt = 0
for i in range(start, end):
    if re.match(r"^\.LBB\d+_\d+:", L[i]):
        for j in range(i + 1, min(i + 40, end)):
            m = re.match(r"\s*\.loc\s+0\s+(\d+)\s", L[j])
            if m:
                if int(m.group(1)) == 0: continue
                if lo <= int(m.group(1)) <= hi:
                    cnt += 1
                    if cnt == nth: target = i
                break
            if re.match(r"^\.LBB\d+_\d+:", L[j]): break
    if target is not None: break
assert target is not None, "no block found"
nbar = int(os.environ.get("INJ_BARRIER", "0"))   # move the injection point to just after the n-th wave barrier that follows the block label
if nbar:
    c = 0
    for i in range(target, end):
        if "wave barrier" in L[i]:
            c += 1
            if c == nbar: target = i; break
    assert c == nbar
# register image: dword r of lane l at byte 65536 + 256 r + 4 l, r = 0..255 arch VGPRs, 256..511 accumulation VGPRs
rd = ["\ts_mov_b64 exec, -1", "\ts_nop 15", "\ts_nop 15", "\tscratch_store_dword off, v0, off offset:1024", "\tscratch_store_dword off, v1, off offset:1028", "\ts_waitcnt vmcnt(0)",
      "\tv_mbcnt_lo_u32_b32 v0, -1, 0", "\tv_mbcnt_hi_u32_b32 v0, -1, v0", "\tv_lshlrev_b32_e32 v0, 2, v0",
      "\ts_load_dwordx2 s[2:3], s[100:101], 0x160", "\ts_waitcnt lgkmcnt(0)", "\ts_add_u32 s2, s2, 0x10000", "\ts_addc_u32 s3, s3, 0"]
for r in range(512):
    if r == 0: rd += ["\tscratch_load_dword v1, off, off offset:1024", "\ts_waitcnt vmcnt(0)"]; srcr = "v1"
    elif r == 1: rd += ["\tscratch_load_dword v1, off, off offset:1028", "\ts_waitcnt vmcnt(0)"]; srcr = "v1"
    elif r < 256: srcr = "v%d" % r
    else: rd += ["\tv_accvgpr_read_b32 v1, a%d" % (r - 256), "\ts_nop 1"]; srcr = "v1"
    rd += ["\tglobal_store_dword v0, %s, s[2:3]" % srcr, "\ts_add_u32 s2, s2, 0x100", "\ts_addc_u32 s3, s3, 0"]
    if srcr == "v1": rd += ["\ts_waitcnt vmcnt(0)"]
for kk in range(160):   # scratch image (spill slots): dword kk of every lane behind the registers
    rd += ["\tscratch_load_dword v1, off, off offset:%d" % (4 * kk), "\ts_waitcnt vmcnt(0)", "\tglobal_store_dword v0, v1, s[2:3]", "\ts_add_u32 s2, s2, 0x100", "\ts_addc_u32 s3, s3, 0", "\ts_waitcnt vmcnt(0)"]
rd += ["\ts_waitcnt vmcnt(0)"]
regdump = "\n".join(rd) + "\n" if len(sys.argv) > 7 and sys.argv[7] == "regs" else ""
dump = """\ts_mov_b64 exec, -1
	s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)
	s_load_dwordx2 s[2:3], s[100:101], 0x160
	v_mbcnt_lo_u32_b32 v0, -1, 0
	v_mbcnt_hi_u32_b32 v0, -1, v0
	v_lshlrev_b32_e32 v0, 3, v0
	s_mov_b32 s4, 0
	s_waitcnt lgkmcnt(0)
.Linj_loop:
	v_add_u32_e32 v1, s4, v0
	ds_read_b64 v[2:3], v1
	s_waitcnt lgkmcnt(0)
	global_store_dwordx2 v1, v[2:3], s[2:3]
	s_add_u32 s4, s4, 0x200
	s_cmp_lt_u32 s4, %d
	s_cbranch_scc1 .Linj_loop
	s_waitcnt vmcnt(0)
%s	s_endpgm""" % (nbytes, regdump)
nm = int(os.environ.get("INJ_MFMA", "0"))   # move the injection point to just before the n-th v_mfma that follows the block label
if nm:
    c = 0
    for i in range(target, end):
        if L[i].strip().startswith("v_mfma"):
            c += 1
            if c == nm: target = i - 1; break
    assert c == nm
    print("mfma-anchored injection before line", target + 2)
jump = ["\ts_getpc_b64 s[2:3]", ".Linj_post:", "\ts_add_u32 s2, s2, (.Linj_dump-.Linj_post)&4294967295", "\ts_addc_u32 s3, s3, (.Linj_dump-.Linj_post)>>32", "\ts_setpc_b64 s[2:3]"]
out = L[:target + 1] + jump + L[target + 1:end] + [".Linj_dump:"] + dump.split("\n") + L[end:]
# keep the kernarg pointer in s[100:101] (unused by the kernel) from the first instruction on
k = next(i for i in range(start, len(out)) if out[i].startswith("; %bb.0:"))
ins = k + 1
while out[ins].lstrip().startswith(".cfi"): ins += 1
out.insert(ins, "\ts_mov_b64 s[100:101], s[0:1]")
txt = "\n".join(out)
assert "s100" not in "\n".join(L[start:end]) and "s[100:101]" not in "\n".join(L[start:end])
txt = txt.replace(".amdhsa_next_free_sgpr 100", ".amdhsa_next_free_sgpr 102")
if regdump:
    txt = re.sub(r"\.amdhsa_private_segment_fixed_size (\d+)", lambda m: ".amdhsa_private_segment_fixed_size 2048", txt)
    txt = re.sub(r"\.private_segment_fixed_size: (\d+)", lambda m: ".private_segment_fixed_size: 2048", txt)
open(dst, "w").write(txt)
ctx = [l for l in L[target:target + 12]]
print("injected at", target + 1, L[target], "| following:", [l.strip() for l in ctx[1:8]])
