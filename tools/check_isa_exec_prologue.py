#!/usr/bin/env python3
"""Static check of generated gfx950 assembly for one compiler defect ("Build-variant fragility: root cause").

At the end of a divergent region the compiler restores the execution mask with `s_or_b64 exec, exec, s[a:b]` as the first
instruction of the join block.  Register-allocator copies and spill code placed at the top of that block must come AFTER the
restore; the ROCm 7.2 compiler sometimes emits a vector instruction (seen: `v_accvgpr_write_b32 aN, vM`, the VGPR->AGPR copy of a
live-range split) BEFORE it, so the copy runs under the narrowed mask of the region that just ended and the lanes that sat the
region out keep stale data.  This scan flags every vector instruction (VALU / LDS / vector memory; v_readlane / v_writelane /
v_readfirstlane ignore EXEC and are exempt) between a basic-block label and a mask-restoring s_or_b64 / s_mov_b64 of exec that
follows it in the same block, in blocks that a skip branch (s_cbranch_execz) targets, when the instruction is allocator-made
(AGPR copy or scratch spill / reload).
usage: check_isa_exec_prologue.py [--fix] file.s [...]   (device assembly: hipcc -S --cuda-device-only)
Without --fix the exit code is 1 if anything is flagged.  With --fix ONLY the pattern that was proven on hardware is repaired
(DESIGN.md 5c): `v_accvgpr_write_b32 aN, vM` copies sitting directly ahead of the restore -- nothing but such copies, exempt lane
reads and scalar non-wait instructions between the block label and the restore, and no copy's destination read in that prologue.
They are moved to just behind the restore (they touch no scalar state, the restore touches no vector state; with no other vector
instruction and no s_waitcnt in the prologue no dependence or wait is crossed).  Every other flagged shape -- spill loads / stores,
AGPR reads, an s_waitcnt or another vector instruction in the prologue -- is NOT touched: the exit code is 2 and the build stops,
so that a person looks at it.  The build (tools/hipcc_checked.sh) runs this on every translation unit and keeps the report next to
the object."""
import re
import sys

EXEMPT = ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")
VEC = ("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")


SPILLISH = ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32", "scratch_store", "scratch_load")


def scan(path):
    text = open(path).read()
    skip_targets = set(re.findall(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", text))   # blocks that can be entered with EXEC = 0
    hits = []
    func = None
    pend = []          # vector instructions seen since the last label in the current block
    opened = set()     # mask registers saved (regions opened) since the last label: their restores close regions of this block
    label = None
    for no, raw in enumerate(open(path), 1):
        s = raw.split(";")[0].rstrip()
        st = s.strip()
        if not st:
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:", s):     # label (block or function)
            if not st.startswith(".L"):
                func = st[:-1]
            if st.startswith(".LBB") or not st.startswith(".L"):
                label, pend, opened = st[:-1], [], set()
            continue
        if st.startswith("."):
            continue
        op = st.split()[0]
        if label is None:
            continue
        if op.startswith(VEC) and op not in EXEMPT:
            pend.append((no, st))
            continue
        if op.startswith("s_waitcnt") and label is not None:
            pend.append((no, st))          # a wait in the prologue: anything hoisted across it loses its wait (never auto-repaired)
            continue
        m = re.match(r"^s_(?:and|or|xor|andn2|orn2)_saveexec_b64\s+(s\[\d+:\d+\]|vcc)", st) or re.match(r"^s_mov_b64\s+(s\[\d+:\d+\]|vcc)\s*,\s*exec", st)
        if m:
            opened.add(m.group(1))
            continue
        m = re.match(r"^s_or_b64\s+exec\s*,\s*exec\s*,\s*(s\[\d+:\d+\]|vcc)", st)
        if m:
            if m.group(1) in opened:
                opened.discard(m.group(1))      # closes a region opened inside this block
            else:
                # restore of a region opened before this block = the block's prologue.  In a block that the region's skip
                # branch targets, nothing the register allocator adds (AGPR copies, spill code) may come before it.
                moved = [x for x in pend if x[1].split()[0].startswith(SPILLISH)]
                if moved and label in skip_targets:
                    hits.append((func, label, no, st, moved, list(pend)))
                label = None
            continue
        if re.match(r"^s_\w+\s+exec\s*,", st):  # any other write of exec ends the prologue
            label = None
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            label = None
            continue
    return hits


def safe_to_move(hit):
    """The proven pattern only: every vector instruction of the prologue is a `v_accvgpr_write_b32 aN, vM` (so nothing in the
    prologue can consume what is moved, and nothing else is left ahead of the restore under the stale mask), there is no s_waitcnt
    in the prologue, and no copy reads an AGPR another copy of the prologue writes."""
    moved, prologue = hit[4], hit[5]
    if len(moved) != len(prologue):
        return False
    dests = set()
    for _, ins in prologue:
        m = re.match(r"^v_accvgpr_write_b32\s+(a\d+)\s*,\s*(v\d+)\s*$", ins)
        if not m or m.group(1) in dests:
            return False
        dests.add(m.group(1))
    return True


def repair(path, hits):
    lines = open(path).read().split("\n")
    for func, label, no, st, moved, _prologue in sorted(hits, key=lambda h: -h[2]):   # bottom-up keeps the line numbers valid
        take = [lines[pno - 1] for pno, _ in moved]
        for pno, _ in sorted(moved, reverse=True):
            del lines[pno - 1]
        at = no - len(moved)                 # index just behind the restore after the deletions (no is 1-based)
        lines[at:at] = take
    open(path, "w").write("\n".join(lines))


def main():
    args = sys.argv[1:]
    fix = "--fix" in args
    bad = 0
    for path in (a for a in args if a != "--fix"):
        hits = scan(path)
        print("%s: %d allocator-made instruction group(s) ahead of an exec restore%s" % (path, len(hits), " -- moved behind it" if (fix and hits) else ""))
        unsafe = [h for h in hits if not safe_to_move(h)]
        for h in hits:
            func, label, no, st, pend = h[:5]
            bad += 1
            print("  %s  block %s: `%s` (line %d) is preceded by%s" % (func[:60], label, st, no, "" if safe_to_move(h) else "  [NOT the proven pattern: left as is]"))
            for pno, pst in h[5][:6]:
                print("      line %d: %s" % (pno, pst))
        if fix and hits:
            if unsafe:
                print("  %d group(s) are not plain v_accvgpr_write copies directly ahead of the restore: no automatic repair, the build stops here" % len(unsafe))
                return 2
            repair(path, hits)
            assert not scan(path), "repair left something behind in " + path
    return 1 if (bad and not fix) else 0


if __name__ == "__main__":
    sys.exit(main())
