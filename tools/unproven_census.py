"""What the join-block rule (csrc/isacheck.cpp, tools/isa_check.py) cannot classify, looked at over the whole kernel cache.

The rule flags `vector instruction ... s_or_b64 exec, exec, s[a:b]` inside one basic block only when the block is PROVEN to be the
join block of the region that restore closes.  Every other such block is "unproven" (rh_code_object_report's count;
tests/golden/unproven_census.json holds the ceilings the CPU suite enforces).  This tool walks llvm-objdump's text of every cached
code object and sorts the unproven blocks by what precedes the restore, so that what hides behind the count is on record:

  nested     the block opens with the exec restore of ANOTHER pair (it IS a proven join block -- of an inner region) and then runs the
             rest of an outer region's body up to that region's own restore: the vector instructions belong to the outer body
  arm        the pair restored was saved by an `s_or_saveexec_b64` / `s_and_saveexec_b64` / `s_xor_b64 exec` earlier in the same
             FUNCTION with no label of that name as an execz target: an if / else arm without a skip branch that ends in its restore
  arm-ool    the block is the target of the `s_cbranch_execnz` that follows a saveexec into the SAME pair: a then-arm the block placement
             laid out of line (typically behind s_endpgm): its vector instructions are the arm's body, the restore ends the arm
  loop-exit  the block is the fall-through of a back edge (`s_cbranch_execnz` / `s_cbranch_scc*` / `s_cbranch_vcc*` to an earlier label)
  other      none of the above (listed one by one)

and by the vector instruction that stands immediately ahead of the restore: a register copy (`v_mov vA, vB`), an AGPR copy
(`v_accvgpr_*`) or a scratch access there is what the allocator's fault would ALSO leave (a copy meant for the join block, placed
ahead of the restore) -- and what a then-arm legitimately ends with (the copy of the value the arm defines into the merged
register, for the arm's lanes).  The two cannot be told apart from the text; the table says how many there are in kernels the
engine launches, which is what the device-side checks (create-time self-check on ragged prefixes, the parity tiers) stand for.

usage: python tools/unproven_census.py [--jobs N] > profiles/r5_parity/unproven_blocks.md"""
import collections
import os
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import re
import sys
from concurrent.futures import ProcessPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isa_check as I
from rainier_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KCACHE = os.path.join(ROOT, "rainier_amd", "kcache")
_RESTORE = re.compile(r"^s_or_b64 exec, exec, (s\[\d+:\d+\])")
_BACK = re.compile(r"^s_cbranch_(execnz|scc0|scc1|vccz|vccnz) (\S+)")


_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_COPY = re.compile(r"^(v_mov_b(32|64)(_e32|_e64)?|v_accvgpr_write_b32|v_accvgpr_mov_b32|v_accvgpr_read_b32)\s")


def regs_of(operand):
    """'v[82:83]' -> {('v', 82), ('v', 83)}; 'a35' -> {('a', 35)}; anything else -> empty"""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def arm_value_witness(body, at):
    """The trailing run of register / AGPR copies ahead of the exec restore at body[at]: is every copied VALUE defined by the block itself?

    What the allocator's fault leaves there is a copy of a value that is live ACROSS the region (defined before it, meant for every lane,
    executed under the arm's mask).  What an arm legitimately ends with is the copy of a value the ARM computed into the register the join
    reads -- its source is written, by something that is not itself a copy from outside, earlier in the same block.  Returns
    (copies in the run, copies whose source the block defines); a run with AGPR -> VGPR reads (reloads) counts them as not witnessed."""
    run = []
    k = at - 1
    while k >= 0 and (body[k].startswith(("s_waitcnt", "s_nop")) or _COPY.match(body[k])):
        if _COPY.match(body[k]):
            run.append(k)
        k -= 1
    defined = set()      # registers written by non-copy vector instructions (or loads) of this block, in program order up to the run
    for ins in body[:k + 1]:
        if ins.startswith(("v_cmp", "v_cmpx")) or not I._VECTOR.match(ins) or I._EXEC_IGNORING.match(ins):
            continue
        if ins.startswith(("ds_write", "global_store", "flat_store", "scratch_store", "buffer_store", "ds_bpermute", "ds_permute")) and not ins.startswith(("ds_bpermute", "ds_permute")):
            continue
        ops = ins.split(None, 1)
        if len(ops) < 2:
            continue
        dest = ops[1].split(",")[0]
        if _COPY.match(ins):
            src = ops[1].split(",", 1)[1] if "," in ops[1] else ""
            if regs_of(src) and regs_of(src) <= defined:     # a copy of a value the block defined passes the definition on
                defined |= regs_of(dest)
            else:
                defined -= regs_of(dest)
            continue
        defined |= regs_of(dest)
    ok = 0
    for j in sorted(run):
        ins = body[j]
        ops = ins.split(None, 1)[1]
        dest, src = ops.split(",", 1)
        if ins.startswith("v_accvgpr_read"):
            continue
        s = regs_of(src)
        if not s and re.match(r"^\s*(-?\d|0x|-?\d*\.\d)", src.strip()):   # an immediate: the arm's own constant
            ok += 1; defined |= regs_of(dest); continue
        if s and s <= defined:
            ok += 1; defined |= regs_of(dest)
    return len(run), ok


def blocks_of(lines):
    blocks, kernel = [], None
    for ln in lines:
        m = I._LABEL.match(ln)
        if m:
            name = m.group(1)
            if not re.match(r"^L\d+$", name):
                kernel = name
            blocks.append((kernel, name, []))
            continue
        if ln.startswith("\t") and blocks:
            ins = ln.strip().split("//")[0].strip()
            if ins:
                blocks[-1][2].append(ins)
    return blocks


def one(path):
    lines = I.disassemble(path)
    proven = {(k, lab) for k, lab, _, _ in I.check_lines(lines, strict=False)}
    blocks = blocks_of(lines)
    order = {(k, lab): i for i, (k, lab, _) in enumerate(blocks)}
    ool = set()          # (kernel, label, pair): label is the execnz target behind a saveexec into pair
    for kern, lab, body in blocks:
        for k, ins in enumerate(body):
            m = re.match(r"^s_cbranch_execnz (\S+)", ins)
            if m:
                for back in body[max(0, k - 4):k]:
                    w = re.match(r"^s_\w+_saveexec_b64 (s\[\d+:\d+\])", back)
                    if w:
                        ool.add((kern, m.group(1), w.group(1)))
    fit = {k: v["fit"] for (_, k), v in _capi.code_object_report(open(path, "rb").read()).items()}
    out = []
    for bi, (kern, lab, body) in enumerate(blocks):
        first_vec, opened_by_restore, seen_any = None, None, False
        for at, ins in enumerate(body):
            m = _RESTORE.match(ins)
            if m:
                if first_vec is None:
                    if not seen_any:
                        opened_by_restore = m.group(1)
                    seen_any = True
                    continue
                if (kern, lab) in proven:
                    break
                pair = m.group(1)
                if opened_by_restore and opened_by_restore != pair:
                    cls = "nested"
                elif (kern, lab, pair) in ool:
                    cls = "arm-ool"
                else:
                    prev = blocks[bi - 1][2] if bi and blocks[bi - 1][0] == kern else []
                    back = _BACK.match(prev[-1]) if prev else None
                    if back and order.get((kern, back.group(2)), 1 << 30) < bi:
                        cls = "loop-exit"
                    else:
                        saved = any(re.match(r"^s_\w+_saveexec_b64 %s\b" % re.escape(pair), i2) or
                                    re.match(r"^s_(xor|and|andn2)_b64 %s, .*exec" % re.escape(pair), i2) or
                                    re.match(r"^s_mov_b64 %s, exec" % re.escape(pair), i2)
                                    for k2, _, b2 in blocks[:bi + 1] if k2 == kern for i2 in b2)
                        cls = "arm" if saved else "other"
                last = next((b for b in reversed(body[:at]) if not b.startswith(("s_waitcnt", "s_nop"))), "")
                kind = ("copy" if re.match(r"^v_mov_b(32|64)(_e32|_e64)? v\[?\d+(:\d+)?\]?, v", last) else
                        "agpr" if "accvgpr" in last else "scratch" if last.startswith("scratch_") else "-")
                ncopy, nwit = arm_value_witness(body, at) if kind in ("copy", "agpr") else (0, 0)
                out.append((os.path.basename(path), kern, lab, cls, first_vec, ins, kind, fit.get(kern, 0), ncopy, nwit))
                break
            seen_any = True
            if I.writes_exec(ins):
                break
            if I._VECTOR.match(ins) and not I._EXEC_IGNORING.match(ins) and not ins.startswith("v_cmp") and first_vec is None:
                first_vec = ins
    return out


def main(argv):
    jobs = int(argv[argv.index("--jobs") + 1]) if "--jobs" in argv else 8
    files = sorted(os.path.join(KCACHE, f) for f in os.listdir(KCACHE) if f.endswith(".hsaco"))
    with ProcessPoolExecutor(jobs) as ex:
        rows = [r for rs in ex.map(one, files, chunksize=8) for r in rs]
    by = collections.Counter((r[1], r[3]) for r in rows)
    kernels = sorted({k for k, _ in by})
    classes = ["nested", "arm-ool", "arm", "loop-exit", "other"]
    print("# Exec restores behind vector instructions in blocks the join-block rule cannot prove (`tools/unproven_census.py`)\n")
    print("%d code objects of `rainier_amd/kcache`, %d such blocks (first one per block counted, as the rule does).\n" % (len(files), len(rows)))
    print("| kernel | " + " | ".join(classes) + " |\n|---|" + "---|" * len(classes))
    for k in kernels:
        print("| `%s` | " % k + " | ".join(str(by.get((k, c), 0)) for c in classes) + " |")
    others = [r for r in rows if r[3] == "other"]
    if others:
        print("\n## `other`, one by one\n")
        for f, k, lab, _, v, r, *_ in others[:60]:
            print("* `%s` `%s` <%s>: `%s` ahead of `%s`" % (f[:16], k, lab, v, r))
    tail = collections.Counter((r[1], r[6], r[7]) for r in rows)
    print("\n## The instruction immediately ahead of the restore (kernels the engine launches / kernels it refuses)\n")
    kinds = ["-", "copy", "agpr", "scratch"]
    print("| kernel | " + " | ".join("arm's own code" if k == "-" else k for k in kinds) + " |\n|---|" + "---|" * len(kinds))
    for k in kernels:
        print("| `%s` | " % k + " | ".join("%d / %d" % (tail.get((k, kd, 1), 0), tail.get((k, kd, 0), 0)) for kd in kinds) + " |")
    print("\n## Register / AGPR copies ahead of the restore: is the copied value the arm's own?\n")
    print("A copy there is what the allocator's fault would leave (a value live across the region, copied for every lane under the arm's mask) "
          "and what an arm legitimately ends with (the value IT computed, copied into the register the join reads).  Witness: every source of the "
          "trailing copy run is written earlier in the same block by a non-copy instruction (`arm_value_witness`).\n")
    print("| kernel (launched) | blocks ending in a copy / AGPR copy | every copy witnessed | not witnessed: single-block arm | not witnessed: nested (may copy an inner region's result) |\n|---|---|---|---|---|")
    unw = []
    for k in kernels:
        rs = [r for r in rows if r[1] == k and r[6] in ("copy", "agpr") and r[7] == 1]
        if not rs:
            continue
        good = [r for r in rs if r[8] > 0 and r[9] == r[8]]
        bad = [r for r in rs if not (r[8] > 0 and r[9] == r[8])]
        unw += bad
        print("| `%s` | %d | %d | %d | %d |" % (k, len(rs), len(good), len([r for r in bad if r[3] != "nested"]), len([r for r in bad if r[3] == "nested"])))
    if unw:
        print("\n### not witnessed, one by one\n")
        for r in unw[:40]:
            print("* `%s` `%s` <%s> (%s): %d of %d copies witnessed" % (r[0][:16], r[1], r[2], r[3], r[9], r[8]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
