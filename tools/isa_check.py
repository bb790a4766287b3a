"""Static check of gfx950 code objects for a register-allocator fault of this toolchain (ROCm 7.2 comgr / LLVM).

The fault (root cause of the wrong sums / wrong draws / faults of kernels that spill): at a control-flow JOIN block
the exec mask is restored by `s_or_b64 exec, exec, s[a:b]` (SI_END_CF), which must be the first thing the block does.  The register
allocator inserts spill stores, reloads and live-range-split copies "at the top of the block, after the prologue"
(MachineBasicBlock::SkipPHIsLabelsAndDebug -> SIInstrInfo::isBasicBlockPrologue), but the prologue scan stops at the first
instruction that is not a prologue instruction -- and a scalar rematerialisation (`s_mov_b32 s62, 0`, legal before the restore)
placed there first makes it stop BEFORE the exec restore.  Every vector instruction inserted afterwards lands ahead of the restore
and runs under the mask of the region that just ended: lanes that skipped the region keep stale registers, spill slots of those
lanes are never written.  Seen in rh_chain_kernel of hier_negbin(6, 7): the 6-row prior target's loop exit spills four chain-state
vectors with 6 lanes active (profiles/r4_spill_rootcause/).

What is flagged, per basic block (label to label, labels = branch targets): a vector instruction that writes vector registers or
memory and precedes an `s_or_b64 exec, exec, <sgpr pair>` with no other write of EXEC in between in that block -- when the block is
proven to be the join block of the region that restore closes (see check_lines).  The engine applies the same rule to the machine
code itself before it launches a kernel (csrc/isacheck.cpp); this tool is the independent second statement of it, over
llvm-objdump's text, that the CPU suite compares the engine's verdicts with.

usage: python tools/isa_check.py [--strict] file.hsaco ... | --cache DIR
"""
import os
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
_VECTOR = re.compile(r"^(v_|ds_|scratch_|global_|flat_|buffer_|tbuffer_|image_)")
_EXEC_IGNORING = re.compile(r"^(v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)\b")   # v_readfirstlane reads EXEC but writes an SGPR
_END_CF = re.compile(r"^s_or_b64 exec, exec, s\[\d+:\d+\]")
_EXEC_WRITE = re.compile(r"^s_\w+ (exec\b|s\[\d+:\d+\], .*)")


def disassemble(path):
    out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", "--mcpu=gfx950", path],
                         check=True, capture_output=True, text=True).stdout
    return out.splitlines()


def writes_exec(ins):
    if ins.startswith("s_") and re.match(r"^s_\w+ exec\b", ins):
        return True
    return bool(re.match(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|andn1|orn1)\w*_saveexec_b64\b", ins))


def check_lines(lines, strict=False):
    """-> list of (kernel, label, first vector instruction, the exec restore it precedes).  A second statement of the rule
    csrc/isacheck.cpp applies (that one walks the machine code itself): the block must be PROVEN the join block of the region the
    restore closes -- the target of the s_cbranch_execz behind the saveexec into the same pair, or the fall-through of an
    s_cbranch_execnz back edge behind `s_andn2_b64 exec, exec, <pair>`; strict = every block (many false positives: region
    bodies that end in the restore)."""
    blocks = []          # (kernel, label, [instructions])
    kernel = None
    for ln in lines:
        m = _LABEL.match(ln)
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
    # witnesses: label -> set of pairs for which the label is a proven join block
    joins = {}
    for bi, (kern, lab, body) in enumerate(blocks):
        for k, ins in enumerate(body):
            m = re.match(r"^s_cbranch_execz (\S+)", ins)
            if m:
                for back in body[max(0, k - 4):k]:
                    w = re.match(r"^s_\w+ (s\[\d+:\d+\]), .*", back)
                    if w and ("exec" in back or "saveexec" in back):
                        joins.setdefault((kern, m.group(1)), set()).add(w.group(1))
            if re.match(r"^s_cbranch_execnz ", ins) and k == len(body) - 1 and k >= 1 and bi + 1 < len(blocks):
                w = re.match(r"^s_andn2_b64 exec, exec, (s\[\d+:\d+\])", body[k - 1])
                if w:
                    joins.setdefault((kern, blocks[bi + 1][1]), set()).add(w.group(1))
    bad = []
    for kern, lab, body in blocks:
        first_vec = None
        for ins in body:
            m = re.match(r"^s_or_b64 exec, exec, (s\[\d+:\d+\])", ins)
            if m:
                if first_vec is not None and (strict or m.group(1) in joins.get((kern, lab), ())):
                    bad.append((kern, lab, first_vec, ins))
                    break
                continue
            if writes_exec(ins):
                break
            if _VECTOR.match(ins) and not _EXEC_IGNORING.match(ins) and not ins.startswith("v_cmp") and first_vec is None:
                first_vec = ins
    return bad


def check_file(path, strict=False):
    return check_lines(disassemble(path), strict)


def main(argv):
    strict = "--strict" in argv
    argv = [a for a in argv if a != "--strict"]
    files = []
    if argv and argv[0] == "--cache":
        d = argv[1]
        files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hsaco"))
    else:
        files = argv
    nbad = 0
    for f in files:
        bad = check_file(f, strict)
        for k, lab, v, r in bad:
            print("%s  %s <%s>: `%s` ahead of `%s`" % (os.path.basename(f), k, lab, v, r))
        nbad += bool(bad)
    print("%d of %d code objects flagged" % (nbad, len(files)))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
