// isacheck.cpp -- what the engine reads out of a gfx950 code object before it agrees to launch it.
//
// (1) kernel_meta: the kernel's entry of the AMDGPU metadata note (msgpack): register counts, spill counts, scratch size.
//
// (2) check_code_object: a static check of the machine code for a fault of this toolchain's register allocator (ROCm 7.2 comgr),
//     the root cause of the wrong sums / wrong draws / faults of kernels that spill (profiles/r4_spill_rootcause):
//
//     At a control-flow join the exec mask is restored by `s_or_b64 exec, exec, s[a:b]` (SI_END_CF), which has to be the first
//     thing the join block does.  The allocator places spill stores, reloads and live-range-split copies "at the top of the block,
//     behind its prologue" (MachineBasicBlock::SkipPHIsLabelsAndDebug -> SIInstrInfo::isBasicBlockPrologue), and that scan stops
//     at the first instruction that is not a prologue instruction.  A scalar rematerialisation (`s_mov_b32 s62, 0`), legal ahead
//     of the restore and put there by the scalar allocation that runs first, makes it stop BEFORE the restore: every vector
//     instruction inserted afterwards runs under the mask of the region that just ended.  Lanes that skipped the region keep a
//     stale register, or their spill slot is never written.
//
//     What is flagged: an exec restore that is preceded, in its basic block, by a vector instruction that writes vector
//     registers or memory, with no other write of EXEC in between -- when the block is PROVEN to be the region's join block by
//     the way it is entered (the target of the header's s_cbranch_execz, or the fall-through of a divergent loop's back edge).
//     Seen in the wild as scratch stores (spills), v_accvgpr_write (the allocator's register-to-AGPR copies, which the metadata
//     does not count as spills) and v_mov (live-range splits).  The instruction walk needs only instruction LENGTHS and a handful
//     of opcodes of the GFX9 encodings; the CPU suite checks the walk against llvm-objdump on every code object build() leaves
//     in the kernel cache.
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "rir.hpp"

namespace rh {
namespace {

// ---- ELF64 little-endian, just enough -----------------------------------------------------------------
struct Sec { uint32_t name, type; uint64_t flags, addr, off, size; uint32_t link, info; uint64_t align, entsize; };
template <class T> bool rd(const std::vector<char> &b, uint64_t off, T &out) {
  if (off + sizeof(T) > b.size()) return false;
  std::memcpy(&out, b.data() + off, sizeof(T));
  return true;
}
bool sections(const std::vector<char> &b, std::vector<Sec> &out, std::vector<std::string> &names) {
  if (b.size() < 64 || std::memcmp(b.data(), "\177ELF", 4) != 0 || b[4] != 2 || b[5] != 1) return false;
  uint64_t shoff; uint16_t shentsize, shnum, shstrndx;
  if (!rd(b, 0x28, shoff) || !rd(b, 0x3A, shentsize) || !rd(b, 0x3C, shnum) || !rd(b, 0x3E, shstrndx)) return false;
  if (shentsize != 64) return false;
  out.clear();
  for (uint16_t i = 0; i < shnum; i++) {
    const uint64_t o = shoff + (uint64_t)i * 64;
    Sec s;
    if (!rd(b, o, s.name) || !rd(b, o + 4, s.type) || !rd(b, o + 8, s.flags) || !rd(b, o + 16, s.addr) || !rd(b, o + 24, s.off) ||
        !rd(b, o + 32, s.size) || !rd(b, o + 40, s.link) || !rd(b, o + 44, s.info) || !rd(b, o + 48, s.align) || !rd(b, o + 56, s.entsize))
      return false;
    out.push_back(s);
  }
  if (shstrndx >= out.size()) return false;
  const Sec &st = out[shstrndx];
  names.clear();
  for (const Sec &s : out) {
    std::string n;
    for (uint64_t p = st.off + s.name; p < b.size() && p < st.off + st.size && b[p]; p++) n.push_back(b[p]);
    names.push_back(n);
  }
  return true;
}

// ---- GFX9 (gfx940 / gfx950) instruction walk --------------------------------------------------------------
// K_VEC: a vector instruction that writes vector registers or memory under EXEC; K_VEC_NOEXEC: one that does not (lane access
// instructions, compares into VCC / an SGPR pair -- the structurizer's own loop-exit conditions legitimately precede a restore)
enum Kind : uint8_t { K_SALU, K_SMEM, K_VEC, K_VEC_NOEXEC, K_BRANCH, K_CBRANCH, K_END, K_INDIRECT };
struct Ins {
  uint32_t off; uint8_t len; Kind kind;
  bool writes_exec = false;     // any write of EXEC (sdst = exec_lo / exec_hi, the saveexec family, v_cmpx)
  bool end_cf = false;          // s_or_b64 exec, exec, s[a:b]
  bool reads_exec = false;      // scalar instruction with EXEC among its sources
  int sdst = -1;                // scalar destination register (first of a pair)
  int pair = -1;                // end_cf: the SGPR pair a
  int src_pair = -1;            // SOP2 with EXEC as destination and as one source: the other source (an SGPR pair)
  int64_t target = -1;          // branch target (byte offset in the section)
  uint8_t bop = 0;              // SOPP opcode of a branch (8 = s_cbranch_execz, 9 = s_cbranch_execnz)
  bool scratch = false;         // an access to private (scratch) memory
};

bool decode(const unsigned char *p, size_t left, uint32_t off, Ins &I) {
  if (left < 4) return false;
  uint32_t w;
  std::memcpy(&w, p, 4);
  I = Ins();
  I.off = off; I.len = 4; I.kind = K_SALU;
  auto lit = [&](unsigned src) { return src == 255; };
  if ((w >> 23) == 0x17F) {                       // SOPP
    const unsigned op = (w >> 16) & 0x7F;
    const int16_t simm = (int16_t)(w & 0xFFFF);
    I.bop = (uint8_t)op;
    if (op == 1) I.kind = K_END;                  // s_endpgm
    else if (op == 2) { I.kind = K_BRANCH; I.target = (int64_t)off + 4 + (int64_t)simm * 4; }
    else if (op >= 4 && op <= 9) { I.kind = K_CBRANCH; I.target = (int64_t)off + 4 + (int64_t)simm * 4; }
    else if (op >= 23 && op <= 26) { I.kind = K_CBRANCH; I.target = (int64_t)off + 4 + (int64_t)simm * 4; }  // s_cbranch_cdbg*
    return true;
  }
  if ((w >> 23) == 0x17E) {                       // SOPC
    if (lit(w & 0xFF) || lit((w >> 8) & 0xFF)) I.len = 8;
    I.reads_exec = (w & 0xFF) == 126 || ((w >> 8) & 0xFF) == 126;
    return true;
  }
  if ((w >> 23) == 0x17D) {                       // SOP1
    const unsigned op = (w >> 8) & 0xFF, sdst = (w >> 16) & 0x7F, s0 = w & 0xFF;
    if (lit(s0)) I.len = 8;
    I.sdst = (int)sdst;
    I.reads_exec = s0 == 126 || s0 == 127;
    const bool saveexec = (op >= 32 && op <= 39) || (op >= 51 && op <= 54);   // s_*_saveexec_b64, s_andn1/orn1_saveexec, s_andn{1,2}_wrexec
    if (saveexec) { I.writes_exec = true; I.reads_exec = true; }
    if (sdst == 126 || sdst == 127) I.writes_exec = true;
    if (op == 29 || op == 31) I.kind = K_END;      // s_setpc_b64 (a function's return; a computed jump is never emitted inside our kernels), s_rfe_b64
    // (op 30, s_swappc_b64, is a CALL: control comes back to the next instruction; the callee is walked as a function of its own)
    return true;
  }
  if ((w >> 28) == 0xB) {                         // SOPK
    const unsigned op = (w >> 23) & 0x1F, sdst = (w >> 16) & 0x7F;
    if (op == 20) I.len = 8;                      // s_setreg_imm32_b32
    if (op == 21) I.kind = K_INDIRECT;            // s_call_b64
    I.sdst = (int)sdst;
    if ((sdst == 126 || sdst == 127) && op != 17 && op != 18 && op != 20 && !(op >= 2 && op <= 13)) I.writes_exec = true;
    return true;
  }
  if ((w >> 30) == 0x2) {                         // SOP2
    const unsigned op = (w >> 23) & 0x7F, sdst = (w >> 16) & 0x7F, s1 = (w >> 8) & 0xFF, s0 = w & 0xFF;
    if (lit(s0) || lit(s1)) I.len = 8;
    I.sdst = (int)sdst;
    I.reads_exec = s0 == 126 || s1 == 126 || s0 == 127 || s1 == 127;
    if (sdst == 126 || sdst == 127) I.writes_exec = true;
    if (sdst == 126 && (s0 == 126 || s1 == 126)) { const unsigned other = s0 == 126 ? s1 : s0; if (other <= 107) I.src_pair = (int)other; }
    if (op == 15 && sdst == 126) {                // s_or_b64 exec, ...
      const unsigned other = s0 == 126 ? s1 : (s1 == 126 ? s0 : 999);
      if (other <= 107) { I.end_cf = true; I.pair = (int)other; }   // exec | an SGPR pair (or vcc): the SI_END_CF form
    }
    return true;
  }
  if ((w >> 26) == 0x30) { I.kind = K_SMEM; I.len = 8; return true; }          // SMEM
  if ((w >> 31) == 0) {                           // VOP2 / VOP1 / VOPC
    const unsigned s0 = w & 0x1FF;
    I.kind = K_VEC;
    if (s0 == 255 || s0 == 249 || s0 == 250) I.len = 8;                        // literal, SDWA, DPP
    if ((w >> 25) == 0x3F) {                      // VOP1
      const unsigned op = (w >> 9) & 0xFF;
      if (op == 2) { I.kind = K_VEC_NOEXEC; I.sdst = (int)((w >> 17) & 0xFF); }  // v_readfirstlane_b32 (writes an SGPR)
    } else if ((w >> 25) == 0x3E) {               // VOPC: writes VCC / EXEC only -- the structurizer's own loop-exit conditions sit ahead of a restore
      const unsigned op = (w >> 17) & 0xFF;
      I.kind = K_VEC_NOEXEC;
      // v_cmpx_* write EXEC: class ops 0x11/0x13 (f32/f64 cmpx_class), 0x15 f16; compare ops with bit 4 set in each 0x20 block
      if (op == 0x11 || op == 0x13 || op == 0x15 || (op >= 0x30 && op <= 0x3F) || (op >= 0x50 && op <= 0x5F) || (op >= 0x70 && op <= 0x7F) ||
          (op >= 0xB0 && op <= 0xBF) || (op >= 0xD0 && op <= 0xDF) || (op >= 0xF0 && op <= 0xFF))
        I.writes_exec = true;
    } else {                                      // VOP2
      const unsigned op = (w >> 25) & 0x3F;
      if (op == 23 || op == 24 || op == 37 || op == 38) I.len = 8;              // v_{mad,fma}mk / ak: a mandatory literal
    }
    return true;
  }
  const unsigned enc = w >> 26;
  if (enc == 0x34) {                              // VOP3 / VOP3P (110100)
    I.len = 8; I.kind = K_VEC;
    const unsigned op = (w >> 16) & 0x3FF;
    if ((w >> 23) != 0x1A7) {                     // VOP3 proper (VOP3P = 110100111)
      if (op == 0x289 || op == 0x28A) I.kind = K_VEC_NOEXEC;                    // v_readlane_b32 / v_writelane_b32
      if (op == 0x142) I.kind = K_VEC_NOEXEC;                                   // v_readfirstlane_b32, VOP3 form
      if (op < 0x100) {                           // VOPC in VOP3 form: sdst in [7:0]
        const unsigned sd = w & 0xFF;
        I.kind = K_VEC_NOEXEC;
        if (sd == 126 || sd == 127) I.writes_exec = true;
        if (op == 0x11 || op == 0x13 || op == 0x15 || (op >= 0x30 && op <= 0x3F) || (op >= 0x50 && op <= 0x5F) || (op >= 0x70 && op <= 0x7F) ||
            (op >= 0xB0 && op <= 0xBF) || (op >= 0xD0 && op <= 0xDF) || (op >= 0xF0 && op <= 0xFF))
          I.writes_exec = true;
      }
    }
    return true;
  }
  if (enc == 0x36 || enc == 0x37 || enc == 0x38 || enc == 0x3A || enc == 0x3C || enc == 0x31) {   // DS, FLAT, MUBUF, MTBUF, MIMG, EXP
    I.len = 8; I.kind = K_VEC;
    if (enc == 0x37 && ((w >> 14) & 3) == 1) I.scratch = true;   // FLAT encoding, SEG = scratch: private memory (spills, local arrays)
    if (enc == 0x38) I.scratch = true;                            // MUBUF: the other way private memory is reached
    return true;
  }
  if (enc == 0x35) { I.kind = K_VEC; return true; }                              // VINTRP (not in compute code)
  return false;
}

struct Func { std::string name; uint64_t off, size; };

}  // namespace

bool kernel_meta(const std::vector<char> &code, const std::string &name, KernelMeta &out) {
  out = KernelMeta();
  auto find = [&](const std::string &needle, size_t from, size_t to) -> size_t {
    if (needle.size() > code.size()) return std::string::npos;
    to = std::min(to, code.size());
    for (size_t i = from; i + needle.size() <= to; i++)
      if (std::memcmp(code.data() + i, needle.data(), needle.size()) == 0) return i;
    return std::string::npos;
  };
  auto mstr = [](const std::string &v) {   // msgpack string header + bytes (fixstr | str8)
    std::string o;
    if (v.size() < 32) o.push_back((char)(0xa0 | v.size())); else { o.push_back((char)0xd9); o.push_back((char)v.size()); }
    return o + v;
  };
  // a kernel's map is written with its keys in alphabetical order: ".name" comes before ".private_segment_fixed_size",
  // ".sgpr_count", ".sgpr_spill_count", ".symbol", ".vgpr_count", ".vgpr_spill_count"; the next kernel's ".name" bounds the search
  const size_t at = find(mstr(".name") + mstr(name), 0, code.size());
  if (at == std::string::npos) return false;
  size_t next = find(mstr(".name"), at + 1, code.size());
  if (next == std::string::npos) next = code.size();
  auto num = [&](const char *key, long &v) {
    const std::string k = mstr(key);
    const size_t p = find(k, at, next);
    if (p == std::string::npos || p + k.size() >= code.size()) return false;
    const unsigned char *q = (const unsigned char *)code.data() + p + k.size();
    const size_t left = code.size() - (p + k.size());
    if (q[0] <= 0x7f) { v = q[0]; return true; }
    if (q[0] == 0xcc && left >= 2) { v = q[1]; return true; }
    if (q[0] == 0xcd && left >= 3) { v = (long)q[1] << 8 | q[2]; return true; }
    if (q[0] == 0xce && left >= 5) { v = (long)q[1] << 24 | (long)q[2] << 16 | (long)q[3] << 8 | q[4]; return true; }
    return false;
  };
  out.found = num(".vgpr_spill_count", out.vgpr_spills) && num(".sgpr_spill_count", out.sgpr_spills) && num(".vgpr_count", out.vgprs) &&
              num(".sgpr_count", out.sgprs) && num(".private_segment_fixed_size", out.scratch_bytes);
  return out.found;
}

bool list_kernels(const std::vector<char> &code, std::vector<std::string> &names, bool kernels_only) {
  std::vector<Sec> secs; std::vector<std::string> sn;
  names.clear();
  if (!sections(code, secs, sn)) return false;
  for (size_t i = 0; i < secs.size(); i++) {
    if (secs[i].type != 2 /*SHT_SYMTAB*/ || secs[i].link >= secs.size()) continue;
    const Sec &str = secs[secs[i].link];
    for (uint64_t o = secs[i].off; o + 24 <= secs[i].off + secs[i].size; o += 24) {
      uint32_t nm; unsigned char info; uint16_t shndx; uint64_t value, size;
      if (!rd(code, o, nm) || !rd(code, o + 4, info) || !rd(code, o + 6, shndx) || !rd(code, o + 8, value) || !rd(code, o + 16, size)) return false;
      if ((info & 0xF) != 2 /*STT_FUNC*/ || shndx >= secs.size() || sn[shndx] != ".text") continue;
      std::string n;
      for (uint64_t p = str.off + nm; p < code.size() && code[p]; p++) n.push_back(code[p]);
      names.push_back(n);
    }
  }
  if (!kernels_only) return true;
  // a kernel has a descriptor object "<name>.kd"; the other functions are device functions the kernels call (noinline)
  std::set<std::string> kd;
  for (size_t i = 0; i < secs.size(); i++) {
    if (secs[i].type != 2 || secs[i].link >= secs.size()) continue;
    const Sec &str = secs[secs[i].link];
    for (uint64_t o = secs[i].off; o + 24 <= secs[i].off + secs[i].size; o += 24) {
      uint32_t nm;
      if (!rd(code, o, nm)) return false;
      std::string n;
      for (uint64_t p = str.off + nm; p < code.size() && code[p]; p++) n.push_back(code[p]);
      if (n.size() > 3 && n.compare(n.size() - 3, 3, ".kd") == 0) kd.insert(n.substr(0, n.size() - 3));
    }
  }
  std::vector<std::string> out;
  for (const std::string &n : names) if (kd.count(n)) out.push_back(n);
  names.swap(out);
  return true;
}

// The instruction offsets (relative to the kernel's first byte) of kernel `name`, for the decoder's own test.
bool kernel_instruction_offsets(const std::vector<char> &code, const std::string &name, std::vector<uint32_t> &offs) {
  std::vector<Sec> secs; std::vector<std::string> sn;
  offs.clear();
  if (!sections(code, secs, sn)) return false;
  for (size_t i = 0; i < secs.size(); i++) {
    if (secs[i].type != 2 || secs[i].link >= secs.size()) continue;
    const Sec &str = secs[secs[i].link];
    for (uint64_t o = secs[i].off; o + 24 <= secs[i].off + secs[i].size; o += 24) {
      uint32_t nm; unsigned char info; uint16_t shndx; uint64_t value, size;
      if (!rd(code, o, nm) || !rd(code, o + 4, info) || !rd(code, o + 6, shndx) || !rd(code, o + 8, value) || !rd(code, o + 16, size)) return false;
      if ((info & 0xF) != 2 || shndx >= secs.size()) continue;
      std::string n;
      for (uint64_t p = str.off + nm; p < code.size() && code[p]; p++) n.push_back(code[p]);
      if (n != name) continue;
      const Sec &tx = secs[shndx];
      const uint64_t foff = tx.off + (value - tx.addr);
      if (foff + size > code.size()) return false;
      uint64_t p = 0;
      while (p < size) {
        Ins I;
        if (!decode((const unsigned char *)code.data() + foff + p, size - p, (uint32_t)p, I)) return false;
        offs.push_back((uint32_t)p);
        p += I.len;
      }
      return true;
    }
  }
  return false;
}

// Does the kernel's own code touch private (scratch) memory at all?  `.vgpr_spill_count` also counts the allocator's VGPR -> AGPR
// copies (they start life as spills and are turned into v_accvgpr_write later), so a non-zero count on a kernel WITHOUT a single
// scratch instruction is not a spill to memory -- and the copies themselves are what the join-block walk looks at.
// returns -1 when the kernel cannot be walked
int kernel_touches_scratch(const std::vector<char> &code, const std::string &name) {
  std::vector<Sec> secs; std::vector<std::string> sn;
  if (!sections(code, secs, sn)) return -1;
  for (size_t i = 0; i < secs.size(); i++) {
    if (secs[i].type != 2 || secs[i].link >= secs.size()) continue;
    const Sec &str = secs[secs[i].link];
    for (uint64_t o = secs[i].off; o + 24 <= secs[i].off + secs[i].size; o += 24) {
      uint32_t nm; unsigned char info; uint16_t shndx; uint64_t value, size;
      if (!rd(code, o, nm) || !rd(code, o + 4, info) || !rd(code, o + 6, shndx) || !rd(code, o + 8, value) || !rd(code, o + 16, size)) return -1;
      if ((info & 0xF) != 2 || shndx >= secs.size()) continue;
      std::string n;
      for (uint64_t p = str.off + nm; p < code.size() && code[p]; p++) n.push_back(code[p]);
      if (n != name) continue;
      const Sec &tx = secs[shndx];
      const uint64_t foff = tx.off + (value - tx.addr);
      if (foff + size > code.size()) return -1;
      for (uint64_t p = 0; p < size;) {
        Ins I;
        if (!decode((const unsigned char *)code.data() + foff + p, size - p, (uint32_t)p, I)) return -1;
        if (I.scratch) return 1;
        p += I.len;
      }
      return 0;
    }
  }
  return -1;
}

// -> findings, one line each: "<kernel>+0x<offset>: ..."; returns false when the code object cannot be walked at all
bool check_code_object(const std::vector<char> &code, const std::string &only_kernel, std::vector<std::string> &findings, int *unproven) {
  if (unproven) *unproven = 0;
  std::vector<Sec> secs; std::vector<std::string> sn;
  if (!sections(code, secs, sn)) { findings.push_back("not an ELF64 code object"); return false; }
  std::vector<Func> funcs;
  std::set<std::string> kernel_names;
  { std::vector<std::string> kn; if (list_kernels(code, kn, true)) kernel_names.insert(kn.begin(), kn.end()); }
  for (size_t i = 0; i < secs.size(); i++) {
    if (secs[i].type != 2 || secs[i].link >= secs.size()) continue;
    const Sec &str = secs[secs[i].link];
    for (uint64_t o = secs[i].off; o + 24 <= secs[i].off + secs[i].size; o += 24) {
      uint32_t nm; unsigned char info; uint16_t shndx; uint64_t value, size;
      if (!rd(code, o, nm) || !rd(code, o + 4, info) || !rd(code, o + 6, shndx) || !rd(code, o + 8, value) || !rd(code, o + 16, size)) return false;
      if ((info & 0xF) != 2 || shndx >= secs.size() || sn[shndx] != ".text") continue;
      std::string n;
      for (uint64_t p = str.off + nm; p < code.size() && code[p]; p++) n.push_back(code[p]);
      if (!only_kernel.empty() && n != only_kernel && kernel_names.count(n)) continue;   // (device functions: possible callees of any kernel)
      const Sec &tx = secs[shndx];
      funcs.push_back({n, tx.off + (value - tx.addr), size});
    }
  }
  if (funcs.empty() || (!only_kernel.empty() && !kernel_names.count(only_kernel))) {
    findings.push_back(only_kernel.empty() ? "no kernels in the code object" : "kernel " + only_kernel + " not found"); return false; }
  bool walked = true;
  char buf[256];
  for (const Func &f : funcs) {
    if (f.off + f.size > code.size()) { findings.push_back(f.name + ": symbol outside the file"); walked = false; continue; }
    std::vector<Ins> ins;
    uint64_t p = 0;
    bool ok = true;
    while (p < f.size) {
      Ins I;
      if (!decode((const unsigned char *)code.data() + f.off + p, f.size - p, (uint32_t)p, I) || p + I.len > f.size) { ok = false; break; }
      ins.push_back(I);
      p += I.len;
    }
    if (!ok) {
      std::snprintf(buf, sizeof buf, "%s+0x%llx: undecodable instruction", f.name.c_str(), (unsigned long long)p);
      findings.push_back(buf); walked = false; continue;
    }
    std::set<uint32_t> leaders;
    leaders.insert(0);
    for (size_t i = 0; i < ins.size(); i++) {
      const Ins &I = ins[i];
      if (I.kind == K_INDIRECT) {
        std::snprintf(buf, sizeof buf, "%s+0x%x: indirect branch or call (control flow cannot be followed)", f.name.c_str(), I.off);
        findings.push_back(buf);
      }
      if (I.kind == K_BRANCH || I.kind == K_CBRANCH) {
        if (I.target < 0 || (uint64_t)I.target >= f.size) {
          std::snprintf(buf, sizeof buf, "%s+0x%x: branch out of the kernel", f.name.c_str(), I.off);
          findings.push_back(buf);
        } else leaders.insert((uint32_t)I.target);
      }
      if ((I.kind == K_BRANCH || I.kind == K_CBRANCH || I.kind == K_END || I.kind == K_INDIRECT) && i + 1 < ins.size()) leaders.insert(ins[i + 1].off);
    }
    for (size_t i = 0; i < ins.size(); i++) {
      if (!ins[i].end_cf) continue;
      bool seen_vec = false, closed = false;
      uint32_t first_vec = 0;
      size_t j = i;
      while (true) {
        if (leaders.count(ins[j].off)) break;       // ins[j] opens the block (it may itself be the restore)
        j--;
        if (ins[j].writes_exec) { closed = true; break; }
        if (ins[j].kind == K_VEC) { seen_vec = true; first_vec = ins[j].off; }
      }
      if (closed || !seen_vec) continue;
      // j = the block's first instruction.  The instructions ahead of the restore ran under the mask of the region that ends here;
      // that is a fault exactly when this block is the region's JOIN block -- where every lane is supposed to be back -- and not
      // a piece of the region's body that happens to end in the restore (a then / else arm the join block was merged or
      // duplicated into, the tail of a region that contains uniform loops).  The join block is proven by how it is entered:
      //   (a) it is the target of the s_cbranch_execz that follows the region's header -- [saveexec into the same pair | an
      //       s_xor / s_mov of that pair with EXEC], then the branch: the lanes that skip the region land here;
      //   (b) it is what an s_cbranch_execnz back edge falls through to whose loop mask -- `s_andn2_b64 exec, exec, s[a:b]` just
      //       ahead of the branch -- is the pair: the exit of a divergent loop.
      // (A region whose skip branch was removed -- a few instructions, no memory access -- has no such witness and is not seen.)
      auto opens_region = [&](size_t br) {   // do the (at most four) instructions ahead of ins[br] write the pair from EXEC?
        size_t k = br;
        for (int n = 0; n < 4 && k > 0; n++) {
          k--;
          if (ins[k].sdst == ins[i].pair && ins[k].reads_exec) return true;
          if (ins[k].kind == K_BRANCH || ins[k].kind == K_CBRANCH || ins[k].kind == K_END) return false;
        }
        return false;
      };
      bool join = false;
      for (size_t bk = 0; bk < ins.size() && !join; bk++)
        if (ins[bk].kind == K_CBRANCH && ins[bk].bop == 8 && ins[bk].target == (int64_t)ins[j].off) join = opens_region(bk);
      if (!join && j > 0 && ins[j - 1].kind == K_CBRANCH && ins[j - 1].bop == 9 && j >= 2) {
        const Ins &lm = ins[j - 2];   // s_andn2_b64 exec, exec, s[a:b]
        join = lm.writes_exec && lm.reads_exec && lm.src_pair == ins[i].pair;
      }
      // (not proven: counted and reported per kernel -- rh_code_object_report's `unproven` -- so that what the rule cannot classify is
      //  visible; such blocks are common and mostly legitimate: a then / else arm that ends in the restore, profiles/r5_parity)
      if (!join) { if (unproven) ++*unproven; continue; }
      std::snprintf(buf, sizeof buf, "%s+0x%x: exec restore (s[%d:%d]) behind vector instructions of its own block (first at +0x%x): "
                    "they ran under the mask of the region that ended", f.name.c_str(), ins[i].off, ins[i].pair, ins[i].pair + 1, first_vec);
      findings.push_back(buf);
    }
  }
  return walked;
}

}  // namespace rh
