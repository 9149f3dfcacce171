#!/usr/bin/env python3
"""Build-time guard for the 16-byte store-data hazard that bit the split-K kernel in round 3.

hipcc's hazard recogniser protects the DATA registers of a >64-bit VMEM store against an immediate VALU overwrite only when
the store's soffset is NOT an SGPR ("this hazard only exists if the instruction is not using a register in the soffset
field").  On gfx950 a `buffer_store_dwordx4 v[a:a+3], vN, s[..], sM offen` whose v[a..a+3] is rewritten right behind it
published the NEW register contents in the lanes it reads last whenever the memory pipeline was slow to take the store.  The
kernels defend themselves in source (data registers pinned until `s_waitcnt vmcnt(0)`, or the whole offset in the VGPR);
this script checks the machine code those sources compile to, so a compiler upgrade cannot bring the bug back silently:

  rule A  a 12/16-byte buffer store with an SGPR soffset: no instruction may write any of its data registers before the next
          `s_waitcnt` that waits for vmcnt(0) (or the end of the program);
  rule B  any 12/16-byte buffer / global / flat / scratch store: no VALU instruction may write its data registers within the
          next 2 wait states (what LLVM inserts for gfx940+ when it does see the hazard).

usage: check_store_hazard.py a.o [b.o ...]     (host objects with an embedded gfx950 bundle, or .s / .txt disassembly)
exit status 1 and one line per finding when a rule is violated -- or when an object holds FEWER stores / functions than the floors
committed in check_store_hazard.floors.json (fail closed: a toolchain change that makes the rules match nothing is a failure, not a
pass; `--print-floors a.o ...` prints the current counts in that file's format).  Linear scan per function (branches are not followed: a
region that crosses one is still checked in address order, which can only over-report).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_REG = re.compile(r"^(v|a)(?:\[(\d+):(\d+)\]|(\d+))$")
_WIDE_STORE = re.compile(r"^(buffer|global|flat|scratch)_store_dwordx[34]$")
_NO_VGPR_DEST = re.compile(r"^(s_|buffer_store|global_store|flat_store|scratch_store|ds_write|ds_store|ds_gws|ds_nop|v_nop|"
                           r"v_cmpx|buffer_wbl2|buffer_inv|buffer_gl|exp\b|v_readfirstlane|v_readlane)")
_TWO_DESTS = re.compile(r"^(v_swap_b32|v_permlane16_swap|v_permlane32_swap)")


def vgprs(op):
    """set of ('v', i) a register operand names; empty for anything that is not a plain VGPR (range)"""
    m = _REG.match(op.strip())
    if not m or m.group(1) != "v":
        return set()
    if m.group(4) is not None:
        return {int(m.group(4))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def disassemble(path):
    """text of the gfx950 device code inside `path` (.o with offload bundles) or `path` itself when it already is text"""
    if path.endswith((".s", ".txt", ".asm")):
        return open(path).read()
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = []
        for fn in sorted(os.listdir(d)):
            if "amdgcn" in fn and "gfx" in fn:
                out.append(subprocess.run([OBJDUMP, "-d", os.path.join(d, fn)], check=True, stdout=subprocess.PIPE,
                                          text=True).stdout)
        if not out:   # a plain device code object
            out.append(subprocess.run([OBJDUMP, "-d", local], check=True, stdout=subprocess.PIPE, text=True).stdout)
        return "\n".join(out)


def parse(text):
    """{function: [(mnemonic, [operands], raw line)]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith(("\t", " ")):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        mnem = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((mnem, ops, body))
    return funcs


def dest_vgprs(mnem, ops, body):
    if not ops or _NO_VGPR_DEST.match(mnem):
        return set()
    if mnem.startswith(("buffer_load", "global_load")) and re.search(r"\blds\b", body):
        return set()      # LDS-DMA: no register destination
    d = vgprs(ops[0])
    if _TWO_DESTS.match(mnem) and len(ops) > 1:
        d |= vgprs(ops[1])
    return d


def wait_states(mnem, ops):
    if mnem == "s_nop" and ops:
        try:
            return int(ops[0], 0) + 1
        except ValueError:
            return 1
    return 1


def store_info(mnem, ops):
    """(data registers, soffset-is-SGPR) of a 12/16-byte store, or None"""
    m = _WIDE_STORE.match(mnem)
    if not m:
        return None
    kind = m.group(1)
    if kind == "buffer":
        data = vgprs(ops[0])
        sgpr_soffset = False
        # operands: vdata, vaddr | off, srsrc, soffset [modifiers ...]
        for i, o in enumerate(ops):
            if re.match(r"^s\[\d+:\d+\]$", o) and i + 1 < len(ops):
                so = ops[i + 1].split()[0]
                sgpr_soffset = bool(re.match(r"^(s\d+|m0|ttmp\d+)$", so))
                break
        return data, sgpr_soffset
    data = vgprs(ops[1]) if len(ops) > 1 else set()
    return data, False


def check(text, name="<input>"):
    findings = []
    for fn, ins in parse(text).items():
        for i, (mnem, ops, body) in enumerate(ins):
            info = store_info(mnem, ops)
            if not info or not info[0]:
                continue
            data, sgpr = info
            # rule B: VALU write within the next two wait states
            ws = 0
            for mn2, op2, b2 in ins[i + 1:]:
                if ws >= 2:
                    break
                if mn2.startswith("v_") and dest_vgprs(mn2, op2, b2) & data:
                    findings.append("%s: %s: rule B: `%s` overwritten by `%s` after %d wait state(s)" % (name, fn, body, b2, ws))
                    break
                ws += wait_states(mn2, op2)
            if not sgpr:
                continue
            # rule A: nothing writes the data registers before vmcnt(0)
            for mn2, op2, b2 in ins[i + 1:]:
                if mn2 == "s_endpgm" or (mn2 == "s_waitcnt" and re.search(r"vmcnt\(0\)", b2)):
                    break
                if dest_vgprs(mn2, op2, b2) & data:
                    findings.append("%s: %s: rule A: `%s` (SGPR soffset) data overwritten by `%s` before vmcnt(0)"
                                    % (name, fn, body, b2))
                    break
    return findings


def count_wide_stores(text):
    """(12/16-byte stores with an SGPR soffset -- rule A's subjects, all 12/16-byte stores -- rule B's subjects, functions)"""
    sgpr = wide = 0
    funcs = parse(text)
    for ins in funcs.values():
        for mnem, ops, _ in ins:
            info = store_info(mnem, ops)
            if info:
                wide += 1
                sgpr += bool(info[1])
    return sgpr, wide, len(funcs)


def count_wide_sgpr_stores(text):
    return count_wide_stores(text)[0]


FLOORS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "check_store_hazard.floors.json")


def load_floors(path=FLOORS_FILE):
    import json
    with open(path) as f:
        return {k: v for k, v in json.load(f).items() if not k.startswith("_")}


def below_floor(name, text, floors):
    """The guard fails CLOSED: an object this script is told to check must still contain at least the stores it was written
    against (floors committed next to the script: today's counts).  A hipcc that renames a mnemonic, changes the operand syntax
    or the disassembly format would otherwise make both rules match nothing and turn the Makefile step into a no-op.  An object
    without an entry in the floors file is an error too (add one with --print-floors)."""
    fl = floors.get(name)
    if fl is None:
        return ["%s: no entry in %s: the guard cannot tell a clean object from one it no longer understands"
                % (name, os.path.basename(FLOORS_FILE))]
    sgpr, wide, nfunc = count_wide_stores(text)
    out = []
    for key, got in (("sgpr_soffset_stores", sgpr), ("wide_stores", wide), ("functions", nfunc)):
        if got < fl.get(key, 0):
            out.append("%s: fail closed: %d %s matched, floor %d (did hipcc / llvm-objdump change a mnemonic or the operand "
                       "syntax?  update the parser, then the floors file)" % (name, got, key, fl[key]))
    return out


def main(argv):
    if not argv:
        print(__doc__)
        return 2
    if argv[0] == "--print-floors":
        import json
        doc = {}
        for path in argv[1:]:
            sgpr, wide, nfunc = count_wide_stores(disassemble(path))
            doc[os.path.basename(path)] = {"sgpr_soffset_stores": sgpr, "wide_stores": wide, "functions": nfunc}
        print(json.dumps(doc, indent=1))
        return 0
    floors = load_floors()
    bad = []
    for path in argv:
        text = disassemble(path)
        name = os.path.basename(path)
        f = check(text, name) + below_floor(name, text, floors)
        bad += f
        sgpr, wide, _ = count_wide_stores(text)
        print("check_store_hazard: %s: %d wide stores (%d with an SGPR soffset), %d finding(s)" % (name, wide, sgpr, len(f)))
    for line in bad:
        print(line, file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
