#!/usr/bin/env python3
"""Build-time guard for the per-wave activation ring of the small-batch kernel (streamk_kernel<..., XM >= 2>).

A ring slot is filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, kDma per stage) and read back with hand-written `ds_read_b128`s.
Nothing but ORDER makes a read see the DMA's data: the DMAs of a stage are issued BEFORE its weight loads, the reads come AFTER the
wait for those weights, and vector-memory operations retire in order (one vmcnt counts them all on gfx9).  The source pins that order
(a compiler barrier + a scheduling barrier behind the DMA, the stage's weight register among the inputs of the inline reads); this
script checks the machine code, so that a compiler upgrade that moves a load across a DMA, or drops a wait, cannot turn into silently
stale activations.  Per ring kernel, one linear pass in address order (the kernel is prologue / loop / tail / epilogue; the vector-
memory queue at the loop head, at its end and on the path that skips it is the same D stages in flight, so a linear pass sees every
state the hardware can be in):

  rule 1  when a weight load (`global_load_dwordx4`) is issued, the DMAs of its stage have been issued: DMA groups >= weight groups + 1
  rule 2  at every `ds_read_b128` of the main path at most (D - 1) * kDma LDS-DMAs are outstanding in the simulated queue -- the slot
          being read is the OLDEST stage in flight, so its DMAs must have retired; the refills of the other D - 1 stages may fly
  rule 3  every ring kernel has LDS-DMAs and `ds_read_b128`s (the check is not vacuous)

usage: check_ring_order.py streamk.o     (or .s / .txt disassembly); exit status 1 and one line per finding.
"""
import importlib.util
import os
import re
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(_here, "check_store_hazard.py"))
_hz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_hz)
disassemble, parse = _hz.disassemble, _hz.parse

_RING = re.compile(r"streamk_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi([2-9])EE")
_VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")


def ring_params(fn):
    m = _RING.search(fn)
    if not m:
        return None
    mt, nt, waves, depth, occ, bits, xm = (int(g) for g in m.groups())
    rows = 4 if xm == 3 else 16 if xm == 4 else 32 if xm == 5 else 8
    kdma = rows * (128 if bits == 8 else 256) // 1024
    return {"nt": nt, "depth": depth, "kdma": kdma}


def _vmcnt(body):
    m = re.search(r"vmcnt\((\d+)\)", body)
    return int(m.group(1)) if m else None


def check(text, name="<input>"):
    findings, seen = [], 0
    for fn, ins in parse(text).items():
        p = ring_params(fn)
        if not p:
            continue
        seen += 1
        queue = []                      # outstanding vector-memory operations, oldest first: "dma" / "w" / "other"
        n_dma = n_w = n_reads = 0
        limit = (p["depth"] - 1) * p["kdma"]
        for mnem, ops, body in ins:
            if mnem == "s_waitcnt":
                n = _vmcnt(body)
                if n is not None and len(queue) > n:
                    queue = queue[len(queue) - n:] if n else []
                continue
            if _VMEM.match(mnem):
                if mnem.startswith("buffer_load") and re.search(r"\blds\b", body):
                    queue.append("dma")
                    n_dma += 1
                elif mnem == "global_load_dwordx4":
                    if n_dma // p["kdma"] < n_w // p["nt"] + 1:
                        findings.append("%s: %s: rule 1: `%s` issued before the DMAs of its stage (%d DMAs, %d weight loads so far)"
                                        % (name, fn, body, n_dma, n_w))
                    queue.append("w")
                    n_w += 1
                else:
                    queue.append("other")
                continue
            if mnem == "ds_read_b128":
                n_reads += 1
                out = queue.count("dma")
                if out > limit:
                    findings.append("%s: %s: rule 2: `%s` with %d LDS-DMAs outstanding (at most %d may be: the slot's own must have retired)"
                                    % (name, fn, body, out, limit))
        if n_dma == 0 or n_reads == 0:
            findings.append("%s: %s: rule 3: no LDS-DMA / no ds_read_b128 in a ring kernel" % (name, fn))
    return findings, seen


def main(argv):
    bad = 0
    for path in argv:
        findings, seen = check(disassemble(path), os.path.basename(path))
        print("check_ring_order: %s: %d ring kernels, %d finding(s)" % (os.path.basename(path), seen, len(findings)))
        for f in findings:
            print("  " + f)
        bad += len(findings)
        if seen == 0:
            print("  no ring kernel found (streamk_kernel<..., XM >= 2>): the check would be vacuous")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
