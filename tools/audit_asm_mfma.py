#!/usr/bin/env python3
"""Audit of a kernel that issues its MFMAs as asm statements (kz_tower4.hip): hipcc pads no hazard around them, so the
ISA must not contain (a) a compiler-generated read of an accumulator register within the wait states of the MFMA that
wrote it, (b) a compiler write (v_accvgpr_write / VALU) of an MFMA operand right in front of the statement.
usage: audit_asm_mfma.py file.s   (exit code 1 on a finding)"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
recent = []   # (line, lo, hi) of the last MFMA destinations
findings = 0
last_valu_writes = []  # (line, reg) VGPR writes by VALU in the last few instructions
for i, l in enumerate(lines):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if t.startswith('v_mfma'):
        m = re.match(r'v_mfma\S+ a\[(\d+):(\d+)\], (\S+), (\S+), (\S+)', t)
        if m:
            recent.append((i, int(m.group(1)), int(m.group(2))))
            recent = recent[-2:]
            for op in (m.group(3), m.group(4)):
                mm = re.match(r'v\[(\d+):(\d+)\]', op.rstrip(','))
                if mm:
                    lo, hi = int(mm.group(1)), int(mm.group(2))
                    for (j, r) in last_valu_writes:
                        if lo <= r <= hi and i - j <= 3:
                            findings += 1
                            print(f"line {i}: MFMA operand v{r} written by VALU at line {j}: {lines[j].strip()}")
        continue
    if t.startswith('s_nop 15'):
        recent = []
    if t.startswith('v_accvgpr_read'):
        r = int(re.search(r'a(\d+)$', t).group(1))
        for (j, lo, hi) in recent:
            if lo <= r <= hi:
                findings += 1
                print(f"line {i}: {t}  reads the result of the MFMA at line {j} inside its wait states")
    if t.startswith('v_accvgpr_write'):
        r = int(re.match(r'v_accvgpr_write_b32 a(\d+)', t).group(1))
        # a compiler write into an accumulator between MFMAs = an accumulator that does not live in the accumulator file
        if recent:
            findings += 1
            print(f"line {i}: {t}  compiler write into the accumulator file among the MFMAs")
    m = re.match(r'v_(?!mfma|accvgpr)\w+ v(\d+)', t)
    if m:
        last_valu_writes.append((i, int(m.group(1))))
        last_valu_writes = last_valu_writes[-8:]
print(f"{sys.argv[1]}: {findings} finding(s)")
sys.exit(1 if findings else 0)
