"""Guard against the hazard found while moving gA_{depth-1} off the tape: the result register of an
inline-asm VALU instruction may be one that an MFMA issued a few instructions earlier is still reading as a source --
inline asm is opaque to hipcc's hazard recogniser.  Measured on gfx950: an asm write into SrcB six instructions after the
MFMA corrupted the product; writes into SrcA three or more instructions later are what the shipped kernels do and are
covered by the parity tests.

Until round 2 the relu bits were taken with an inline-asm v_pk_min_u16 (fused_common.h); when the pack was software-
pipelined over the k-steps this check caught that instruction two slots behind an MFMA whose SrcA it overwrote, and the
bits are now a compiler-visible signed packed min (PolBF16::nonzero_halves, v_pk_min_i16).  No VALU instruction is emitted
through inline asm any more; the check stays as a guard: the built objects are disassembled and every instruction with one
of the opcodes below is checked against the most recent MFMA (no other MFMA in between):

    python tools/check_asm_hazard.py [objects ...]        # default: bhnerf_amd/csrc/fused_{fwd,bwd}.o
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ASM_OPCODES = ('v_pk_min_u16',)
MIN_DIST_SRCA = 3           # smallest distance (instructions) the verified kernels have


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, capture_output=True)
        code = [f for f in os.listdir(d) if 'amdgcn' in f]
        if not code:
            raise RuntimeError('no device code object in %s' % obj)
        return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', os.path.join(d, code[0])], check=True,
                              capture_output=True, text=True).stdout


def scan(text):
    """[(kernel, operand 'A'|'B', distance, instruction)] for asm-opcode writes into a source of the latest MFMA."""
    hits, kern, last_mfma, dist = [], None, None, 0
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            kern, last_mfma = m.group(1), None
            continue
        t = line.split('//')[0].strip()
        if not t or not re.match(r'^[sv]_|^ds_|^buffer_|^global_|^scratch_|^flat_', t):
            continue
        mm = re.match(r'v_mfma\S+\s+\S+,\s*v\[(\d+):(\d+)\],\s*v\[(\d+):(\d+)\]', t)
        if mm:
            last_mfma, dist = tuple(map(int, mm.groups())), 0
            continue
        dist += 1
        if last_mfma and t.startswith(ASM_OPCODES):
            dst = int(re.match(r'\S+\s+v(\d+)', t).group(1))
            a0, a1, b0, b1 = last_mfma
            if a0 <= dst <= a1:
                hits.append((kern, 'A', dist, t))
            elif b0 <= dst <= b1:
                hits.append((kern, 'B', dist, t))
    return hits


def violations(objs):
    bad = []
    for obj in objs:
        for kern, which, dist, ins in scan(disassemble(obj)):
            if which == 'B' or dist < MIN_DIST_SRCA:
                bad.append((os.path.basename(obj), kern, which, dist, ins))
    return bad


if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = sys.argv[1:] or [os.path.join(root, 'bhnerf_amd', 'csrc', f) for f in ('fused_fwd.o', 'fused_bwd.o')]
    bad = violations(objs)
    for b in bad:
        print('HAZARD %s %s: write into Src%s %d instructions after the MFMA: %s' % b)
    print('%d violation(s) in %d object(s)' % (len(bad), len(objs)))
    sys.exit(1 if bad else 0)
