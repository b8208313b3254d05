"""ISA lint: MUBUF stores of more than 8 bytes (buffer_store_dwordx3/x4) whose write-data VGPRs are overwritten by the very next
VALU instruction.  hipcc (LLVM GCNHazardRecognizer::createsVALUHazard) assumes that hazard away when the store's soffset is an
SGPR; measured on gfx950 (experiments/gemm_cuh_bench.hip, round 5) it is real: pw_gemm_cuh_kernel<_Float16> stored the NEXT
row's values in 1.6 % of its 16-byte stores until a wait state followed the store.
usage: python experiments/lint_store_hazard.py file.s [...]   (hipcc --save-temps assembly)"""
import re, sys

def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

bad = 0
for path in sys.argv[1:]:
    func = None
    lines = open(path).read().split('\n')
    code = []
    for ln in lines:
        t = ln.split(';')[0].strip()
        m = re.match(r'^(_Z\w+|\w+):$', t)
        if m:
            func = m.group(1)
        if t and not t.startswith('.') and not t.endswith(':'):
            code.append((func, t))
    for i, (fn, t) in enumerate(code[:-1]):
        if not re.match(r'buffer_store_dwordx[34]\b', t):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        data = regs(ops[0])
        soff_sgpr = len(ops) > 3 and re.match(r's\d+', ops[3].split()[0]) is not None
        nxt = code[i + 1][1]
        if not re.match(r'v_', nxt):
            continue
        dst = regs(nxt.split(None, 1)[1].split(',')[0].strip())
        if data & dst:
            bad += 1
            print(f'{path}: {fn}: "{t}" -> "{nxt}"  (soffset sgpr: {soff_sgpr})')
print('hazards:', bad)
sys.exit(1 if bad else 0)
