"""Register / spill / LDS figures of the kernels in a hipcc --save-temps assembly file.
usage: python experiments/kernel_regs.py file.s [name-substring ...]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    name, blk = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    g = lambda k: (re.search(k + r':\s+(\d+)', blk) or [0, '-'])[1]
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    print(d[:170], '| vgpr', g(r'\.vgpr_count'), 'agpr', g(r'\.agpr_count'), 'spill', g(r'\.vgpr_spill_count'), 'sgpr', g(r'\.sgpr_count'))
