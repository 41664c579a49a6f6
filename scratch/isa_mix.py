"""Instruction mix of one kernel in a hipcc -S dump:  python scratch/isa_mix.py file.s pc_step_kernelILi16E"""
import sys, re
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
name = m.group(1)
i = m.end(); j = s.index('.Lfunc_end', i)
b = s[i:j]
c = Counter(l.split()[0] for l in b.split('\n') if l.strip() and not l.strip().startswith(('.', ';')))
keys = [k for k in c if k.startswith(('ds_', 'flat_', 'global_', 'scratch_', 'buffer_', 'v_mfma', 's_barrier'))]
print('; '.join(f'{k} {c[k]}' for k in sorted(keys)))
for what in ('num_vgpr', 'num_agpr', 'private_seg_size'):
    mm = re.search(re.escape(name) + r'\.' + what + r', (\d+)', s)
    print(what, mm.group(1) if mm else None, end='  ')
print()
