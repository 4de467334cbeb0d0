"""Instruction histogram of one kernel in a hipcc -S --cuda-device-only dump (scratch tool).
usage: asm_stats.py file.s mangled_name_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith('E') and ':' in l)
end = next(i for i in range(start, len(s)) if '.end_amdhsa_kernel' in s[i])
body = s[start:end]
ins = [l.strip().split(';')[0].strip() for l in body]
ins = [l for l in ins if l and not l.startswith('.') and not l.endswith(':')]
c = Counter(l.split()[0] for l in ins)
print(len(ins), 'instructions')
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print('%6d %s' % (v, k))
for l in body:
    if any(k in l for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', 'private_segment_fixed_size', '.amdhsa_group_segment_fixed_size')):
        print(l.strip())
