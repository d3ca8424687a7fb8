"""Where a kernel's scratch accesses (register spills) and its loads / waits sit: per basic-block label, for the part of the assembly between
two s_endpgm.  usage: python tools/isa_scratch_map.py /tmp/plsvo_isa/align_kernels.s <kernel index, 0-based among the file's kernels>"""
import re, sys
from collections import Counter, OrderedDict
lines = open(sys.argv[1]).read().split('\n')
ends = [i for i, l in enumerate(lines) if 's_endpgm' in l]
k = int(sys.argv[2])
lo = ends[k - 1] + 1 if k > 0 else 0
body = lines[lo:ends[k] + 1]
lab = 'entry'
blocks = OrderedDict()
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        lab = m.group(1)
    b = blocks.setdefault(lab, Counter())
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    b['n'] += 1
    for key, pat in (('scratch', r'^scratch_'), ('glds', r'^global_load_lds'), ('gload', r'^global_load_(dword|ub|us|sb)'), ('gstore', r'^global_store'),
                     ('ds', r'^ds_'), ('vm_wait', r'vmcnt'), ('f64', r'_f64'), ('readlane', r'v_readlane|v_writelane'), ('branch_back', r'^s_cbranch|^s_branch')):
        if re.search(pat, t):
            b[key] += 1
print(f"kernel {k}: {sum(b['n'] for b in blocks.values())} instructions, {len(blocks)} blocks")
for lab, b in blocks.items():
    if b['n'] >= 40 or b['scratch'] or b['glds']:
        print(f"{lab:12s} n={b['n']:5d} f64={b['f64']:4d} scratch={b['scratch']:3d} glds={b['glds']:3d} gload={b['gload']:3d} gstore={b['gstore']:3d} ds={b['ds']:3d} vmwait={b['vm_wait']:2d} lanes={b['readlane']:3d}")
