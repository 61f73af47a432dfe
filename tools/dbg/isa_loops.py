"""List the loops of one kernel in a -save-temps .s file with instruction counts (mfma / loads / LDS / VALU)."""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index(name + ':')
j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
print(len(body), 'lines')
labels = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = n
VALU = r'^\s+v_'
for n, l in enumerate(body):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if not m:
        continue
    t = m.group(1) or m.group(2)
    if labels.get(t, 1 << 30) < n:
        seg = body[labels[t]:n]
        cnt = lambda pat: sum(1 for x in seg if re.search(pat, x))
        print("loop %s: lines %d-%d mfma %d gload %d gstore %d dsr %d dsw %d barrier %d valu %d accvgpr %d" % (
            t, labels[t], n, cnt('v_mfma'), cnt('global_load|buffer_load'), cnt('global_store|buffer_store'), cnt('ds_read'),
            cnt('ds_write'), cnt('s_barrier'), cnt(VALU), cnt('accvgpr')))
