"""One character per instruction of a kernel's line range in a -save-temps .s file: M mfma, r/w LDS read/write, . VALU, W waitcnt,
B barrier, G global load, S global store, J branch, n nop, s other scalar."""
import sys

path, name, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
s = open(path).read()
body = s[s.index(name + ':'):].split('\n')[lo:hi]
seq = []
for l in body:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        if l.startswith('.LBB'):
            seq.append('\n' + l.split(':')[0] + ': ')
        continue
    op = l.split()[0]
    seq.append('M' if op.startswith('v_mfma') else 'r' if op.startswith('ds_read') else 'w' if op.startswith('ds_write') else
               '.' if op.startswith('v_') else 'W' if op.startswith('s_waitcnt') else 'B' if op.startswith('s_barrier') else
               'G' if op.startswith(('global_load', 'buffer_load')) else 'S' if op.startswith(('global_store', 'buffer_store')) else
               'J' if op.startswith(('s_cbranch', 's_branch')) else 'n' if op.startswith('s_nop') else 's')
print(''.join(seq))
