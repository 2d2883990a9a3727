"""Where does a kernel wait for memory more often than its dataflow needs?  For every kernel of an assembly listing (hipcc -S --cuda-device-only)
whose name contains one of the given substrings: the sequence of global loads and `s_waitcnt vmcnt(n)` up to the first global store, compressed -
`loads xN | wait(0)` repeated means the loads are drained group by group (control flow between loads makes the compiler do that).
    python tools/isa_drains.py /tmp/kstat.s k_warp_tile k_pyr_down0 ..."""
import re
import sys

s = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r'^(\S*%s\S*):\s*; @' % re.escape(pat), s, re.M):
        i = m.end(); j = s.find('.Lfunc_end', i)
        out = []
        for l in s[i:j].split('\n'):
            t = l.strip().split(' ')[0]
            if t.startswith(('global_load', 'buffer_load', 'flat_load')): out.append('L')
            elif t == 's_waitcnt' and 'vmcnt' in l: out.append('w' + re.search(r'vmcnt\((\d+)\)', l).group(1))
            elif t.startswith('global_store'): out.append('S')
            elif t == 's_barrier': out.append('B')
        res, prev, cnt = [], None, 0
        for o in out + [None]:
            if o == prev: cnt += 1
            else:
                if prev: res.append(prev + ('x%d' % cnt if cnt > 1 else ''))
                prev, cnt = o, 1
        loads = out.count('L'); drains = sum(1 for k, o in enumerate(out) if o.startswith('w') and k > 0 and out[k - 1] == 'L')
        print(m.group(1)[:110]); print('   loads %d, waits right behind a load %d :  %s' % (loads, drains, ' '.join(res[:90])))
