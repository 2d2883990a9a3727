#!/usr/bin/env bash
# static instruction histogram of one kernel: tools/isa_count.sh <file.hip> <mangled-name-substring>
f=$1; pat=$2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -x hip -S --cuda-device-only "$f" -o /tmp/isa.s 2>/dev/null
python3 - "$pat" <<'PY'
import re,collections,sys
s=open('/tmp/isa.s').read(); pat=sys.argv[1]
for m in re.finditer(r'^(\S*%s\S*):\s*; @'%re.escape(pat), s, re.M):
    name=m.group(1); i=m.end(); j=s.find('s_endpgm',i)
    ops=collections.Counter()
    for line in s[i:j].split('\n'):
        mm=re.match(r'\s+([a-z][a-z0-9_]+)',line)
        if mm: ops[mm.group(1)]+=1
    v=sum(c for o,c in ops.items() if o.startswith('v_')); sc=sum(c for o,c in ops.items() if o.startswith('s_'))
    print(name[:90], 'total',sum(ops.values()),'valu',v,'salu',sc,'vmem',sum(c for o,c in ops.items() if o.startswith(('global_','buffer_','flat_'))),'lds',sum(c for o,c in ops.items() if o.startswith('ds_')))
    print('   ', ops.most_common(14))
PY
