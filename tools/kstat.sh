#!/usr/bin/env bash
# tools/kstat.sh <file.hip> <kernel-name-substring>: registers, scratch, occupancy, code size and an instruction histogram per matching kernel
f=$1; pat=$2
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize ${ISX_EXTRA_FLAGS:-} -x hip -S --cuda-device-only "$(basename "$f")" -o /tmp/kstat.s 2>/dev/null || exit 1
python3 - "$pat" <<'PY'
import re,collections,sys
s=open('/tmp/kstat.s').read(); pat=sys.argv[1]
for m in re.finditer(r'^(\S*%s\S*):\s*; @'%re.escape(pat), s, re.M):
    name=m.group(1); i=m.end(); j=s.find('s_endpgm',i)
    body=s[i:j]
    tail=s[j:s.find("; Occupancy:",j)+40]
    ops=collections.Counter()
    for line in body.split('\n'):
        mm=re.match(r'\s+([a-z][a-z0-9_]+)',line)
        if mm: ops[mm.group(1)]+=1
    g=lambda k:(re.search(r'; %s[:=]? *=? *(\d+)'%k,tail) or re.search(r'(0)',' 0')).group(1)
    v=sum(c for o,c in ops.items() if o.startswith('v_')); sc=sum(c for o,c in ops.items() if o.startswith('s_'))
    print(name[:100]); print('   vgpr',g('NumVgprs'),'sgpr',g('NumSgprs'),'scratch',g('ScratchSize'),'occupancy',g('Occupancy'),'code bytes',g('codeLenInByte'),
          '| insts',sum(ops.values()),'valu',v,'salu',sc,'vmem',sum(c for o,c in ops.items() if o.startswith(('global_','buffer_','flat_'))),'lds',sum(c for o,c in ops.items() if o.startswith('ds_')),
          'dpp',sum(1 for l in body.split('\n') if 'dpp' in l and not l.strip().startswith(';')))
    print('   ', ops.most_common(16))
PY
