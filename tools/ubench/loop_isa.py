import re,sys
from collections import Counter
lines=open(sys.argv[1]).read().split('\n')
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
loops=[]
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        loops.append((labels[m.group(1)],i))
for a,b in loops:
    ops=[l.strip().split()[0] for l in lines[a+1:b+1] if re.match(r'^\s+[a-z]',l)]
    c=Counter(ops)
    valu=sum(v for k,v in c.items() if k.startswith('v_'))
    vmem=sum(v for k,v in c.items() if k.startswith(('global_','buffer_','flat_','scratch_')))
    lds=sum(v for k,v in c.items() if k.startswith('ds_'))
    print(f"loop lines {a}-{b}: total {len(ops)} valu {valu} vmem {vmem} lds {lds} salu {sum(v for k,v in c.items() if k.startswith('s_'))}")
    if len(sys.argv)>2 and int(sys.argv[2])==a:
        for k,v in c.most_common(): print('   ',v,k)
