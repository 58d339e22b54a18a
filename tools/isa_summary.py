import re,sys
txt=open(sys.argv[1]).read()
pat=sys.argv[2]
for m in re.finditer(r'^(_Z\w*(?:%s)\w*):.*?\n(.*?)s_endpgm'%pat, txt, re.S|re.M):
    body=m.group(2); lines=body.split('\n')
    print(m.group(1)[:100])
    prev=None;cnt=0;out=[]
    for i,l in enumerate(lines):
        l=l.strip()
        k=None
        if re.search(r'global_load|buffer_load',l): k='LOAD'
        elif 'vmcnt' in l: k=l.split(';')[0].strip().replace('s_waitcnt ','')
        elif 's_cbranch' in l: k='br'
        elif 's_barrier' in l: k='BARRIER'
        elif 'v_mfma' in l: k='mfma'
        elif re.search(r'global_store|buffer_store',l): k='STORE'
        elif 'ds_write' in l: k='dsw'
        elif 'scratch_' in l: k='SCRATCH'
        if k is None: continue
        if k==prev: cnt+=1
        else:
            if prev: out.append(f'{prev}x{cnt}')
            prev=k;cnt=1
    out.append(f'{prev}x{cnt}')
    print('   ',' | '.join(out))
