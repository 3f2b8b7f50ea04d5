#!/usr/bin/env python3
"""CPU: what the postings a read end expands ARE (profiles/EXPERIMENTS.md, round 5) -- credited, owned by an earlier probe,
or too short, whether the short ones stop at a node boundary, and how many a two-base junction pretest would reject
(never a credited one: asserted) -- counted with the string-level model of the device
algorithm (tests/seed_extend_model.py) on a prefix of a bench stream.

    python tools/posting_classes.py <config> <pairs>
"""
import sys, tempfile, collections
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import seed_extend_model as M
from oracle import pe_oracle_c
from vstrains_amd.workloads import CONFIGS, workload_for
config=int(sys.argv[1]); n=int(sys.argv[2])
cfg=CONFIGS[config]; L,k=cfg['read_len'],cfg['k']; K=k+1
st, pre, names, seqs, cum, logger, _ = workload_for(config, tempfile.mkdtemp())
tab,w,s=M.build(seqs,K); rcs=[M.rc(x) if len(x)>=K else '' for x in seqs]
fw,rv=pe_oracle_c.synth_pairs(st.genomes,cum,20250000+config,0,n,L,int(0.005*2**32),int(0.001*2**32))
cnt=collections.Counter(); lens=[]
for arr in (fw,rv):
  for p in range(n):
    read=arr[p].tobytes().decode()
    if 'N' in read: continue
    rlen=len(read); j=M.phase(rlen,w,s); accepted=set()
    while j+w<=rlen:
        f=read[j:j+w]; r=M.rc(f); key,sr=(f,0) if f<r else (r,1)
        for node,pp,sn in tab.get(key,()):
            opp=sn^sr; text=rcs[node] if opp else seqs[node]; tlen=len(text); q=tlen-pp-w if opp else pp
            c=min(s,j,q); left=0
            while left<c and read[j-1-left]==text[q-1-left]: left+=1
            ext=0
            while j+w+ext<rlen and q+w+ext<tlen and read[j+w+ext]==text[q+w+ext]: ext+=1
            ln=left+w+ext
            # the junction-base pretest (EXPERIMENTS r5): a seed inside the first k bases of the strand can reach K bases
            # only through the strand's base at position k; one inside the last k only through the base at tlen-k-1
            rej=False
            if q+w<=k:
                rp=j+(k-q)
                if rp>=rlen or read[rp]!=text[k]: rej=True
            if q>=tlen-k:
                rp=j-(q-(tlen-k-1))
                if rp<0 or read[rp]!=text[tlen-k-1]: rej=True
            if rej:
                cnt['pretest_rejects']+=1
                assert left>=s or ln<K, 'the pretest rejected a posting that is credited'
            if left>=s: cnt['owned_by_earlier']+=1
            elif ln<K:
                cnt['too_short']+=1
                # which side stops it: node boundary or mismatch?
                lb = (left==q and left<min(s,j)); rb=(q+w+ext==tlen and j+w+ext<rlen)
                cnt['too_short_node_boundary_both' if (lb and rb) else 'too_short_boundary_one' if (lb or rb) else 'too_short_mismatch']+=1
            else: cnt['credited']+=1
            cnt['all']+=1
        j+=s
    cnt['ends']+=1
print({k:round(v/cnt['ends'],2) for k,v in cnt.items()})
print('node lengths: median', int(np.median([len(x) for x in seqs])), 'mean', round(np.mean([len(x) for x in seqs]),1))
