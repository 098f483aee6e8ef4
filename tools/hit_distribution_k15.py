import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
k,K,L,W,R,n,nq=15,16,8,212.0,40.0,10_000_000,100_000
a,b=synth.make_planes(k,K,L,W); codes=synth.make_db(n,k); centers,_=synth.make_queries(codes,nq,seed=synth.SEED_QUERIES)
eng=Engine(k,K,L,W,a,b,device=0); eng.index_build(codes)
got=eng.query(centers,R)
c=np.bincount(got["q"],minlength=nq)
print(json.dumps({"hits":int(c.sum()),"max":int(c.max()),"mean":float(c.mean()),"p50":float(np.median(c)),"gt48":int((c>48).sum()),"gt1024":int((c>1024).sum()),"gt8192":int((c>8192).sum()),"gt65536":int((c>65536).sum())}))
