import sys, numpy as np
sys.path.insert(0,'.')
from hsearch_amd import Engine, synth
k,K,L,W,R,n,nq=25,16,8,200.0,40.0,10_000_000,100_000
a,b=synth.make_planes(k,K,L,W); codes=synth.make_db(n,k); centers,_=synth.make_queries(codes,nq)
eng=Engine(k,K,L,W,a,b); eng.index_build(codes)
for mode in ("join","join16","stream"):
    eng.set_verify_mode(mode)
    r=eng.query(centers,R,want_cand=False); p=eng.profile()
    r=eng.query(centers,R,want_cand=False); p=eng.profile()
    print(mode, "hits",len(r["q"]),"provisional",p["provisional"],"cand",p["candidates"],"join_pairs",p["join_pairs"],"ms_join %.3f ms_verify %.3f ms_final %.3f ms_total %.3f"%(p["ms_join"],p["ms_verify"],p["ms_finalize"],p["ms_total"]))
