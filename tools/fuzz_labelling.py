#!/usr/bin/env python3
"""Randomised cross-check of the shortest-path labelling mirrors (pairs: directed / undirected, invalid edges, self loops,
empty graphs; the single shortest path) against the oracle.   python tools/fuzz_labelling.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evi_rag_amd import labelling
from oracle import graph as og
dev=torch.device("cuda:0"); rng=np.random.default_rng(9)
for c in range(40):
    B=int(rng.choice([1,3,20])); directed=bool(rng.integers(0,2))
    per=[]
    for g in range(B):
        n=int(rng.choice([1,2,6,50,700])); m=int(rng.choice([0,1,5,120,3000]))
        src=rng.integers(-1 if rng.random()<0.2 else 0, n+ (1 if rng.random()<0.2 else 0), m)
        dst=rng.integers(0, n, m)
        if m>3: dst[0]=src[0]  # a self loop
        q=rng.integers(0,n,int(rng.integers(0,4))).tolist(); a=rng.integers(0,n,int(rng.integers(0,6))).tolist()
        per.append((n,src.astype(np.int64),dst.astype(np.int64),q,a))
    gb=labelling.GraphBatch([p[0] for p in per],[p[1] for p in per],[p[2] for p in per],device=dev)
    got=labelling.shortest_path_union_mask_by_pair_batch(gb,[p[3] for p in per],[p[4] for p in per],directed=directed)
    for g,(n,src,dst,q,a) in enumerate(per):
        want=og.shortest_path_union_mask_by_pair(n,src.tolist(),dst.tolist(),q,a,directed=directed)
        assert np.array_equal(np.asarray(want[0],bool), got[g][0]), (c,g,"mask")
        for j in range(1,6):
            assert list(want[j])==list(got[g][j]), (c,g,j,want[j],got[g][j])
    # single shortest path for the whole batch (valid sources / targets only, as the batch API takes local ids)
    qs=[[x for x in p[3] if 0<=x<p[0]] for p in per]; as_=[[x for x in p[4] if 0<=x<p[0]] for p in per]
    gps=labelling.shortest_path_single_batch(gb, qs, as_, path_cap=64)
    for g,(n,src,dst,q,a) in enumerate(per):
        wp=og.shortest_path_single(n,src.tolist(),dst.tolist(),qs[g],as_[g])
        if len(wp[1])<=64:
            assert list(wp[0])==list(gps[g][0]) and list(wp[1])==list(gps[g][1]), (c,g,"single",wp,gps[g])
    print(c,B,directed,flush=True)
print("label fuzz ok")
