#!/usr/bin/env python3
"""The casts (per-row int8 / fp8, MXFP8 1 x 32) and the weight preparations as HBM streams: us per call and algorithmic TB/s (bf16 in + codes and scales out),
cold inputs (several copies per size in one hipGraph).  python tools/cast_bw.py > profiles/cast_bw_rNN.jsonl"""
import torch, json, sys, os
sys.path.insert(0, os.getcwd())
from ao_amd import ops
from tools.bench_dec8 import graph_time
dev=torch.device('cuda',0)
for m,k in [(128,4096),(128,8192),(2048,4096),(2048,14336),(16384,4096),(16384,14336),(65536,4096)]:
    n=max(2, (600<<20)//(m*k*2))
    n=min(n,16)
    xs=[torch.randn(m,k,device=dev,dtype=torch.bfloat16) for _ in range(n)]
    for name,fn,outb in [('int8_rowwise',ops.int8_quantize_rowwise,1.0),('fp8_rowwise',ops.fp8_quantize_rowwise,1.0),('mx_row_rceil',lambda x: ops.mxfp8_quantize(x,"rceil"),1+1/32)]:
        try:
            t=graph_time([lambda x=x: fn(x) for x in xs])
            print(json.dumps({'op':name,'M':m,'K':k,'us':round(t*1e6,2),'TBps':round(m*k*(2+outb)/t/1e12,2)}),flush=True)
        except Exception as e:
            print(json.dumps({'op':name,'M':m,'K':k,'error':repr(e)[:150]}),flush=True)
for n_,k in [(4096,4096),(14336,4096),(4096,14336),(28672,8192)]:
    ws=[torch.randn(n_,k,device=dev,dtype=torch.bfloat16)*0.02 for _ in range(3)]
    for name,fn,outb in [('int4_tinygemm_g128',lambda w: ops.int4_quantize_tinygemm(w,128),0.5+4/128),('int8_w',ops.int8_quantize_rowwise,1.0),('mx_w',lambda w: ops.mxfp8_quantize(w,"rceil"),1+1/32)]:
        try:
            t=graph_time([lambda w=w: fn(w) for w in ws])
            print(json.dumps({'op':name,'N':n_,'K':k,'us':round(t*1e6,2),'TBps':round(n_*k*(2+outb)/t/1e12,2)}),flush=True)
        except Exception as e:
            print(json.dumps({'op':name,'N':n_,'K':k,'error':repr(e)[:150]}),flush=True)
