import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests.test_gpu_llama import gemm
from llamarec_amd.synth import bf16_round, hash_uniform
for (M,N,K) in [(1000,256,4096),(1024,256,4096),(768,256,4096),(1000,512,4096),(1000,256,2048)]:
    A = bf16_round(hash_uniform(M*7+K,(M,K),1.0)); B = bf16_round(hash_uniform(N*13+K,(N,K),1.0))
    ref = A@B.T
    for rep in range(3):
        got = gemm(A,B,2)
        err = np.abs(got-ref); tol = np.maximum(np.abs(ref),1e-3)*2.0**-7
        bad = err>tol
        print((M,N,K), "rep",rep,"bad",int(bad.sum()),"of",bad.size,"maxerr",float(err.max()))
        if bad.any():
            r,c = np.nonzero(bad)
            print("  rows:", np.unique(r)[:20], "n_rows", len(np.unique(r)), " cols:", np.unique(c)[:20], "n_cols", len(np.unique(c)))
            print("  row%16 hist", np.bincount(r%16, minlength=16), "col%16 hist", np.bincount(c%16,minlength=16))
