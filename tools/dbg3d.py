import sys
sys.path[:0]=['/root/repo','/root/repo/lsa-fw_amd']
import numpy as np, scipy.sparse as sp, lsa_hip
from synthetic import fem
ctx=lsa_hip.Context(0)
for name in ("C9k","C20k"):
    es=fem.cube_case(name)
    for sig in (fem.SIGMA_CUBE, 0.3+0.2j):
        C=sp.csr_matrix((es.A.data-sig*es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
        for leaf in (64,128,256):
            try:
                f=lsa_hip.NdLu(ctx, lsa_hip.CsrMatrix.from_scipy(ctx,C), leaf)
                b=np.ones(es.n,dtype=C.dtype); dx=lsa_hip.DeviceVector(ctx,es.n,C.dtype); f.solve(lsa_hip.DeviceVector.from_numpy(ctx,b),dx)
                print(name,sig,leaf,"ok residual",np.linalg.norm(C@dx.numpy()-b)/np.linalg.norm(b), f.info()["max_front"], flush=True)
                del f
            except Exception as e:
                print(name,sig,leaf,"FAIL",e, flush=True)
