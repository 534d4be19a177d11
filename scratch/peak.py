import sys; sys.path.insert(0,'.'); sys.path.insert(0,'poisson-gpfa_amd')
from funs import _hip
ctx=_hip.Context(8,2,64,2,10.0)
for it in (2000,20000,100000):
    print('mfma f64 sustained TF/s', it, ctx.bench_mfma_peak(it))
for (b,n,k) in ((64,2048,512),(256,2048,512),(64,4096,512),(64,4096,128)):
    ms,fl=ctx.bench_syrk(b,n,k,5); print('syrk',b,n,k,'ms',ms,'TF',fl/ms/1e9)
