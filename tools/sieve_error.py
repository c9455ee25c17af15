# sieve accuracy: GPU search grid vs oracle row maxima on the reduced-rate B1C case
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, bds_amd
from oracle import acquisition as oacq
from helpers import medium_b2a
s,x,_=medium_b2a()
diag={}; ref=oacq.acquisition_b2a(x.astype(np.float64), s, diag)
got=bds_amd.acquisition(x,s,verbose=False)
ctx=bds_amd.get_context(0)
rm,ra=ctx.acq_grid(4,26)
rel=np.array([rm[i]/diag[p]["row_max"]-1 for i,p in enumerate([5,9,19,33])])
print("mode", ctx.timing()["half_storage"], "grid rel err: max %.2e rms %.2e"%(np.abs(rel).max(), np.sqrt((rel**2).mean())), "argmatch", np.mean([ra[i]==diag[p]["row_arg"] for i,p in enumerate([5,9,19,33])]))
print(np.array_equal(got.codePhase, ref.codePhase), np.array_equal(got.carrFreq, ref.carrFreq), np.max(np.abs(got.peakMetric/np.where(ref.peakMetric==0,1,ref.peakMetric)-1)))
