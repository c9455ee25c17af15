import sys, numpy as np
sys.path.insert(0, '.')
import bds_amd, bench
base = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)), acqCohT=10, pilotACQflag=1)
s, ch, blocks, order, shift, n, spc = bench.cfg4_record(base, 400)
x = bench.record_bytes(blocks, order, shift, n)
res, _ = bds_amd.tracking(x, ch, s, mode="WB")
for r in res[:12]:
    h = len(r.DataCNo)//2
    z = r.I_P**2 + r.Q_P**2; zp = r.Pilot_I_P**2 + r.Pilot_Q_P**2
    print(r.PRN, "Data %.2f Pilot %.2f Sig %.2f" % (np.mean(r.DataCNo[h:]), np.mean(r.PilotCNo[h:]), np.mean(r.B1C_CNo[h:])),
          "| data z: mean %.4g std/mean %.4f  pilot z: mean %.4g std/mean %.4f" % (z[200:].mean(), z[200:].std()/z[200:].mean(), zp[200:].mean(), zp[200:].std()/zp[200:].mean()),
          "IP %.4g QP %.4g pIP %.4g pQP %.4g" % (np.abs(r.I_P[200:]).mean(), np.abs(r.Q_P[200:]).mean(), np.abs(r.Pilot_I_P[200:]).mean(), np.abs(r.Pilot_Q_P[200:]).mean()))
print("CNoInterval", s.CNoInterval, "intTime", s.intTime)
