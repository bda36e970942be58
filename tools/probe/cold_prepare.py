import sys, time, numpy as np
sys.path.insert(0,'piper-swift_amd/python'); sys.path.insert(0,'tests')
import piper_hip as ph, katdata as kd
b=ph.HipBackend(0); cfg=ph.voice_config("medium"); rt=ph.HipRuntime(b,cfg,ph.synthetic_blob(cfg,1234))
for T in (14, 112, 113, 130, 300, 896):
    ids=(kd.FIXTURE_IDS*100)[:T]; dur=[3]*T; noise=kd.sym(1,(192,3*T),1.7)
    t0=time.perf_counter(); rt.prepare(0,ids,dur,noise,0.667); t1=time.perf_counter(); rt.launch(0); rt.collect(0)
    print(T, "cold prepare ms %.2f"%((t1-t0)*1e3), rt.last_build_breakdown())
    t0=time.perf_counter(); rt.prepare(0,ids,dur,noise,0.667); t1=time.perf_counter(); rt.launch(0); rt.collect(0)
    print(T, "warm prepare ms %.3f"%((t1-t0)*1e3))
