import os, sys, time
sys.path.insert(0, os.getcwd())
import oracle
oracle.build()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
nv=22
st=oracle.random_scalars(5,nv+1)
for th in (1,4,8,16,32,64,128,256):
    mls=[oracle.random_b128(10+j,1<<nv) for j in range(2)]
    t=time.perf_counter(); r=oracle.fast_bivariate_sumcheck_prove(mls,nv,[(0,1)],[0],st[0],st[1:],threads=th); dt=time.perf_counter()-t
    print(th,"threads fast:", round(2*(1<<nv)/dt/1e6,1),"M elems/s", round(dt,3),"s", flush=True)
