import sys, os, threading
sys.path.insert(0, "/root/repo")
import windflow_amd
ext = windflow_amd.load_ext()
print("ext loaded", flush=True)
world=2
fabric = ext.LocalFabric(world)
stats=[None]*world
def rank_main(r):
    print("rank", r, "start", flush=True)
    pipe = ext.WinAggPipeline(0, 2, 8192, 5000, 4096, 1024, True, 8*8192, 4, ext.local_transport(fabric, r), 2, 2, 1.3, True, False, False, True)
    print("rank", r, "built", flush=True)
    pipe.step(200000)
    print("rank", r, "stepped", flush=True)
    pipe.close()
    stats[r]=pipe.stats()
ts=[threading.Thread(target=rank_main,args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
print(stats)
