import sys, time
sys.path.insert(0, '/root/repo')
import bench
if __name__ == '__main__':
    for procs, threads in ((16, 4), (32, 4), (32, 8)):
        t = time.time(); r = bench.cpu_baseline_sharded(procs=procs, threads=threads); print(procs, threads, round(r['value'], 1), r['sample'][-40:], round(time.time() - t, 1), flush=True)
