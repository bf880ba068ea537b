set -x
mkdir -p gpurun_out/r05e
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; echo "nproc: $(nproc)"; echo "loadavg: $(cat /proc/loadavg)"; echo "mem: $(grep -E 'MemTotal|MemAvailable' /proc/meminfo | tr '\n' ' ')"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8; } > gpurun_out/r05e/host_diag.txt 2>&1
python3 - >> gpurun_out/r05e/host_diag.txt 2>&1 <<'PY'
# pure CPU scaling: N processes spinning on arithmetic for 1 s each; aggregate iterations per second
import multiprocessing as mp, time
def spin(q):
    t0=time.perf_counter(); n=0
    while time.perf_counter()-t0<1.0:
        x=0
        for i in range(20000): x+=i*i
        n+=1
    q.put(n)
for N in (1,8,32,64,128,256):
    q=mp.Queue(); ps=[mp.Process(target=spin,args=(q,)) for _ in range(N)]
    [p.start() for p in ps]; tot=sum(q.get() for _ in ps); [p.join() for p in ps]
    print(f"spin N={N:3d} aggregate {tot} per-proc {tot/N:.1f}")
PY
cat gpurun_out/r05e/host_diag.txt
bash tools/ab_libs.sh 2 $PWD/variants/libma_head.so tree > gpurun_out/r05e/ab_epilogue.txt 2>&1
cat gpurun_out/r05e/ab_epilogue.txt
python -m pytest tests/test_gpu_primitives.py -k "farneback" -x -q > gpurun_out/r05e/pytest_fb.log 2>&1; tail -2 gpurun_out/r05e/pytest_fb.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r05e/freg_kt --output-format csv -- python3 tools/freg_profile.py 4096 > gpurun_out/r05e/freg_under_trace.txt 2>&1
python3 - <<'PY'
import csv,glob
fs=sorted(glob.glob('gpurun_out/r05e/freg_kt/*/*_kernel_stats.csv'))
rows=list(csv.DictReader(open(fs[-1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over 6 register() calls', tot/1e6)
for r in rows[:25]: print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Calls']:>5} {r['Name'][:90]}")
PY
