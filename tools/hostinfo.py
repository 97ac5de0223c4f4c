import os, time
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; rocm-smi --showclocks 2>/dev/null | head -20")
