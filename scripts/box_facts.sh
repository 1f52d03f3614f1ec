# what the GPU box gives a run: cores, cgroup CPU quota, memory, scratch disks, deflate libraries
echo "nproc $(nproc)  getconf $(getconf _NPROCESSORS_ONLN)"
python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
cat /sys/fs/cgroup/memory.max 2>/dev/null
df -h /tmp /dev/shm $GRAFT_REPO_ROOT 2>/dev/null
ldconfig -p | grep -i -E "deflate|libz|isal|zstd" 
ls /opt/conda/lib | grep -i -E "deflate|libz|isal" 
lscpu | head -20
