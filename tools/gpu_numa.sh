#!/bin/bash
# the CLI by hand on the bench inputs (no parent process holding the GPU), free / bound to either NUMA node
cd "$GRAFT_REPO_ROOT" || exit 1
NTEDIT_BENCH_KEEP_E2E=1 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gather 2>&1 | grep "kept in" > /tmp/kept.txt
W=$(sed 's/.*kept in //' /tmp/kept.txt); echo "work=$W"; ls $W | head
lscpu | grep -i "numa\|socket" | head -8
for d in /sys/class/drm/card*/device; do echo "$d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done | head -12
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -2; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { "$@" $GRAFT_REPO_ROOT/ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/o --report 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['seconds'], d['polish_call_s'], d['screen_ms'])"; rm -f $W/o_*; }
for i in 1 2 3 4 5 6; do echo -n "bound by the CLI (default): "; run env; done
for i in 1 2 3 4 5 6; do echo -n "NTEDIT_HIP_NO_BIND=1: "; run env NTEDIT_HIP_NO_BIND=1; done
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 cpus $N0 ; node1 cpus $N1"
for i in 1 2 3 4; do echo -n "node0: "; run taskset -c $N0; done
[ -n "$N1" ] && for i in 1 2 3 4; do echo -n "node1: "; run taskset -c $N1; done
rm -rf $W  # (the kept inputs)
