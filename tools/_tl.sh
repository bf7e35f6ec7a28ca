repo=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
cd /tmp && rm -rf /tmp/pf_$v
lib=$repo/limap_amd/variants/lib$v.so; [ $v = main ] && lib=$repo/limap_amd/liblimap_amd.so
LIMAP_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -- python $repo/bench.py --no-cpu-baseline --no-extras --strong-leg off --steps 60 --warmup 3 > /dev/null 2>/tmp/pf_$v.err
db=$(find /tmp/pf_$v -name "*.db" | head -1)
echo "== $v"; python $repo/tools/rocpd_step_timeline.py $db | tail -3
done
