# instruction-cache / scalar-cache behaviour of k_select (is the ~2 k cycles per level beyond the memory round trip instruction fetch?)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_icache; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|dcache|SQC_|INST_LEVEL|WAIT" | cut -c1-200 > $O/avail.txt
B="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 50"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d /tmp/i1 -o i1 -- $B > $O/i1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/i2 -o i2 -- $B > $O/i2.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE -d /tmp/i3 -o i3 -- $B > $O/i3.log 2>&1
cd $R
for k in 1 2 3; do python tools/prof_summary.py /tmp/i$k/i${k}_results.db 4 2>&1 | grep -E "k_select|k_v80" | cut -c1-30,100- > $O/i$k.md; done
cat $O/i1.md $O/i2.md $O/i3.md
tools/ubench/chase 4096 40 64; tools/ubench/chase 4096 40 64 7; tools/ubench/chase 4096 1 64; tools/ubench/chase 4096 0.0625 64; tools/ubench/chase 64 40 64
