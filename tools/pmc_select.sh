set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_select; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 50"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d /tmp/s1 -o s1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAVES -d /tmp/s2 -o s2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d /tmp/s3 -o s3 -- $B > /dev/null 2>&1
cd $R
for k in 1 2 3; do python tools/prof_summary.py /tmp/s$k/s${k}_results.db 3 2>&1 | grep -E "k_select" > $O/s$k.md; done
cat $O/s1.md $O/s2.md $O/s3.md | cut -c1-40,110-
