set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_h2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*Counter_Name|Name\s*:" | grep -E "SQ_|TCP_|TA_" | awk '{print $NF}' | sort -u | tr '\n' ' ' > $O/sq_counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/p1 -o p1 -- python $R/tools/run_h2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d /tmp/p2 -o p2 -- python $R/tools/run_h2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_FLAT -d /tmp/p3 -o p3 -- python $R/tools/run_h2.py > /dev/null 2>&1
cd $R
for k in 1 2 3; do python tools/prof_summary.py /tmp/p$k/p${k}_results.db 3 2>&1 | grep -E "h2|counter" > $O/p$k.md; done
cat $O/p1.md $O/p2.md $O/p3.md
