cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pA -- python3 tools/r2_one.py kin 40 4096 3 > $O/pA.log 2>&1; echo "A rc=$?"
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $O/pB -- python3 tools/r2_one.py kin 40 4096 3 > $O/pB.log 2>&1; echo "B rc=$?"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM --output-format csv -d $O/pC -- python3 tools/r2_one.py kin 40 4096 3 > $O/pC.log 2>&1; echo "C rc=$?"
tail -2 $O/pA.log $O/pB.log $O/pC.log
