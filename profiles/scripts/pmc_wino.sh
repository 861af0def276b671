#!/bin/bash
# usage: pmc_wino.sh <python script> <kernel substring> <out json>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wino; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pw_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pw_$i -- python3 $R/$1 > $O/pass_$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/toda_amd/tools/pmc_summary.py "$2" $R/gpurun_out/$3 $(find /tmp/pw_* -name "*counter_collection.csv") > /dev/null
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/$3"))
for s in d["launch_shapes"]:
    print(s["kernel"][:120], {k:(round(v/1e6,3) if isinstance(v,float) and v>1e5 else (round(v,3) if isinstance(v,float) else v)) for k,v in s.items() if k!="kernel"})
PY
