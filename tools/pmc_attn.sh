#!/bin/bash
# PMC passes over the attention shapes only (fabric-side bytes, SQ set): bash tools/pmc_attn.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${1:-attn}; mkdir -p "$O"; cd $R
cd /tmp; export TMPDIR=/tmp; cd $R
export PMC_MANIFEST=$O/${TAG}_pmc_manifest.json
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  T=${P%%:*}; C=${P#*:}
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${TAG}_pass_$T -o p -- python3 tools/pmc_target.py attn > $O/pmc_${TAG}_pass_$T.log 2>&1 || { echo "PMC pass $T failed"; tail -3 $O/pmc_${TAG}_pass_$T.log; exit 1; }
  python3 tools/pmc_collect.py $O/pmc_${TAG}_pass_$T $PMC_MANIFEST $O/${TAG}_pmc_$T.json || exit 1
  rm -rf $O/pmc_${TAG}_pass_$T
done
python3 - <<PY
import json
f=json.load(open("$O/${TAG}_pmc_fetch.json")); w=json.load(open("$O/${TAG}_pmc_write.json")); q=json.load(open("$O/${TAG}_pmc_sq.json"))
for a,b,c in zip(f,w,q):
    fab=2*a["counters"].get("FETCH_SIZE",0)*1024+b["counters"].get("WRITE_SIZE",0)*1024
    cc=c["counters"]; gui=cc.get("GRBM_GUI_ACTIVE",0)/8
    print(f'{a["cls"]:9s} {a["label"]:18s} fabric {fab/1e6:8.1f} MB  algorithmic {a["algorithmic_bytes_per_launch"]/1e6:7.1f} MB  ratio {fab/a["algorithmic_bytes_per_launch"]:5.2f}x   '
          f'LDS bank-conflict cycles / LDS active {cc.get("SQ_LDS_BANK_CONFLICT",0)/max(cc.get("SQ_LDS_IDX_ACTIVE",1),1):.3f}  VALU per MFMA {cc.get("SQ_INSTS_VALU",0)/max(cc.get("SQ_INSTS_MFMA",1),1):.1f}  MFMA busy {cc.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(gui*256*4,1):.3f}')
PY
